(echo "# tools/parity_at_size.py on one MI355X (round 2 tree: hand-written grouping with group repair, k_probe_h2, minimizer cache for 2 replicated ranks): single-GPU engine, sharded engine (8 virtual ranks split reads, 2 replicated hashing) against the CPU oracle, bit for bit"
echo "## k25_16M"; timeout 1500 python tools/parity_at_size.py 16000000 25 2>&1 | grep -v amdgpu.ids
echo "## k35_4M"; timeout 900 python tools/parity_at_size.py 4000000 35 2>&1 | grep -v amdgpu.ids
echo "## k64_4M"; timeout 900 python tools/parity_at_size.py 4000000 64 2>&1 | grep -v amdgpu.ids) > gpurun_out/parity_at_size.txt
cat gpurun_out/parity_at_size.txt
