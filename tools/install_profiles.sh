#!/bin/bash
# Copies what tools/final_profile.sh, tools/loopback_profile.sh and tools/parity_at_size_set.sh left under gpurun_out/ into profiles/<tag>_*
# (run here, after the gpurun call that produced them):  tools/install_profiles.sh r06
# Refuses when the PMC passes were not made with the library this tree builds (csrc id).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); TAG=${1:-r06}; F=$R/gpurun_out/final_$TAG; P=$R/profiles
want=$(python3 $R/tools/csrc_id.py)
have=$(python3 -c "import json; print(json.load(open('$F/pmc_meta.json'))['csrc_id'])")
[ "$want" = "$have" ] || { echo "install_profiles: gpurun_out/final_$TAG was made with csrc id $have, this tree is $want"; exit 1; }
tail -1 $F/bench.json > $P/${TAG}_bench.json
tail -1 $F/bench_serial_stages.json > $P/${TAG}_bench_serial_stages.json
tail -1 $F/bench_under_rocprof.json > $P/${TAG}_bench_under_rocprof.json
cp $F/kernel_stats.csv $P/${TAG}_kernel_stats.csv
cp $F/pmc_FETCH_SIZE.csv $P/${TAG}_pmc_fetch_size.csv
cp $F/pmc_WRITE_SIZE.csv $P/${TAG}_pmc_write_size.csv
cp $F/pmc_meta.json $P/${TAG}_pmc_meta.json
cp $F/pmc_calibration.txt $P/${TAG}_pmc_calibration.txt
cp $F/pytest_gpu.txt $P/${TAG}_pytest_gpu.txt
[ -d $R/gpurun_out/loop_$TAG ] && python3 $R/tools/loopback_summary.py $TAG > $P/${TAG}_loopback.txt
[ -f $R/gpurun_out/${TAG}_parity_at_size.txt ] && cp $R/gpurun_out/${TAG}_parity_at_size.txt $P/${TAG}_parity_at_size.txt
python3 - <<PY
import json
b = json.load(open("$P/${TAG}_bench.json")); h = b["host_resident"]; r = b["roofline"]
print("bench: %.1f ms = %.2f G k-mers/s; host-resident %.1f ms = %.2f G; dominant kernel %.2f ms per launch, frac %.3f; traffic %s"
      % (b["ms_per_step"], b["value"] / 1e9, h["ms_per_step"], h["value"] / 1e9, r["avg_launch_ms"], r["frac"], r["traffic"]))
print(open("$P/${TAG}_pytest_gpu.txt").read().strip())
PY
grep -c "equal: \[True, True, True\]" $P/${TAG}_parity_at_size.txt || true
grep -c "False" $P/${TAG}_parity_at_size.txt || true
head -9 $P/${TAG}_loopback.txt | cut -c1-220
