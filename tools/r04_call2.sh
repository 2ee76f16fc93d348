#!/bin/bash
# round-4 measurement batch 2 (through gpurun): at-size tests, allocation experiment, sharded loopback profile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_configs_at_size.py -x -q 2>&1 | tail -15 > $O/r04_b_atsize.txt
timeout 600 tools/microbench/alloc_explain > $O/r04_b_alloc.txt 2>&1
python tools/loopback_bench.py --ranks 8 --pairs 50000000 --trace > $O/r04_b_loop8_py.txt 2>&1
python tools/loopback_bench.py --ranks 1 --pairs 50000000 --trace > $O/r04_b_loop1_py.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_l8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l8 -o s -- python $R/tools/loopback_bench.py --ranks 8 --pairs 50000000 --native > $O/r04_b_loop8_native.txt 2>$O/r04_b_loop8_native.err
python $R/profiles/summarize.py stats $(find /tmp/prof_l8 -name '*kernel_stats.csv' | head -1) > $O/r04_b_loop8_native_kernels.csv
