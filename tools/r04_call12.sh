#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_l8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l8 -o s -- python $R/tools/loopback_bench.py --ranks 8 --pairs 50000000 > $O/r04_l_loop8_py_prof.txt 2>$O/r04_l_loop8_py_prof.err
python $R/profiles/summarize.py stats $(find /tmp/prof_l8 -name '*kernel_stats.csv' | head -1) > $O/r04_l_loop8_py_kernels.csv
cd $R
python tools/loopback_bench.py --ranks 4 --pairs 50000000 > $O/r04_l_loop4_py.txt 2>&1
python tools/loopback_bench.py --ranks 2 --pairs 50000000 > $O/r04_l_loop2_py.txt 2>&1
python tools/loopback_bench.py --ranks 1 --pairs 50000000 > $O/r04_l_loop1_py.txt 2>&1
python tools/loopback_bench.py --ranks 1 --pairs 50000000 --native > $O/r04_l_loop1_native.txt 2>&1
