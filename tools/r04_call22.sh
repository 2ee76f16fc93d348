#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_primitives.py -x -q 2>&1 | tail -25 > $O/r04_w_tests.txt
RB_DEBUG=1 timeout 600 python tools/longread_insert_ab.py 250000 2>&1 | grep -E "swept stage|G k-mers" | head -12 > $O/r04_w_debug.txt
timeout 900 python tools/longread_insert_ab.py 1500000 RB_SWEEP_LEAN=0 RB_EARLY_AUTO=0 > $O/r04_w_longread_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_lr
RB_EARLY_AUTO=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lr -o s -- python $R/tools/longread_insert_ab.py 1500000 > $O/r04_w_longread.txt 2>&1
python $R/profiles/summarize.py stats $(find /tmp/prof_lr -name '*kernel_stats.csv' | head -1) > $O/r04_w_kernel_stats.csv
