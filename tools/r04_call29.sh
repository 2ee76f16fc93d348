#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5 > $O/r04_ad_tests.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_lr
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lr -o s -- python $R/tools/longread_insert_ab.py 1500000 > $O/r04_ad_longread.txt 2>&1
python $R/profiles/summarize.py stats $(find /tmp/prof_lr -name '*kernel_stats.csv' | head -1) > $O/r04_ad_kernel_stats.csv
cd $R; for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_ad_bench$i.json; done
