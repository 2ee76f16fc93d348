#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_lr
RB_EARLY_AUTO=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lr -o s -- python $R/tools/longread_insert_ab.py 1500000 > $O/r04_t_longread.txt 2>&1
python $R/profiles/summarize.py stats $(find /tmp/prof_lr -name '*kernel_stats.csv' | head -1) > $O/r04_t_kernel_stats.csv
