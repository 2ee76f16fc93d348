#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
tools/final_profile.sh r04 > $O/final_r04_stdout.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_lr
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lr -o s -- python $R/tools/longread_insert_ab.py 1500000 > $O/final_r04/longread_1500k_under_rocprof.txt 2>&1
python $R/profiles/summarize.py stats $(find /tmp/prof_lr -name '*kernel_stats.csv' | head -1) > $O/final_r04/longread_1500k_kernel_stats.csv
