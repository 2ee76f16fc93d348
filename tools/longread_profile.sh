#!/bin/bash
# The long-read (configs[4]) evidence of profiles/r04_longreads.txt, run through gpurun from the repo root:  tools/longread_profile.sh
#   the full 5 M-read tool, the A/B of the swept Bloom-bit stage at 1.5 M reads (filters compared), and the kernels of the swept path.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/longread; mkdir -p $O
cd $R
timeout 1500 python tools/longread_full.py 5000000 > $O/full_5m.txt 2>&1
timeout 900 python tools/longread_insert_ab.py 1500000 RB_SWEEP=0 > $O/ab_1500k.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_lr
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lr -o s -- python $R/tools/longread_insert_ab.py 1500000 > $O/under_rocprof_1500k.txt 2>&1
python $R/profiles/summarize.py stats $(find /tmp/prof_lr -name '*kernel_stats.csv' | head -1) > $O/kernel_stats_1500k.csv
