#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_lr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lr -o s -- python $R/tools/longread_insert_ab.py 1500000 > $O/r04_f_longread_prof.txt 2>&1
python $R/profiles/summarize.py stats $(find /tmp/prof_lr -name '*kernel_stats.csv' | head -1) > $O/r04_f_longread_kernels.csv
cd $R
python -m pytest tests/test_gpu_scale.py -x -q -k index_keyed 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/r04_f_scale.txt
