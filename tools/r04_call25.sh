#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_z_bench$i.json; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r04_z_bench_rocprof.json 2>/dev/null
python $R/profiles/summarize.py stats $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) > $O/r04_z_kernel_stats.csv
