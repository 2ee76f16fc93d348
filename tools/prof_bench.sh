#!/bin/bash
# rocprofv3 kernel stats of one bench step (run through gpurun from the repo root): prof_bench.sh <tag> [env assignments...]
R=$GRAFT_REPO_ROOT; TAG=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_${TAG}_rocprof.json 2> $R/gpurun_out/rocprof_${TAG}.err
python $R/profiles/summarize.py stats $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) > $R/gpurun_out/kernel_stats_${TAG}.csv
head -30 $R/gpurun_out/kernel_stats_${TAG}.csv
