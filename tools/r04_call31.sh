#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_af_bench$i.json; RB_PAIRS_NOCOUNT=1 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_af_bench_nocount$i.json; done
