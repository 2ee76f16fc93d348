#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for v in "RB_WINDOW_MUL=3" "RB_WINDOW_MUL=4" "RB_WINDOW_MUL=2" "RB_RAMP_DIV=32" "RB_RAMP_DIV=128" "RB_WINDOW_MUL=4 RB_RAMP_DIV=128"; do
  n=$(echo $v | tr ' =' '__')
  env $v python bench.py --no-cpu-baseline > $O/r04_j_bench_$n.json 2>/dev/null
done
