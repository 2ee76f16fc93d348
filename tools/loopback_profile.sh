#!/bin/bash
# The sharded engine's per-rank work on one GPU (run through gpurun from the repo root):  tools/loopback_profile.sh <tag> [kernels]
#   tools/loopback_bench.py at 1, 2, 4, 8 virtual ranks with the Python driver (phase split) and the exchange driver below the C ABI;
#   with "kernels": also rocprofv3 kernel stats of the 8-rank runs of both drivers.
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}; OUT=$R/gpurun_out/loop_$TAG
mkdir -p $OUT
cd $R
for g in 8 4 2 1; do
  python tools/loopback_bench.py --ranks $g --pairs 50000000 --native 2>/dev/null | tail -1 > $OUT/native_$g.json
  python tools/loopback_bench.py --ranks $g --pairs 50000000 --trace 2>/dev/null | tail -1 > $OUT/python_$g.json
done
if [ "$2" = "kernels" ]; then
  cd /tmp && export TMPDIR=/tmp
  for drv in native python; do
    rm -rf /tmp/prof_lb
    flag=""; [ $drv = native ] && flag="--native"
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lb -o s -- python $R/tools/loopback_bench.py --ranks 8 --pairs 50000000 --warmup 0 $flag > $OUT/under_rocprof_$drv.json 2> $OUT/rocprof_$drv.err
    python $R/profiles/summarize.py stats $(find /tmp/prof_lb -name '*kernel_stats.csv' | head -1) > $OUT/kernels8_$drv.csv
  done
fi
cd $R
for f in $OUT/native_*.json $OUT/python_*.json; do echo "$(basename $f): $(python -c "
import json,sys
j=json.load(open('$f')); print(j['wall_ms_per_step'], j['per_rank_ms_if_concurrent'], j.get('phase_ms_per_step_all_ranks',''))")"; done
[ -f $OUT/kernels8_native.csv ] && head -40 $OUT/kernels8_native.csv
