#!/bin/bash
# A/B of environment switches on one box (run through gpurun from the repo root):  tools/ab_env.sh <tag> "<VAR=val ...>" "<VAR=val ...>" ...
# every setting runs `bench.py --no-cpu-baseline` once; prints ms_per_step and the stage times, keeps the lines under gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
i=0
for setting in "$@"; do
  i=$((i+1))
  ( for kv in $setting; do [ "$kv" != "-" ] && export "$kv"; done; cd $R && python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/$TAG/run$i.json 2> gpurun_out/$TAG/run$i.err )
  python - "$setting" $R/gpurun_out/$TAG/run$i.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[2]))
    st = j["stages_ms_per_step"]
    print("%-40s %7.1f ms  %s" % (sys.argv[1], j["ms_per_step"], " ".join("%s=%.1f" % (k.replace("group_", "g_").replace("conflict_", "c_"), v) for k, v in st.items())))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
