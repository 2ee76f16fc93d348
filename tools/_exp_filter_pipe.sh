# A/B of the prefilter with its bucket fetch one / two steps ahead of its use (RB_FILTER_PIPE=1 / 2); run through gpurun from the repo root
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/exp_$name.json 2>/dev/null; python - $name <<'PY'
import json,sys
d=json.load(open("gpurun_out/exp_%s.json"%sys.argv[1]))
st=d["stages_ms_per_step"]
print(sys.argv[1], "%.1f ms"%d["ms_per_step"], "sorted %.3fG"%(d["config"]["sorted_kmers_per_step"]/1e9), {k:round(v,1) for k,v in st.items() if v>15})
PY
}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|FAILED" | tail -3
run base RB_FILTER_PIPE=0
run pipe1 RB_FILTER_PIPE=1
run base_again RB_FILTER_PIPE=0
run pipe1_again RB_FILTER_PIPE=1
