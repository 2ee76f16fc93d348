cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/exp_$name.json 2>/dev/null; python - $name <<'PY'
import json,sys
d=json.load(open("gpurun_out/exp_%s.json"%sys.argv[1]))
st=d["stages_ms_per_step"]
print(sys.argv[1], "%.1f ms"%d["ms_per_step"], "sorted %.2fG"%(d["config"]["sorted_kmers_per_step"]/1e9), {k:round(v) for k,v in st.items() if v>15})
PY
}
python -c "import torch; print(torch.cuda.get_device_name(0))"
run off RB_TWO_PHASE=0
run on A=1
run eqprio RB_PRODUCER_PRIORITY=0
run consprio RB_PRODUCER_PRIORITY=0 RB_CONSUMER_PRIORITY=-1
run rst24 RB_RST=24
run rst24_consprio RB_RST=24 RB_PRODUCER_PRIORITY=0 RB_CONSUMER_PRIORITY=-1
run rst0 RB_RST=0
