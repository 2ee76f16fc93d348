# A/B of the paired-k-mer walker's placement (profiles/r03_pairs_side.txt); run through gpurun from the repo root
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/exp_$name.json 2>/dev/null; python - $name <<'PY'
import json,sys
d=json.load(open("gpurun_out/exp_%s.json"%sys.argv[1]))
st=d["stages_ms_per_step"]
print(sys.argv[1], "%.1f ms"%d["ms_per_step"], "pairs %d"%d["config"]["read_pairs_per_step"], {k:round(v) for k,v in st.items() if v>15})
PY
}
python -c "import torch; print(torch.cuda.get_device_name(0))"
run inline RB_PAIRS_SIDE=0
run side1 RB_PAIRS_SIDE=1
run side2 RB_PAIRS_SIDE=2
run default RB_PAIRS_SIDE=3
run side4 RB_PAIRS_SIDE=4
run inline_again RB_PAIRS_SIDE=0
run default_again RB_PAIRS_SIDE=3
