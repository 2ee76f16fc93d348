# A/B runs of bench.py under environment switches (development helper; run through gpurun)
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/v_$name.json 2> gpurun_out/v_$name.err; }
run late3 RB_X=0
run early3 RB_PREPARE_EARLY=1
run early2 RB_PREPARE_EARLY=1 RB_WINDOW_MUL=2
run early1 RB_PREPARE_EARLY=1 RB_WINDOW_MUL=1
run late2 RB_WINDOW_MUL=2
run late1 RB_WINDOW_MUL=1
for f in gpurun_out/v_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:d['stages_ms_per_step'][k] for k in ('filter_windows','hash_windows','sort_occurrences','probe_claim')}, d['config']['sorted_kmers_per_step'], d['config']['conflict_ops_per_step'])
PY
done
