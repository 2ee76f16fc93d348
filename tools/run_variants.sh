# A/B runs of bench.py under environment switches (development helper; run through gpurun)
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/v_$name.json 2> gpurun_out/v_$name.err; }
run base RB_X=0
run async RB_PAIRS_ASYNC=1
run base2 RB_X=0
run async2 RB_PAIRS_ASYNC=1
for f in gpurun_out/v_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:d['stages_ms_per_step'].get(k) for k in ('filter_windows','hash_windows','sort_occurrences','pairs_insert','probe_claim')}, d['config']['read_pairs_per_step'])
PY
done
