# A/B runs of bench.py under environment switches (development helper; run through gpurun)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x 2>&1 | tail -3
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/v_$name.json 2> gpurun_out/v_$name.err; }
run resume RB_X=0
run noresume RB_EMIT_RESUME=0
run resume2 RB_X=0
for f in gpurun_out/v_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:d['stages_ms_per_step'][k] for k in ('filter_windows','hash_windows','sort_occurrences')}, d['config']['sorted_kmers_per_step'])
PY
done
