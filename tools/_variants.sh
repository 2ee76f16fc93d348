cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fastq.py -x -q 2>&1 | grep -E "passed|failed|FAILED|^E " | tail -6
one() { name=$1; shift; env "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name: %.1f ms, %.2f G k-mers/s, sorted %.2f G'%(d['ms_per_step'], d['value']/1e9, d['config']['sorted_kmers_per_step']/1e9), {k:round(v,1) for k,v in d['stages_ms_per_step'].items() if v>30})"; }
one "k=35" A=1 python bench.py --no-cpu-baseline --k 35
one "k=35 no resume" RB_EMIT_RESUME=0 python bench.py --no-cpu-baseline --k 35
one "k=47" A=1 python bench.py --no-cpu-baseline --k 47
one "k=63" A=1 python bench.py --no-cpu-baseline --k 63
