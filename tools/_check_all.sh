cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do RB_RESOLVE_PREFETCH=$v python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefetch=$v %.1f ms'%d['ms_per_step'], {k:round(v,1) for k,v in d['stages_ms_per_step'].items() if v>25})"; done
