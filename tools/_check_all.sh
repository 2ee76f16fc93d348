cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|FAILED|^E " | tail -5
for i in 1 2 3; do python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms'%d['ms_per_step'], {k:round(v,1) for k,v in d['stages_ms_per_step'].items() if v>25})"; done
