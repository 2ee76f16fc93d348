cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error|^E " | tail -12
python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f ms'%d['ms_per_step'], {k:round(v,1) for k,v in d['stages_ms_per_step'].items() if v>5}); print(d['roofline']['frac'], d['roofline']['kernel'])"
