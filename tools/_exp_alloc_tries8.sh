# A/B of RB_ALLOC_TRIES = 4 / 8 in fresh processes; run through gpurun
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for t in 4 8; do RB_ALLOC_TRIES=$t RB_ALLOC_DEBUG=1 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tries=$t %.1f ms'%d['ms_per_step'], {k:round(v,1) for k,v in d['stages_ms_per_step'].items() if v>38})"; grep "cbf allocation" /tmp/err.txt | sed 's/.*allocation \([0-9]\): \([0-9.]*\) ms.*/\1:\2/' | tr '\n' ' '; echo; done; done
