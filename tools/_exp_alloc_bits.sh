# A/B: bit filters through the placement trial too (RB_ALLOC_BITS=1: in creation order, =2: the counting filter draws first); run through gpurun
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for t in 0 1 2; do
  if [ $t = 0 ]; then unset RB_ALLOC_BITS; else export RB_ALLOC_BITS=$t; fi
  RB_ALLOC_DEBUG=1 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bits=$t %.1f ms'%d['ms_per_step'], {k:round(v,1) for k,v in d['stages_ms_per_step'].items() if v>38})"; grep "allocation" /tmp/err.txt | sed 's/\[rb\] \([a-z]*\) allocation \([0-9]\): \([0-9.]*\) ms.*/\1\2:\3/' | tr '\n' ' '; echo; done; done
