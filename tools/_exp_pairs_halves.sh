# RB_PAIRS_SIDE=3 (default: the walker beside the emit pass and whatever follows) against 5 (half beside the emit pass, half beside the bucket kernel); run through gpurun
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for m in 3 5; do RB_PAIRS_SIDE=$m python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_step']; print('side=$m %.1f ms pairs %d'%(d['ms_per_step'], d['config'].get('read_pairs_per_step',0)), {k:round(s[k],1) for k in ('pairs_insert','hash_windows','group_part_count','group_part_scatter','group_buckets','filter_windows','probe_claim')})"; done; done
