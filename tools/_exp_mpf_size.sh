# prefilter cache size sweep (2^RB_MPF buckets of 128 B; default 25 at config 2's filter size); run through gpurun
cd $GRAFT_REPO_ROOT
for i in 1 2; do for m in 25 23 24 26; do RB_MPF=$m python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_step']; print('log2 buckets=$m %.1f ms, sorted %.3f G'%(d['ms_per_step'], d['config']['sorted_kmers_per_step']/1e9), {k:round(s[k],1) for k in ('filter_windows','probe_claim','resolve_apply','hash_windows','group_part_scatter')})"; done; done
