cd $GRAFT_REPO_ROOT
one() { name=$1; shift; env "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name: %.1f ms, %.2f G k-mers/s, sorted %.3f G'%(d['ms_per_step'], d['value']/1e9, d['config']['sorted_kmers_per_step']/1e9), {k:round(v,1) for k,v in d['stages_ms_per_step'].items() if v>40})"; }
cp rna-bloom_amd/lib/librb_hip.so /tmp/orig.so
for t in 21 19 17 21; do
  if [ $t = 25 ]; then cp /tmp/orig.so rna-bloom_amd/lib/librb_hip.so; else cp gpurun_scratch/kp/librb_hip_kp$t.so rna-bloom_amd/lib/librb_hip.so; fi
  one "kp=$t k=25" A=1 python bench.py --no-cpu-baseline
  one "kp=$t k=35" A=1 python bench.py --no-cpu-baseline --k 35
done
