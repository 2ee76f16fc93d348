# bench.py at other k and with unmasked errors (profiles/r03_variants.txt); run through gpurun
cd $GRAFT_REPO_ROOT
one() { name=$1; shift; env "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name: %.1f ms, %.2f G k-mers/s, sorted %.2f G'%(d['ms_per_step'], d['value']/1e9, d['config']['sorted_kmers_per_step']/1e9), {k:round(v,1) for k,v in d['stages_ms_per_step'].items() if v>40})"; }
one "default" A=1 python bench.py --no-cpu-baseline
one "errors unmasked" RB_SYNTH_KEEP_ERRORS=1 python bench.py --no-cpu-baseline
one "k=31" A=1 python bench.py --no-cpu-baseline --k 31
one "k=35" A=1 python bench.py --no-cpu-baseline --k 35
one "k=47" A=1 python bench.py --no-cpu-baseline --k 47
one "k=63" A=1 python bench.py --no-cpu-baseline --k 63
