#!/bin/bash
# ISA-level differential on one kernel of one translation unit: same instructions, same registers — only wait states or the kernel descriptor change.
#   tools/isa_variants/setup.sh <tree> [extra hipcc flags]      (tree = a checkout of this repository; for the k_pairs_insert failure: git archive 306b610)
# compiles <tree>/rna-bloom_amd/csrc/rb_graph.hip with -save-temps -v into <tree>/rna-bloom_amd/st/, keeps the device assembly as dev_orig.s and
# writes rebuild_from_s.sh: the six commands hipcc runs BEHIND the device assembly (assemble, link, bundle, host compile with the bundle embedded),
# so that an edited rb_graph-hip-amdgcn-amd-amdhsa-gfx950.s becomes build/rb_graph.o again.  mk_variant.py makes the edits; isa_probe.sh (run through
# gpurun) loops over variants: rebuild, relink the library, count wrong bits with tools/pairs_variants.py.  Findings: profiles/r05_miscompile.md.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
T=$(cd "$1" && pwd)/rna-bloom_amd; shift
mkdir -p $T/build $T/st && cd $T/st
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DRB_DIAG_PAIRS "$@" -save-temps -v -c ../csrc/rb_graph.hip -o ../build/rb_graph.o > v.log 2>&1
cp rb_graph-hip-amdgcn-amd-amdhsa-gfx950.s dev_orig.s
python3 - <<'PY'
lines = [l.strip() for l in open('v.log') if l.startswith(' "')]
sel = [l for l in lines if '-cc1as -triple amdgcn' in l or '/lld"' in l or 'clang-offload-bundler' in l
       or ('-cc1 -triple x86_64' in l and ('-emit-llvm-bc' in l or ' -S ' in l)) or '-cc1as -triple x86_64' in l]
assert len(sel) == 6, len(sel)
open('rebuild_from_s.sh', 'w').write('#!/bin/bash\nset -e\ncd "$(dirname "$0")"\n' + '\n'.join(sel) + '\n')
PY
chmod +x rebuild_from_s.sh
cp "$HERE/mk_variant.py" .
echo "ready: $T/st (dev_orig.s, rebuild_from_s.sh, mk_variant.py)"
