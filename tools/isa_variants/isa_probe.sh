#!/bin/bash
# run through gpurun from the repo root:  tools/isa_variants/isa_probe.sh <tree relative to the repo root> <variant> [<variant> ...]
# variants (mk_variant.py): none | valu_all | valu_all2 | valu_sgpr | salu_all | vmem_all | all | mad64 | mul32 | cmp64 | sh64 | add64 | cnd | co | xor_or
#                           | vgpr<N> | sgpr<N> | accum<N> (kernel-descriptor only, '+' joins several: vgpr64+accum64)
T=$GRAFT_REPO_ROOT/$1; shift
cd $T/rna-bloom_amd
make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DRB_DIAG_PAIRS" > /dev/null 2>&1
for v in "$@"; do
  ( cd st && python3 mk_variant.py $v && ./rebuild_from_s.sh > /dev/null 2>&1 ) || { echo "$v: rebuild failed"; continue; }
  touch build/rb_graph.o
  make -s lib/librb_hip.so > /dev/null 2>&1 || { echo "$v: link failed"; continue; }
  ( cd .. && timeout 600 python3 tools/pairs_variants.py 2>&1 | grep -E "run-time branch " | head -2 )
done
