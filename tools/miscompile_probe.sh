#!/bin/bash
# The k_pairs_insert miscompile (HISTORY s5, profiles/r05_miscompile.md): rebuild csrc/rb_graph.hip with -DRB_DIAG_PAIRS under a few code-generation settings
# and count the wrong bits of the run-time-branch form (tools/pairs_variants.py) under each.  Run through gpurun from the repo root; nothing outside the
# box's copy changes.  TREE=<dir>: probe another checkout of this repository instead (the failure lives in the round-2 tree: git archive 306b610 into
# gpurun_scratch/r02 — today's reproducer is exact since round 3 made the one-bit rotations funnel shifts).
R=${TREE:-$GRAFT_REPO_ROOT}; cd $R/rna-bloom_amd
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DRB_DIAG_PAIRS"
try() {  # name, extra flags...
  local name=$1; shift
  rm -f build/rb_graph.o
  make -s CXXFLAGS="${BASE/-O3/${OPT:--O3}} $*" lib/librb_hip.so > /dev/null 2>&1 || { echo "$name: build failed"; return; }
  echo "== $name   [${OPT:--O3} $*]"
  ( cd $R && timeout 600 python tools/pairs_variants.py 2>&1 | grep -E "run-time branch|shipped" | head -6 )
}
try baseline
try waitcnt_forcezero -mllvm -amdgpu-waitcnt-forcezero=1
try no_scalar_global_loads -mllvm -amdgpu-scalarize-global-loads=0
OPT=-O1 try O1
OPT=-O2 try O2
try no_atomic_optimizer -mllvm -amdgpu-atomic-optimizer-strategy=None
