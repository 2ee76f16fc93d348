"""Reproducer for the miscompilation of the general paired-k-mer kernel (k_pairs_insert, csrc/rb_graph.hip): the same
source with the `out_idx` choice as a RUN-TIME branch inside the roll loop sets ~20 % of the rpkbf bits of one 200 000-read
launch at wrong positions (differently from run to run) when compiled -O3 for gfx950 by hipcc 7.2; with the choice as a
template parameter (the shipped kernel) it is exact.  Build the diagnostic kernel first:
    make -C rna-bloom_amd clean && make -C rna-bloom_amd -j4 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRB_DIAG_PAIRS"
then (through gpurun, from the repo root):  python tools/pairs_variants.py
Measured (round 2): shipped kernel 0 wrong bits; RB_PAIRS_VARIANT=9 313 065 - 365 836 wrong bits of 920 943, two streams
or RB_SERIAL=1 alike; 0 wrong bits when the batch is cut into 3 M-k-mer sub-batches (smaller launches)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rna-bloom_amd")]
from rnabloom import _native as N                      # noqa: E402
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
batch = ReadBatch.synthetic(n // 2, 2_000_000, 150, 300, 30, 0.002, 1e-3, 2.0, seed=99, device=0)


def run_full(env, serial=False):
    """the stage-1 path: the pair kernel runs on the producer stream beside the consumer's kernels"""
    for k in ("RB_PAIRS_GENERAL", "RB_PAIRS_VARIANT", "RB_SERIAL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    if serial: os.environ["RB_SERIAL"] = "1"
    g = BloomFilterDeBruijnGraph(300_000_007, 300_000_007, 300_000_007, 2, 2, 2, 25, False, True, maxBatchKmers=int(os.environ.get("MAXB", "0")))
    g.setReadPairedKmerDistance(115)
    st = g.addBatch(batch, storeReadPairedKmers=True, first=0, n=n)
    out = g.exportFilter(N.RPKBF)
    g.destroy()
    return out, st.pairs


print("---- through rb_graph_add_batch (two streams) ----")
ref2, pairs2 = run_full({})
for name, env, serial in [("shipped general", {"RB_PAIRS_GENERAL": "1"}, False), ("run-time branch", {"RB_PAIRS_GENERAL": "1", "RB_PAIRS_VARIANT": "9"}, False),
                          ("run-time branch, serial", {"RB_PAIRS_GENERAL": "1", "RB_PAIRS_VARIANT": "9"}, True)]:
    for rep in range(2):
        got, p2 = run_full(env, serial)
        print("%-18s run %d: pairs %d, wrong bits %d (extra %d, missing %d)" % (name, rep, p2, int(np.unpackbits(got ^ ref2).sum()), int(np.unpackbits(got & ~ref2).sum()), int(np.unpackbits(ref2 & ~got).sum())))
