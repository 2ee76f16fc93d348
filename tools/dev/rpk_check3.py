import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    sys.path.insert(0, p)
import numpy as np
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch
PAIRS = int(sys.argv[1])
bits = N.lib.rb_expected_size(int(sys.argv[2]), 0.01, 2)
batch = ReadBatch.synthetic(PAIRS, 64_000_000, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED, device=0)
outs = []
for rep in range(2):
    g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, 25, False, True, device=0, rngSeed=1)
    g.setReadPairedKmerDistance(115)
    s1 = g.addBatch(batch, storeReadPairedKmers=True, first=0, n=PAIRS)
    outs.append((g.exportFilter(N.RPKBF), s1.pairs, g.popcount(N.RPKBF)))
    g.destroy()
print(sys.argv[1:], os.environ.get("RB_SERIAL"), "pairs", outs[0][1], outs[1][1], "pop", outs[0][2], outs[1][2], "equal:", np.array_equal(outs[0][0], outs[1][0]), flush=True)
