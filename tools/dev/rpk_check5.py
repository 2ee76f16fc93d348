import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, os.environ.get("RB_TREE", "rna-bloom_amd"))):
    sys.path.insert(0, p)
import numpy as np
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch
PAIRS = int(sys.argv[1])
bits = N.lib.rb_expected_size(int(sys.argv[2]), 0.01, 2)
batch = ReadBatch.synthetic(PAIRS, 64_000_000, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED, device=0)
def mk():
    g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, 25, False, True, device=0, rngSeed=1)
    g.setReadPairedKmerDistance(115)
    return g
g = mk()
g.addBatch(batch, storeReadPairedKmers=False, first=0, n=PAIRS)
print("no pairs requested: rpk popcount", g.popcount(N.RPKBF))
# pairs only via the pair-hash API: compute pair hashes? not available; instead insert the same reads in two graphs with k-mer path identical
os.environ["RB_DEBUG_PAIRS_ONLY"] = "1"
ga = mk(); sa = ga.addBatch(batch, storeReadPairedKmers=True, first=0, n=1000)
gb = mk(); sb = gb.addBatch(batch, storeReadPairedKmers=True, first=0, n=1000)
print("1000 reads:", sa.pairs, sb.pairs, ga.popcount(N.RPKBF), gb.popcount(N.RPKBF), np.array_equal(ga.exportFilter(N.RPKBF), gb.exportFilter(N.RPKBF)))
for n in (3000, 10000, 30000, 100000):
    ga = mk(); sa = ga.addBatch(batch, storeReadPairedKmers=True, first=0, n=n)
    gb = mk(); sb = gb.addBatch(batch, storeReadPairedKmers=True, first=0, n=n)
    print(n, "reads:", sa.pairs, sb.pairs, ga.popcount(N.RPKBF), gb.popcount(N.RPKBF), np.array_equal(ga.exportFilter(N.RPKBF), gb.exportFilter(N.RPKBF)), flush=True)
    ga.destroy(); gb.destroy()
