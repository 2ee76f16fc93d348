import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    sys.path.insert(0, p)
import numpy as np
from rnabloom import _native as N
from rnabloom import sharded
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch
PAIRS = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
bits = N.lib.rb_expected_size(450_000_000, 0.01, 2)
batch = ReadBatch.synthetic(PAIRS, 64_000_000, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED, device=0)
def run(tag, mb, env):
    for k, v in env.items(): os.environ[k] = v
    g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, 25, False, True, device=0, rngSeed=1, maxBatchKmers=mb)
    g.setReadPairedKmerDistance(115)
    s1 = g.addBatch(batch, storeReadPairedKmers=True, first=0, n=PAIRS)
    p1 = g.popcount(N.RPKBF)
    s2 = g.addBatch(batch, reverseComplement=True, storeReadPairedKmers=True, first=PAIRS, n=PAIRS)
    print(tag, "pairs", s1.pairs, s2.pairs, "rpk popcount after file 1:", p1, "after file 2:", g.popcount(N.RPKBF), "dbg", g.popcount(N.DBGBF), flush=True)
    for k in env: del os.environ[k]
    g.destroy()
run("default 2^30", 0, {})
run("2^28 noramp", 1 << 28, {"RB_NO_RAMP": "1"})
run("2^28 ramp", 1 << 28, {})
run("2^29", 1 << 29, {})
run("2^30 noramp", 0, {"RB_NO_RAMP": "1"})
run("2^30 nompf", 0, {"RB_NO_MPF": "1"})
