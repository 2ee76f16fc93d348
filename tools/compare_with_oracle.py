"""Development aid: insert N synthetic reads on the GPU three times and compare the three filters with the CPU oracle
(missing / extra bits of the read-pair filter, equality of dbgbf and cbf).  usage: compare_with_oracle.py N [first [total]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, os.environ.get("RB_TREE", "rna-bloom_amd"))):
    sys.path.insert(0, p)
import numpy as np
from oracle import rbo
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch
NREADS = int(sys.argv[1]); FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0; TOTAL = int(sys.argv[3]) if len(sys.argv) > 3 else NREADS
bits = N.lib.rb_expected_size(4_500_000, 0.01, 2)
batch = ReadBatch.synthetic(TOTAL, 64_000_000, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED, device=0)
seq, off = batch.download(FIRST, NREADS)
og = rbo.Graph(bits, bits, bits, 2, 2, 2, 25, False, True, 1)
og.set_read_pair_distance(115)
og.add_reads(seq, None, off, 3, rbo.STORE_READ_PAIRS)
ref = og.rpkbf_bytes()
def pc(a): return int(np.unpackbits(a).sum())
print("oracle rpk popcount", pc(ref))
for rep in range(3):
    g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, 25, False, True, device=0, rngSeed=1)
    g.setReadPairedKmerDistance(115)
    g.addBatch(batch, storeReadPairedKmers=True, first=FIRST, n=NREADS)
    r = g.exportFilter(N.RPKBF)
    print(rep, "gpu popcount", pc(r), "missing vs oracle", pc(ref & ~r), "extra vs oracle", pc(r & ~ref),
          "dbg equal", np.array_equal(g.exportFilter(N.DBGBF), og.dbgbf_bytes()), "cbf equal", np.array_equal(g.exportFilter(N.CBF), og.cbf_bytes()), flush=True)
    g.destroy()
