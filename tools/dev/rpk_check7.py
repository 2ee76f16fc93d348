import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    sys.path.insert(0, p)
import numpy as np
from oracle import rbo
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch
NREADS = 100000
bits = N.lib.rb_expected_size(45_000_000, 0.01, 2)      # sparse filter: a missing pair shows as a failed lookup
batch = ReadBatch.synthetic(NREADS, 64_000_000, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED, device=0)
seq, off = batch.download(0, NREADS)
g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, 25, False, True, device=0, rngSeed=1)
g.setReadPairedKmerDistance(115)
g.addBatch(batch, storeReadPairedKmers=True, first=0, n=NREADS)
bad_reads = []
for r in range(NREADS):
    s = seq[off[r]:off[r + 1]].tobytes()
    if b"N" in s: continue
    p, _, _ = rbo.hash_pairs_region(s, 25, 2, 115, 1)
    if p.shape[0] == 0: continue
    ok = g.lookupReadKmerPair(np.ascontiguousarray(p[:, 0]))
    if not ok.all():
        bad_reads.append((r, int((~ok).sum()), np.nonzero(~ok)[0][:12].tolist()))
print("reads with missing pairs:", len(bad_reads))
print(bad_reads[:30])
idx = np.array([b[0] for b in bad_reads])
if idx.size:
    print("min/max read", idx.min(), idx.max(), "histogram by 10k:", np.bincount(idx // 10000, minlength=10).tolist())
    print("read index mod 64 histogram:", np.bincount(idx % 64, minlength=64).tolist())
