#!/usr/bin/env python3
"""What a batch of reads of DIFFERENT lengths costs against the uniform batches of the bench: the same 20 M synthetic reads inserted as
they are (150 bases each: one read per lane in the prefilter) and trimmed at random to 101..150 bases (4 or 5 packed words per read).
    python tools/ragged_bench.py [reads=20000000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")): sys.path.insert(0, p)
import numpy as np, torch
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
src = ReadBatch.synthetic(n, 64_000_000 * n // 50_000_000, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED)
seq, off = src.download(0, n)
rng = np.random.default_rng(5)
lens = np.where(rng.random(n) < 0.7, 150, rng.integers(101, 151, n)).astype(np.int64)       # 70 % untouched, the rest trimmed
keep = (np.arange(150)[None, :] < lens[:, None])
rag_seq = seq.reshape(n, 150)[keep]
rag_off = np.zeros(n + 1, np.int64); np.cumsum(lens, out=rag_off[1:])
uni = ReadBatch.from_ascii(seq, None, off, 3)
rag = ReadBatch.from_ascii(rag_seq, None, rag_off, 3)
bits = N.lib.rb_expected_size(450_000_000 * n // 100_000_000, 0.01, 2)
for name, b, env in (("uniform", uni, {}), ("ragged", rag, {}), ("ragged, RB_READ_LANES=0", rag, {"RB_READ_LANES": "0"}), ("uniform, RB_READ_LANES=0", uni, {"RB_READ_LANES": "0"})):
    for k_, v in env.items(): os.environ[k_] = v
    g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, 25, False, True, rngSeed=1); g.setReadPairedKmerDistance(115)
    best = None
    for rep in range(3):
        g.clearAllBf(); g.profileEnable(True); g.profileGet(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = g.addBatch(b, storeReadPairedKmers=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pr = g.profileGet(True)
        if best is None or dt < best[0]: best = (dt, st.kmers, {k2: round(v2[0], 1) for k2, v2 in pr.items() if v2[0] > 3})
    print("%s: %.1f ms, %.2f G k-mers/s, stages %s" % (name, best[0] * 1e3, best[1] / best[0] / 1e9, best[2]), flush=True)
    g.destroy()
    for k_ in env: os.environ.pop(k_, None)
