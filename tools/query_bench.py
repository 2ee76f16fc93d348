#!/usr/bin/env python3
"""Rates of the lookup leg on the config-2 graph: getKmers (hashes + counts of every window of N reads, host strings in, host
arrays out) and getCount / contains on the hashes it returned.  Run under rocprofv3 --kernel-trace --stats for the kernels' own times.
    python tools/query_bench.py [pairs=50000000] [reads=2000000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    sys.path.insert(0, p)
import numpy as np
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
nk = 450_000_000 * pairs // 50_000_000
bits = N.lib.rb_expected_size(nk, 0.01, 2)
batch = ReadBatch.synthetic(pairs, 64_000_000 * pairs // 50_000_000, seed=0x5EED)
g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, 25, False, True, rngSeed=1)
g.setReadPairedKmerDistance(115)
g.addBatch(batch, storeReadPairedKmers=True, first=0, n=pairs)
g.addBatch(batch, reverseComplement=True, storeReadPairedKmers=True, first=pairs, n=pairs)
seq, off = batch.download(0, min(pairs, n_reads))
reads = [seq[off[i]:off[i + 1]].tobytes() for i in range(off.size - 1)]
g.getKmers(reads[:1000])
t0 = time.perf_counter()
ko, f, r, c = g.getKmers(reads)
dt = time.perf_counter() - t0
t0 = time.perf_counter()
g.getKmers(reads, out=(f, r, c))
dt2 = time.perf_counter() - t0
print("getKmers: %d reads, %d k-mers: %.3f s = %.2f G k-mers/s (host strings in, f/r/count out: %.1f GB over PCIe); mean count %.1f, zero counts %.2f %%; into arrays that exist already: %.3f s = %.2f G k-mers/s"
      % (len(reads), f.size, dt, f.size / dt / 1e9, (f.size * 20 + seq.size) / 1e9, c.mean(), 100.0 * (c == 0).mean(), dt2, f.size / dt2 / 1e9))
for name, fn in (("getCount", g.getCount), ("contains", g.contains)):
    fn(f[:1000])
    t0 = time.perf_counter()
    out = fn(f)
    dt = time.perf_counter() - t0
    print("%s: %d hashes: %.3f s = %.2f G/s" % (name, f.size, dt, f.size / dt / 1e9))
if hasattr(g, "batchCounts"):
    for nq in (n_reads, min(2 * pairs, 20_000_000)):
        g.batchCounts(batch, 0, 1000)
        t0 = time.perf_counter()
        cc = g.batchCounts(batch, 0, nq)
        dt = time.perf_counter() - t0
        print("batchCounts (resident batch, counts to host): %d reads, %d k-mers: %.3f s = %.2f G k-mers/s" % (nq, cc.size, dt, cc.size / dt / 1e9))
        if nq == n_reads:
            print("  equal to getKmers' counts:", bool(np.array_equal(cc, c)))
        t0 = time.perf_counter()
        g.batchCounts(batch, 0, nq, out=cc)
        dt = time.perf_counter() - t0
        print("  into an array that exists already: %.3f s = %.2f G k-mers/s" % (dt, cc.size / dt / 1e9))
        import torch
        dev = torch.empty(cc.size, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.batchCounts(batch, 0, nq, to_host=False, out=dev)
        dt = time.perf_counter() - t0
        print("batchCounts (counts left on the device, in a tensor that exists already): %.3f s = %.2f G k-mers/s" % (dt, cc.size / dt / 1e9))
        del dev
