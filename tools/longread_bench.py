#!/usr/bin/env python3
"""BASELINE config 5 shape at reduced size: N noisy long reads (~2 kb, 5 % substitutions, k = 35, no pairs) — stage-1
insert of the resident batch, minimizers (m = 13, w = 15) and order-3 strobemers (k = 11, wMin = 12, wMax = 61) of the
same reads from host ASCII.  Prints rates.      python tools/longread_bench.py [reads=500000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    sys.path.insert(0, p)
import numpy as np
from rnabloom import _native as N
from rnabloom import graph as G

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
rng = np.random.default_rng(1)
genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 20_000_000)]
lens = np.clip(rng.lognormal(np.log(2000), 0.5, n), 200, 12000).astype(np.int64)
starts = rng.integers(0, genome.size - 12000, n)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
seq = np.empty(int(off[-1]), np.uint8)
for i in range(n):
    seq[off[i]:off[i + 1]] = genome[starts[i]:starts[i] + lens[i]]
err = rng.random(seq.size) < 0.05
seq[err] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(err.sum()))]
print("reads %d, bases %.2f G" % (n, seq.size / 1e9), flush=True)
nk = int(seq.size * 0.6)
bits = N.lib.rb_expected_size(nk, 0.01, 2)
g = G.BloomFilterDeBruijnGraph(bits, bits, 0, 2, 2, 1, 35, False, False, rngSeed=1)
batch = G.ReadBatch.from_ascii(seq, None, off, 3)
for rep in range(2):
    g.clearAllBf()
    t0 = time.perf_counter(); st = g.addBatch(batch); dt = time.perf_counter() - t0
print("k=35 insert (resident batch): %.3f s, %.2f G k-mers/s (%d k-mers, %d sorted)" % (dt, st.kmers / dt / 1e9, st.kmers, st.sorted_kmers), flush=True)
reads = [seq[off[i]:off[i + 1]].tobytes() for i in range(min(n, 200_000))]
nb = sum(len(r) for r in reads)
for rep in range(2):
    t0 = time.perf_counter(); mo, h, p = G.minimizers(reads, 13, 15, 1); dt = time.perf_counter() - t0
nr = len(reads)
arr = (np.ascontiguousarray(seq[:off[nr]]), np.ascontiguousarray(off[:nr + 1]))
t0 = time.perf_counter(); G.minimizers(arr, 13, 15, 1, out=(h, p)); dt2 = time.perf_counter() - t0
print("minimizers m=13 w=15 (host ASCII in, hashes out): %.3f s, %.2f G bases/s, %d minimizer windows; from packed arrays into arrays that exist already: %.3f s, %.2f G bases/s"
      % (dt, nb / dt / 1e9, int(mo[-1]), dt2, nb / dt2 / 1e9), flush=True)
for rep in range(2):
    t0 = time.perf_counter(); so, h, s, e = G.strobemers(reads, 11, 3, 12, 61); dt = time.perf_counter() - t0
t0 = time.perf_counter(); G.strobemers(arr, 11, 3, 12, 61, out=(h, s, e)); dt2 = time.perf_counter() - t0
print("strobemers k=11 n=3 w=[12,61]: %.3f s, %.2f G bases/s, %d strobemers; from packed arrays into arrays that exist already: %.3f s, %.2f G bases/s"
      % (dt, nb / dt / 1e9, int(so[-1]), dt2, nb / dt2 / 1e9), flush=True)
