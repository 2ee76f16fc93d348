#!/usr/bin/env python3
"""Rate of rb_graph_walk (batched greedy maximum-coverage walks) on the config-2 graph: N seeds taken from the reads,
walked to the right for up to BOUND steps with minKmerCov 2.  Prints walks/s and extension steps/s (one step = the four
graph.getCount of Kmer.getMaxCovSuccessor = 16 Bloom probes) including the host<->device copies of the call.
    python tools/walk_bench.py [pairs=50000000] [seeds=2000000] [bound=100]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    sys.path.insert(0, p)
import numpy as np
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
bound = int(sys.argv[3]) if len(sys.argv) > 3 else 100
nk = 450_000_000 * pairs // 50_000_000
bits = N.lib.rb_expected_size(nk, 0.01, 2)
batch = ReadBatch.synthetic(pairs, 64_000_000 * pairs // 50_000_000, seed=0x5EED)
g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, 25, False, True, rngSeed=1)
g.setReadPairedKmerDistance(115)
g.addBatch(batch, storeReadPairedKmers=True, first=0, n=pairs)
g.addBatch(batch, reverseComplement=True, storeReadPairedKmers=True, first=pairs, n=pairs)
seq, off = batch.download(0, min(pairs, n_seeds))
rng = np.random.default_rng(1)
reads = rng.integers(0, off.size - 1, n_seeds); pos = rng.integers(0, 120, n_seeds)
seeds = [seq[off[r] + p: off[r] + p + 25].tobytes() for r, p in zip(reads, pos)]
for direction in (0, 1):
    g.walkMaxCov(seeds[:1000], direction, bound, 2.0)
    t0 = time.perf_counter()
    bases, f, r, c, ln, reason = g.walkMaxCov(seeds, direction, bound, 2.0, hashes=False)     # bases + counts + lengths come back
    dt = time.perf_counter() - t0
    steps = int(ln.sum()) + int((reason != 3).sum())          # the step that ended a walk was evaluated too
    print("direction %d: %d walks, bound %d: %.3f s = %.2f M walks/s, %.1f M extension steps/s (%.0f M getCount/s); mean length %.1f; reasons %s"
          % (direction, n_seeds, bound, dt, n_seeds / dt / 1e6, steps / dt / 1e6, 4 * steps / dt / 1e6, ln.mean(),
             dict(zip(*np.unique(reason, return_counts=True)))))
for direction in (0, 1):
    t0 = time.perf_counter()
    bases, c, ln, reason = g.greedyExtend(seeds, direction, 5, bound)
    dt = time.perf_counter() - t0
    print("greedy extension, lookahead 5, direction %d: %d walks, bound %d: %.3f s = %.2f M walks/s, %.1f M extension steps/s; mean length %.1f"
          % (direction, n_seeds, bound, dt, n_seeds / dt / 1e6, int(ln.sum()) / dt / 1e6, ln.mean()))
