#!/usr/bin/env python3
"""Rate of the traversals on a SHARDED graph (rb_shard_trav_*, ShardRank.traverse) with G virtual ranks on one GPU: a scaled
config-2 graph, N seeds per call split evenly over the ranks.  Prints, per traversal, the wall time of the call for ALL ranks (the
ranks take turns on one GPU: per-rank time if they ran concurrently is about 1/G of the kernel share), the exchange rounds and
the time per round — the figure that matters on real hardware is rounds x (2 all-to-alls + 5 short kernels).
    python tools/sharded_walk_bench.py [ranks=4] [pairs=5000000] [seeds=200000] [bound=50]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    sys.path.insert(0, p)
import numpy as np
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch
from rnabloom.sharded import LoopbackCluster

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
n_seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
bound = int(sys.argv[4]) if len(sys.argv) > 4 else 50
nk = 450_000_000 * pairs // 50_000_000
bits = N.lib.rb_expected_size(nk, 0.01, 2)
batch = ReadBatch.synthetic(pairs, 64_000_000 * pairs // 50_000_000, seed=0x5EED)
g1 = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, 25, False, False, rngSeed=1)
g1.addBatch(batch, first=0, n=pairs)
cl = LoopbackCluster(G, bits, bits, bits, 2, 2, 2, 25, False, False, rngSeed=1)
cl.addBatch(batch, 150, first=0, n=pairs)
assert cl.fold(N.CBF) == g1.fold(N.CBF) and cl.fold(N.DBGBF) == g1.fold(N.DBGBF)
seq, off = batch.download(0, min(pairs, n_seeds))
rng = np.random.default_rng(1)
reads = rng.integers(0, off.size - 1, n_seeds); pos = rng.integers(0, 120, n_seeds)
seeds = [seq[off[r] + p: off[r] + p + 25].tobytes() for r, p in zip(reads, pos)]
cuts = [n_seeds * i // G for i in range(G + 1)]
per_rank = [seeds[cuts[i]:cuts[i + 1]] for i in range(G)]


def report(name, fn_single, fn_sharded, same):
    fn_single()
    t0 = time.perf_counter(); ref = fn_single(); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); got = fn_sharded(); t2 = time.perf_counter() - t0
    rounds = max(x[6] for x in got)
    steps = sum(int(x[4].sum()) for x in got)
    assert same(ref, got), name
    print("%s: %d walks over %d ranks, bound %d: %.3f s for all ranks (single GPU, filters local: %.3f s), %d exchange rounds = %.2f ms per round, "
          "%.2f M extension steps/s; equal to the single-GPU call" % (name, n_seeds, G, bound, t2, t1, rounds, 1e3 * t2 / max(rounds, 1), steps / t2 / 1e6))


def cat(got, j):
    return np.concatenate([x[j] for x in got])


for direction in (0, 1):
    report("max-coverage walk, direction %d" % direction,
           lambda: g1.walkMaxCov(seeds, direction, bound, 2.0, hashes=False),
           lambda: cl.traverse(0, per_rank, direction, bound=bound, min_cov=2.0),
           lambda ref, got: (cat(got, 4) == ref[4]).all() and (cat(got, 5) == ref[5]).all() and (cat(got, 0) == ref[0])[np.arange(bound)[None, :] < ref[4][:, None]].all())
    report("greedy extension, lookahead 5, direction %d" % direction,
           lambda: g1.greedyExtend(seeds, direction, 5, bound),
           lambda: cl.traverse(1, per_rank, direction, bound=bound, mode_or_lookahead=5, answer_cap=2048),
           lambda ref, got: (cat(got, 4) == ref[2]).all() and (cat(got, 5) == ref[3]).all() and (cat(got, 0) == ref[0])[np.arange(bound)[None, :] < ref[2][:, None]].all())
    report("naive extension (bounded), direction %d" % direction,
           lambda: g1.naiveExtend(seeds, direction, 1, bound=bound),
           lambda: cl.traverse(2, per_rank, direction, bound=bound, mode_or_lookahead=1),
           lambda ref, got: (cat(got, 5) == ref[1]).all() and [bytes(b[:l]) for x in got for b, l in zip(x[0], x[4])] == ref[0])
