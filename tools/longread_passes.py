#!/usr/bin/env python3
"""configs[4]'s k = 35 insert, pass by pass: the reads are inserted P times into one graph (pass 1: nearly every k-mer new; pass 2 on: every k-mer
re-sighted), each pass timed as it runs and — RB_LR_STAGES=1 — stage by stage (HIP events serialise the streams).  Small enough to sit under
rocprofv3 twice (P = 1 and P = 2: the difference of the two kernel summaries is the re-sighting pass, tools/diff_kernel_stats.py).

    python tools/longread_passes.py [reads=1500000] [passes=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    sys.path.insert(0, p)
import numpy as np
from rnabloom import _native as N
from rnabloom import graph as G

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
PIECE, K = 250_000, 35
ACGT = np.frombuffer(b"ACGT", np.uint8)
genome = ACGT[np.random.default_rng(1).integers(0, 4, 200_000_000, dtype=np.uint8)]


def piece(p):                                   # (the reads of tools/longread_full.py)
    rng = np.random.default_rng(1000 + p)
    m = min(PIECE, n - p * PIECE)
    lens = np.clip(rng.lognormal(np.log(2000), 0.5, m), 200, 12000).astype(np.int64)
    starts = rng.integers(0, genome.size - 12000, m)
    off = np.zeros(m + 1, np.int64); np.cumsum(lens, out=off[1:])
    seq = np.empty(int(off[-1]), np.uint8)
    ol, sl, ll = off.tolist(), starts.tolist(), lens.tolist()
    for i in range(m):
        seq[ol[i]:ol[i + 1]] = genome[sl[i]:sl[i] + ll[i]]
    pos = np.cumsum(rng.geometric(0.05, int(seq.size * 0.0525) + 1000)) - 1
    pos = pos[pos < seq.size]
    seq[pos] = ACGT[rng.integers(0, 4, pos.size, dtype=np.uint8)]
    return seq, off


pieces = list(range((n + PIECE - 1) // PIECE))
import multiprocessing as mp
with mp.get_context("fork").Pool(min(len(pieces), 20)) as pool:
    data = pool.map(piece, pieces)
bases = sum(int(o[-1]) for _, o in data)
bits = N.lib.rb_expected_size(int(bases * 0.6), 0.01, 2)


def joined(parts):
    if len(parts) == 1: return parts[0]
    seq = np.concatenate([s for s, _ in parts])
    off = np.concatenate([[0]] + [o[1:] + b for (_, o), b in zip(parts, np.cumsum([0] + [int(o[-1]) for _, o in parts[:-1]]))]).astype(np.int64)
    return seq, off


batches = []
for i in range(0, len(data), 4):
    s_, o_ = joined(data[i:i + 4])
    batches.append(G.ReadBatch.from_ascii(s_, None, o_, 3, device=0))
del data
print("reads %d, %.2f G bases, filters %.1f + %.1f GB" % (n, bases / 1e9, bits / 8e9, bits / 1e9), flush=True)
g = G.BloomFilterDeBruijnGraph(bits, bits, 0, 2, 2, 1, K, False, False, device=0, rngSeed=1)
stages = bool(os.environ.get("RB_LR_STAGES"))
for b in batches[:1]:                            # scratch allocation outside the timed passes
    g.addBatch(b)
g.clearAllBf()
for ps in range(P):
    if stages: g.profileEnable(True); g.profileGet(True)
    t0 = time.perf_counter()
    km = dis = srt = 0
    for b in batches:
        st = g.addBatch(b); km += st.kmers; dis += st.distinct; srt += st.sorted_kmers
    dt = time.perf_counter() - t0
    print("pass %d: %.3f s = %.2f G k-mers/s (%d k-mers, %d records, %d runs, %d conflict ops)%s" % (ps + 1, dt, km / dt / 1e9, km, srt, dis, st.conflict_ops, " [stage by stage]" if stages else ""), flush=True)
    if stages:
        prof = g.profileGet()
        print("    stages (ms): " + ", ".join("%s %.0f" % (k_, v[0]) for k_, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:16]), flush=True)
print("digest: popcounts %d %d folds %x %x" % (g.popcount(N.DBGBF), g.popcount(N.CBF), g.fold(N.DBGBF), g.fold(N.CBF)), flush=True)
