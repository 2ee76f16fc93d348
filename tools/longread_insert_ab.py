#!/usr/bin/env python3
"""The k = 35 stage-1 insert of configs[4]'s long reads (all-new-k-mers regime: at 5 % error nearly every 35-mer is seen once),
A/B over environment switches in ONE process on ONE set of resident batches: for each variant the graph is rebuilt, the reads inserted
three times (warm-up; timed with the streams overlapped; timed stage by stage with HIP events), and the filters' digests compared with the first variant's.

    python tools/longread_insert_ab.py [reads=1000000] [VAR=val[,VAR=val]] [VAR=val] ...        (the empty variant = defaults runs first)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    sys.path.insert(0, p)
import numpy as np
from rnabloom import _native as N
from rnabloom import graph as G

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
variants = [""] + sys.argv[2:]
PIECE, K = 250_000, 35
ACGT = np.frombuffer(b"ACGT", np.uint8)
genome = ACGT[np.random.default_rng(1).integers(0, 4, 200_000_000, dtype=np.uint8)]


def piece(p):
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
with mp.get_context("fork").Pool(min(len(pieces), 16)) as pool:
    data = pool.map(piece, pieces)
bases = sum(int(o[-1]) for _, o in data)
bits = N.lib.rb_expected_size(int(bases * 0.6), 0.01, 2)
def joined(parts):
    """several pieces as one batch: a call then spans several sub-batches, and the producer of one runs beside the consumer of the one before"""
    if len(parts) == 1: return parts[0]
    seq = np.concatenate([s for s, _ in parts])
    off = np.concatenate([[0]] + [o[1:] + b for (_, o), b in zip(parts, np.cumsum([0] + [int(o[-1]) for _, o in parts[:-1]]))]).astype(np.int64)
    return seq, off
PER = int(os.environ.get("RB_LR_PIECES_PER_BATCH", "4"))
batches = []
for i in range(0, len(data), PER):
    s_, o_ = joined(data[i:i + PER])
    batches.append(G.ReadBatch.from_ascii(s_, None, o_, 3, device=0))
print("reads %d, %.2f G bases, filters %.1f + %.1f GB" % (n, bases / 1e9, bits / 8e9, bits / 1e9), flush=True)
ref = None
for var in variants:
    kv = dict(x.split("=", 1) for x in var.split(",") if x)
    for a, b in kv.items(): os.environ[a] = b
    g = G.BloomFilterDeBruijnGraph(bits, bits, 0, 2, 2, 1, K, False, False, device=0, rngSeed=1)
    for rep in range(3):                      # warm-up, timed as it runs (producer and consumer streams overlapped), then stage by stage (serialised by the events)
        g.clearAllBf()
        if rep == 2: g.profileEnable(True); g.profileGet(True)
        t0 = time.perf_counter()
        km = srt = dis = 0
        for b in batches:
            st = g.addBatch(b); km += st.kmers; srt += st.sorted_kmers; dis += st.distinct
        if rep == 1: dt_pipe = time.perf_counter() - t0
        dt = time.perf_counter() - t0
    prof = g.profileGet()
    dig = (g.popcount(N.DBGBF), g.popcount(N.CBF), g.fold(N.DBGBF), g.fold(N.CBF))
    if ref is None: ref = dig
    print("[%s] %.3f s = %.2f G k-mers/s overlapped; stage by stage %.3f s, %.2f G k-mers/s (%d k-mers, %d records, %d runs)%s" % (var or "defaults", dt_pipe, km / dt_pipe / 1e9, dt, km / dt / 1e9, km, srt, dis, "" if dig == ref else "   FILTERS DIFFER FROM THE FIRST VARIANT"), flush=True)
    print("    stages (ms): " + ", ".join("%s %.0f" % (k_, v[0]) for k_, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]), flush=True)
    g.destroy()
    for a in kv: os.environ.pop(a, None)
