#!/usr/bin/env python3
"""BASELINE configs[4] at FULL size: 5 M noisy long reads (~2 kb, log-normal lengths, 5 % substitutions) = ~11 G bases.

  * stage-1 insert at k = 35 (no pairs, as populateGraph2 does for long reads: R/RNABloom.java:1311-1315) of the whole set,
    resident in HBM as packed batches of PIECE reads, into filters sized for nk = 0.6 x bases at FPR 0.01;
  * order-3 strobemers (k = 11, wMin = 12, wMax = 61: SeqSubsampler.strobemerBased, R/util/SeqSubsampler.java:360-367) and
    window minimizers (k = 13, w = 15) of every read, piece by piece, host ASCII in, results into host arrays that are reused;
  * property checks at full size (no false negatives, counts >= 1, strobemer / minimizer positions inside their windows and
    hashes equal to a recomputation from the returned positions on sampled reads) and an oracle comparison of a sample.

N > 1 ("1 -> 8 GPUs": hash-only work has no shared state, DESIGN.md s6 "replicas"): launched through torch.distributed.run,
every rank sketches the pieces p with p % world == rank on its own GPU; no collective but the final reduction of the timings.

    python tools/longread_full.py [reads=5000000 [piece=250000]]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/longread_full.py ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    sys.path.insert(0, p)
import numpy as np

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
dev = int(os.environ.get("LOCAL_RANK", "0"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
PIECE = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
K_INS, SK, SN, SWMIN, SWMAX, MK, MW = 35, 11, 3, 12, 61, 13, 15
ACGT = np.frombuffer(b"ACGT", np.uint8)
genome = ACGT[np.random.default_rng(1).integers(0, 4, 200_000_000, dtype=np.uint8)]


def piece(p):
    """reads [p * PIECE, ...): deterministic per piece, so that every rank of a multi-GPU run can make any piece"""
    rng = np.random.default_rng(1000 + p)
    m = min(PIECE, n - p * PIECE)
    lens = np.clip(rng.lognormal(np.log(2000), 0.5, m), 200, 12000).astype(np.int64)
    starts = rng.integers(0, genome.size - 12000, m)
    off = np.zeros(m + 1, np.int64); np.cumsum(lens, out=off[1:])
    seq = np.empty(int(off[-1]), np.uint8)
    ol, sl, ll = off.tolist(), starts.tolist(), lens.tolist()
    for i in range(m):
        seq[ol[i]:ol[i + 1]] = genome[sl[i]:sl[i] + ll[i]]
    pos = np.cumsum(rng.geometric(0.05, int(seq.size * 0.0525) + 1000)) - 1        # 5 % substitutions (a quarter of them silent)
    pos = pos[pos < seq.size]
    seq[pos] = ACGT[rng.integers(0, 4, pos.size, dtype=np.uint8)]
    return seq, off


pieces = list(range((n + PIECE - 1) // PIECE))
mine = [p for p in pieces if p % world == rank]
t_gen = time.perf_counter()
want = pieces if world == 1 else mine
if len(want) > 2:            # host-side generation is the slow part of this tool: worker processes, forked before any HIP call
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(len(want), 20)) as pool:
        data = dict(zip(want, pool.map(piece, want)))
else:
    data = {p: piece(p) for p in want}
if world > 1:
    import torch, torch.distributed as dist
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if os.environ.get("RB_BENCH_BACKEND") != "gloo" else "gloo")
from rnabloom import _native as N
from rnabloom import graph as G
bases = sum(int(o[-1]) for _, o in data.values())
if rank == 0:
    print("reads %d in %d pieces, %.2f G bases on this rank (generated in %.0f s)" % (n, len(pieces), bases / 1e9, time.perf_counter() - t_gen), flush=True)

# ---- stage-1 insert at k = 35 (single GPU leg; rank 0 only) ----
if rank == 0 and not os.environ.get("RB_LR_SKIP_INSERT"):
    tot = sum(int(o[-1]) for _, o in data.values())
    bits = N.lib.rb_expected_size(int(tot * 0.6), 0.01, 2)
    g = G.BloomFilterDeBruijnGraph(bits, bits, 0, 2, 2, 1, K_INS, False, False, device=dev, rngSeed=1)
    def joined(parts):       # several pieces as one batch: a call then spans several sub-batches, the producer of one beside the consumer of the one before
        if len(parts) == 1: return parts[0]
        seq = np.concatenate([s for s, _ in parts])
        base = np.cumsum([0] + [int(o[-1]) for _, o in parts[:-1]])
        off = np.concatenate([[0]] + [o[1:] + b for (_, o), b in zip(parts, base)]).astype(np.int64)
        return seq, off
    PER = int(os.environ.get("RB_LR_PIECES_PER_BATCH", "4"))
    vals_ = list(data.values())
    batches = []
    for i in range(0, len(vals_), PER):
        s_, o_ = joined(vals_[i:i + PER])
        batches.append(G.ReadBatch.from_ascii(s_, None, o_, 3, device=dev))
    for rep in range(3):                      # warm-up; timed as it runs (producer / consumer streams overlapped); timed stage by stage (HIP events serialise the streams)
        g.clearAllBf()
        if rep == 2: g.profileEnable(True)
        t0 = time.perf_counter()
        km = srt = 0
        for b in batches:
            st = g.addBatch(b); km += st.kmers; srt += st.sorted_kmers
        if rep == 1: dt_pipe = time.perf_counter() - t0
        dt = time.perf_counter() - t0
    prof = g.profileGet()
    print("k=%d insert, %d resident batches, filters %.1f + %.1f GB: %.3f s = %.2f G k-mers/s; stage by stage %.3f s, %.2f G k-mers/s (%d k-mers, %d grouped records)"
          % (K_INS, len(batches), bits / 8e9, bits / 1e9, dt_pipe, km / dt_pipe / 1e9, dt, km / dt / 1e9, km, srt), flush=True)
    print("  stages (ms): " + ", ".join("%s %.0f" % (k_, v[0]) for k_, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:14]), flush=True)
    # properties at full size: no false negatives, every inserted k-mer counts; occupancy below the configured rate
    for b in (batches[0], batches[-1]):
        h0 = b.nthash(K_INS, 1, first=0, n=2000)
        assert h0.size > 1_000_000 and bool(np.all(g.contains(h0))) and float(g.getCount(h0).min()) >= 1.0
    fpr = (g.getDbgbfFPR(), g.getCbfFPR())
    assert 0 < fpr[0] < 0.02 and 0 < fpr[1] < 0.02, fpr
    print("  no false negatives on 4000 sampled reads; FPR dbgbf %.4f cbf %.4f" % fpr, flush=True)
    # a second pass over the same reads: every k-mer is present now (the regime of a coverage > 1 library)
    g.profileEnable(False)
    t0 = time.perf_counter(); km2 = sum(g.addBatch(b).kmers for b in batches); dt2 = time.perf_counter() - t0
    print("  second pass over the same reads (all k-mers present): %.3f s, %.2f G k-mers/s" % (dt2, km2 / dt2 / 1e9), flush=True)
    if os.environ.get("RB_LR_THIRD_PASS"):        # and once more, stage by stage (HIP events serialise the streams)
        g.profileEnable(True); g.profileGet()
        t0 = time.perf_counter(); km3 = sum(g.addBatch(b).kmers for b in batches); dt3 = time.perf_counter() - t0
        prof = g.profileGet()
        print("  third pass, stage by stage: %.3f s, %.2f G k-mers/s; stages (ms): " % (dt3, km3 / dt3 / 1e9)
              + ", ".join("%s %.0f" % (k_, v[0]) for k_, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:14]), flush=True)
    del batches; g.destroy()

# ---- sketches of this rank's pieces ----
def sketch_all(what):
    out, tot_t, items = None, 0.0, 0
    for p in mine:
        seq, off = data[p]
        if what == "strobemers":
            if out is None:
                cap = int(max(int(o[-1]) for _, o in data.values()))
                out = (np.empty(cap, np.uint64), np.empty(cap, np.int32), np.empty(cap, np.int32))
            t0 = time.perf_counter(); so, h, s, e = G.strobemers((seq, off), SK, SN, SWMIN, SWMAX, device=dev, out=out); tot_t += time.perf_counter() - t0
            items += int(so[-1])
            if p == mine[0]: first = (so.copy(), h[:so[-1]].copy(), s[:so[-1]].copy(), e[:so[-1]].copy())
        else:
            if out is None:
                cap = int(max(int(o[-1]) for _, o in data.values()))
                out = (np.empty(cap, np.uint64), np.empty(cap, np.int64))
            t0 = time.perf_counter(); mo, h, pos = G.minimizers((seq, off), MK, MW, 1, device=dev, out=out); tot_t += time.perf_counter() - t0
            items += int(mo[-1])
            if p == mine[0]: first = (mo.copy(), h[:mo[-1]].copy(), pos[:mo[-1]].copy())
    return tot_t, items, first


my_bases = sum(int(data[p][1][-1]) for p in mine)
res = {}
for what in ("strobemers", "minimizers"):
    sketch_all(what) if len(mine) <= 2 else None        # warm-up on small runs (allocation, pinning)
    res[what] = sketch_all(what)

# checks on the first piece: positions inside their windows; hashes recomputed from the positions / compared with the oracle
seq, off = data[mine[0]]
so, sh, ss, se = res["strobemers"][2]
nk = np.maximum(off[1:] - off[:-1] - SK + 1, 0)
cnt = np.where(nk > SWMAX * (SN - 1), nk - SWMAX * (SN - 2) - SWMIN, 0)
assert np.array_equal(np.diff(so), cnt)
rd = np.repeat(np.arange(len(cnt)), cnt)
assert np.array_equal(ss, np.arange(int(so[-1])) - so[rd])                       # start = the strobemer's own k-mer
span = se - (SK - 1) - ss                                                       # last strobe's k-mer position - start
assert int(span.min()) >= SWMAX * (SN - 2) + SWMIN and int(span.max()) < SWMAX * (SN - 1)
mo, mh, mp = res["minimizers"][2]
wn = np.maximum(np.maximum(off[1:] - off[:-1] - MK + 1, 0) - MW + 1, 0)
assert np.array_equal(np.diff(mo), wn)
rdm = np.repeat(np.arange(len(wn)), wn)
pw = np.arange(int(mo[-1])) - mo[rdm]
assert bool(np.all((mp >= pw) & (mp < pw + MW)))
try:
    from oracle import rbo                                  # the checker (tools are not the product path)
    take = 300
    reads = [seq[off[i]:off[i + 1]].tobytes() for i in range(take)]
    o3 = [rbo.strobemers(r, SK, SN, SWMIN, SWMAX) for r in reads]
    for j, got in enumerate((sh, ss, se)):
        assert np.array_equal(got[:so[take]], np.concatenate([o[j] for o in o3]))
    o2 = [rbo.minimizers(r, MK, MW, 1) for r in reads]
    for j, got in enumerate((mh, mp)):
        assert np.array_equal(got[:mo[take]], np.concatenate([o[j] for o in o2]))
    checked = "first %d reads equal the oracle (strobemer hashes / starts / ends, minimizer hashes / positions)" % take
except ImportError:
    checked = "oracle not importable here: structural checks only"

line = {}
for what in ("strobemers", "minimizers"):
    t, items, _ = res[what]
    if world > 1:
        tt = torch.tensor([t], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX); t = float(tt.item())
        cc = torch.tensor([my_bases, items], dtype=torch.int64, device=tt.device)
        dist.all_reduce(cc); tb, items = int(cc[0].item()), int(cc[1].item())
    else:
        tb = my_bases
    line[what] = (t, tb, items)
if rank == 0:
    for what, (t, tb, items) in line.items():
        print("%s on %d GPU(s) (%s): %.2f s (slowest rank), %.2f G bases/s end to end (host ASCII in, results in host arrays), %d results"
              % (what, world, "k=%d n=%d w=[%d,%d]" % (SK, SN, SWMIN, SWMAX) if what == "strobemers" else "k=%d w=%d" % (MK, MW), t, tb / t / 1e9, items), flush=True)
    print("checks: counts per read, positions inside their windows; " + checked, flush=True)
if world > 1:
    dist.destroy_process_group()
