#!/usr/bin/env python3
"""Aggregate GPU work of the sharded engine for G ranks, measured on ONE GPU.

G virtual ranks (rnabloom.sharded.LoopbackCluster-style, one process, one device) run the whole
protocol one after another, so the wall time is approximately the SUM over ranks of the per-rank GPU work plus the
(device-local) exchange copies.  A real G-GPU run does the per-rank work concurrently: its step time
is about wall/G plus xGMI transfer time — this tool is how the sharded path is tuned without a
multi-GPU box.  Prints per-phase totals.

    python tools/loopback_bench.py --ranks 8 --pairs 8000000
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=8_000_000, help="total read pairs (split over the ranks)")
    ap.add_argument("--genome", type=int, default=64_000_000)
    ap.add_argument("--nk", type=int, default=450_000_000)
    ap.add_argument("--k", type=int, default=25)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-kmers", type=int, default=0, help="global sub-batch size in k-mers (default 2^30)")
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--native", action="store_true", help="exchange driver below the C ABI (rb_shard_add_range, one host thread per rank: the ranks' kernels overlap on the one GPU)")
    a = ap.parse_args()
    import torch
    torch.cuda.set_device(0)
    from rnabloom import _native as N
    from rnabloom import sharded
    from rnabloom.graph import ReadBatch

    G, k = a.ranks, a.k
    sz = N.lib.rb_expected_size(a.nk, 0.01, 2)
    ranks = [sharded.ShardRank((sz, sz, sz, 2, 2, 2, k, 0, 1, 0, 0, 1, a.batch_kmers), r, G, 0) for r in range(G)]
    batch = ReadBatch.synthetic(a.pairs, a.genome, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED, device=0)   # shared by the ranks
    for r in ranks:
        r.set_read_pair_distance(max(1, 150 - k - 10))
    pos_bits, rps = sharded.plan(150, k, G, a.batch_kmers or sharded.default_batch_kmers(G, ranks[0].mode))

    comm = sharded.NativeComm.loopback(G) if a.native else None

    def step():
        for r in ranks:
            r.clear()
        for first, fl in ((0, N.ADD_STORE_READ_PAIRS), (a.pairs, N.ADD_STORE_READ_PAIRS | N.ADD_REVCOMP)):
            if comm is not None:
                sharded.run_native_loopback(ranks, comm, batch, first, a.pairs, fl, rps, pos_bits)
            else:
                sharded.run_loopback([r.add_range(batch, first, a.pairs, fl, rps, pos_bits) for r in ranks])

    for _ in range(a.warmup):
        step()
    for r in ranks:
        for kk in r.stats:
            r.stats[kk] = 0
    if a.trace:
        sharded.TRACE = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sharded._t_last[0] = t0
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    tot = {kk: sum(r.stats[kk] for r in ranks) // a.steps for kk in ranks[0].stats}
    out = {"ranks": G, "driver": "native (threads overlap on one GPU)" if a.native else "python", "pairs": a.pairs, "reads_per_substep": rps, "wall_ms_per_step": round(dt * 1e3, 1),
           "per_rank_ms_if_concurrent": round(dt * 1e3 / G, 1),
           "projected_kmers_per_s_without_comm": round(tot["kmers"] / (dt / G)), "stats": tot}
    if a.trace:
        out["phase_ms_per_step_all_ranks"] = {kk: round(v / a.steps, 1) for kk, v in sorted(sharded.TRACE.items(), key=lambda kv: -kv[1])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
