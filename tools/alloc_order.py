#!/usr/bin/env python3
"""One config-2 bench step in a fresh process, with the graph's allocations made in different orders (see tools/alloc_lottery.py:
the probe stages' time comes with the allocation).  mode: plain (batch, then graph: what bench.py does), precreate (a graph is
created and destroyed before the one that is used), prefill (... created, filled once and destroyed), dummy (60 GB allocated, written
and freed before), graph_first (graph before the batch).    python tools/alloc_order.py <mode>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")): sys.path.insert(0, p)
import torch
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch
mode = sys.argv[1]
pairs = 50_000_000
bits = N.lib.rb_expected_size(450_000_000, 0.01, 2)
def make():
    g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, 25, False, True, rngSeed=1); g.setReadPairedKmerDistance(115); return g
def fill(g, batch, timed=True):
    g.clearAllBf(); g.profileEnable(True); g.profileGet(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.addBatch(batch, storeReadPairedKmers=True, first=0, n=pairs)
    g.addBatch(batch, reverseComplement=True, storeReadPairedKmers=True, first=pairs, n=pairs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pr = g.profileGet(True); return dt * 1e3, pr["probe_claim"][0], pr["resolve_apply"][0]
if mode == "graph_first":
    g = make(); batch = ReadBatch.synthetic(pairs, 64_000_000, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED)
else:
    batch = ReadBatch.synthetic(pairs, 64_000_000, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED)
    if mode == "precreate": make().destroy()
    if mode == "prefill": g0 = make(); fill(g0, batch); g0.destroy()
    if mode == "dummy":
        x = torch.empty(60 << 30, dtype=torch.uint8, device="cuda"); x.zero_(); torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
    g = make()
fill(g, batch)
r = fill(g, batch)
print(mode, "step %.1f ms probe_claim %.1f resolve_apply %.1f" % r, flush=True)
