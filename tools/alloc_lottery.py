#!/usr/bin/env python3
"""Does the probe stage's process-to-process variation (55 against 66 ms per step on one box) come with the ALLOCATION of
the filters?  One process, one read batch; the config-2 graph is created, filled once (one bench step), timed per stage, destroyed —
N times, and every other time with the previous graph's memory still held while the new one is allocated (different pages).
    python tools/alloc_lottery.py [rounds=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    sys.path.insert(0, p)
import torch
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pairs = 50_000_000
bits = N.lib.rb_expected_size(450_000_000, 0.01, 2)
batch = ReadBatch.synthetic(pairs, 64_000_000, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED)
held = None
for it in range(rounds):
    g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, 25, False, True, rngSeed=1)
    g.setReadPairedKmerDistance(115)
    if held is not None:
        held.destroy(); held = None
    res = []
    for step in range(2):
        g.clearAllBf()
        g.profileEnable(True); g.profileGet(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.addBatch(batch, storeReadPairedKmers=True, first=0, n=pairs)
        g.addBatch(batch, reverseComplement=True, storeReadPairedKmers=True, first=pairs, n=pairs)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pr = g.profileGet(True)
        res.append((dt * 1e3, pr["probe_claim"][0], pr["resolve_apply"][0], pr["filter_windows"][0]))
    import ctypes as C
    pr0, pr1 = C.c_float(), C.c_float()
    N.check(N.lib.rb_debug_probe_cbf(g.h, 0, C.byref(pr0))); N.check(N.lib.rb_debug_probe_cbf(g.h, 1, C.byref(pr1)))
    N.check(N.lib.rb_debug_probe_cbf(g.h, 0, C.byref(pr0))); N.check(N.lib.rb_debug_probe_cbf(g.h, 1, C.byref(pr1)))
    print("   2 x 2^28 random atomics on its counting filter: OR 0 %.2f ms, XOR pairs %.2f ms" % (pr0.value, pr1.value))
    print("allocation %d%s: step %.1f ms, probe_claim %.1f, resolve_apply %.1f, filter_windows %.1f  (first step %.1f / %.1f)"
          % (it, " (made while the one before was still held)" if it % 2 == 1 else "", res[1][0], res[1][1], res[1][2], res[1][3], res[0][0], res[0][1]), flush=True)
    if it % 2 == 0: held = g
    else: g.destroy()
if held is not None: held.destroy()
