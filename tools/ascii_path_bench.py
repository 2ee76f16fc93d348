#!/usr/bin/env python3
"""rb_graph_add_reads alone on config 2 sized input: host ASCII bases + qualities of N reads in one call (the FastqToGraphWorker boundary), three
repeats from cleared filters:  python tools/ascii_path_bench.py 50000000   (RB_HOST_TIMING=1: the library's own lines; RB_ASCII_CHUNKED=1: the chunk-by-chunk path)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rna-bloom_amd")]
import numpy as np
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch
pairs = int(sys.argv[1])
nk = 450_000_000 * pairs // 50_000_000
size = N.lib.rb_expected_size(nk, 0.01, 2)
b = ReadBatch.synthetic(pairs, 64_000_000 * pairs // 50_000_000, seed=0x5EED)
seq, off = b.download(0, pairs)
qual = np.full(seq.size, ord("I"), np.uint8)
g = BloomFilterDeBruijnGraph(size, size, size, 2, 2, 2, 25, False, True, rngSeed=1)
g.setReadPairedKmerDistance(115)
for rep in range(3):
    g.clearAllBf()
    t0 = time.perf_counter()
    st = g.addReads(seq, qual, off, 3, storeReadPairedKmers=True)
    dt = time.perf_counter() - t0
    print("addReads %.1f ms  %.2f G k-mers/s" % (dt * 1e3, st.kmers / dt / 1e9), flush=True)
