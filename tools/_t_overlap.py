import sys, time, os, threading
sys.path[:0] = ["/root/repo", "/root/repo/rna-bloom_amd"]
import numpy as np
from rnabloom import _native as N
from rnabloom.graph import ReadBatch, PackedStream, BloomFilterDeBruijnGraph
pairs = 50_000_000
sz = N.lib.rb_expected_size(450_000_000, 0.01, 2)
b = ReadBatch.synthetic(pairs, 64_000_000, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED)
g = BloomFilterDeBruijnGraph(sz, sz, sz, 2, 2, 2, 25, False, True, rngSeed=1)
g.setReadPairedKmerDistance(115)
ph = b.downloadPacked(0, 25_000_000)
ps = PackedStream(12_500_000, 62_500_000)
def step():
    g.clearAllBf()
    g.addBatch(b, storeReadPairedKmers=True, first=0, n=pairs)
    g.addBatch(b, reverseComplement=True, storeReadPairedKmers=True, first=pairs, n=pairs)
step()
for rep in range(2):
    t0 = time.perf_counter(); step(); print("resident alone %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
stop = [False]; up = [0, 0.0]
def uploader():
    while not stop[0]:
        t0 = time.perf_counter()
        ps.begin(ph, 0, 12_500_000); ps.finish()
        up[0] += 1; up[1] += time.perf_counter() - t0
for rep in range(2):
    stop[0] = False; up[:] = [0, 0.0]
    th = threading.Thread(target=uploader); th.start()
    time.sleep(0.05)
    t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
    stop[0] = True; th.join()
    print("resident with uploads beside it %.1f ms; %d chunks of 0.8 GB, %.1f ms each = %.1f GB/s" % (dt * 1e3, up[0], up[1] / max(up[0], 1) * 1e3, 0.8 * up[0] / up[1]), flush=True)
