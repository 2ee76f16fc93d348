import sys, time, os
sys.path[:0] = ["/root/repo", "/root/repo/rna-bloom_amd"]
import numpy as np
from rnabloom.graph import ReadBatch, PackedStream
b = ReadBatch.synthetic(12_500_000, 64_000_000, seed=1)
ph = b.downloadPacked(0, 25_000_000)
ps = PackedStream(12_500_000, 12_500_000 * 5)
for rep in range(3):
    for a in (0, 12_500_000):
        t0 = time.perf_counter(); ps.begin(ph, a, 12_500_000); ps.finish(); print("chunk %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
