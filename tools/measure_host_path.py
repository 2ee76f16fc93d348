#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in boundary: rb_graph_add_reads() handed HOST ASCII buffers
(upload + GPU-side 2-bit encode + insert), vs the resident-batch rate bench.py reports."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rna-bloom_amd")]
import numpy as np
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nk = 450_000_000 * pairs // 50_000_000
size = N.lib.rb_expected_size(nk, 0.01, 2)
b = ReadBatch.synthetic(pairs, 64_000_000 * pairs // 50_000_000, seed=0x5EED)
seq, off = b.download(0, pairs)                     # left reads as host ASCII ('N' where unusable)
qual = np.full(seq.size, ord("I"), np.uint8)
g = BloomFilterDeBruijnGraph(size, size, size, 2, 2, 2, 25, False, True, rngSeed=1)
g.setReadPairedKmerDistance(115)
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else pairs   # reads per rb_graph_add_reads call (the library pipelines inside a call)
for rep in range(2):
    g.clearAllBf()
    t0 = time.perf_counter(); km = 0
    for a in range(0, pairs, chunk):
        e = min(pairs, a + chunk)
        st = g.addReads(seq[off[a]:off[e]], qual[off[a]:off[e]], off[a:e + 1] - off[a], 3, storeReadPairedKmers=True)
        km += st.kmers
    dt = time.perf_counter() - t0
g.clearAllBf()
t0 = time.perf_counter()
st = g.addBatch(b, storeReadPairedKmers=True, first=0, n=pairs)
dr = time.perf_counter() - t0
# ---- the same reads as FASTQ text (R/io/FastqReader.java:140-186): file -> rb_fastq_split -> rb_graph_add_reads ----
from rnabloom import io as rio
L = int(off[1] - off[0])
assert (np.diff(off) == L).all()
rec = np.empty((pairs, 10 + 1 + L + 3 + L + 1), np.uint8)
ids = np.arange(pairs, dtype=np.int64)
rec[:, 0] = ord("@")
for d in range(9):
    rec[:, 9 - d] = ord("0") + (ids // 10 ** d) % 10
rec[:, 10] = 10
rec[:, 11:11 + L] = seq.reshape(pairs, L)
rec[:, 11 + L] = 10; rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
rec[:, 14 + L:14 + 2 * L] = qual.reshape(pairs, L)
rec[:, 14 + 2 * L] = 10
fq = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rb_host_path_L.fq")
rec.tofile(fq); fq_bytes = rec.size; del rec
for rep in range(2):
    g.clearAllBf()
    t0 = time.perf_counter()
    text = np.fromfile(fq, np.uint8)                       # page cache -> memory
    t1 = time.perf_counter()
    s2, q2, o2 = rio.splitFastq(text)
    t2 = time.perf_counter()
    st2 = g.addReads(s2, q2, o2, 3, storeReadPairedKmers=True)
    t3 = time.perf_counter()
assert st2.kmers == km and (s2 == seq).all() and (o2 == off).all()
# ---- the same file, records found on the GPU: file -> rb_graph_add_fastq ----
for rep in range(2):
    g.clearAllBf()
    u0 = time.perf_counter()
    text = np.fromfile(fq, np.uint8)
    u1 = time.perf_counter()
    st3, nrec = g.addFastq(text, 3, storeReadPairedKmers=True)
    u2 = time.perf_counter()
assert st3.kmers == km and nrec == pairs
for rep in range(2):
    g.clearAllBf()
    m0 = time.perf_counter()
    st4, nrec = g.addFastq(np.memmap(fq, np.uint8, "r"), 3, storeReadPairedKmers=True)      # no read() copy: the page cache is the source
    m1 = time.perf_counter()
assert st4.kmers == km
# ---- .gz input (FileUtils.getTextFileReader): a tenth of the text as one gzip member and as BGZF (bgzip) blocks ----
import struct, zlib
sub = np.fromfile(fq, np.uint8, count=(pairs // 10) * (fq_bytes // pairs)).tobytes()
def _bgzf(data, block=65280):
    out = []
    for a0 in list(range(0, len(data), block)) + [None]:
        chunk = b"" if a0 is None else data[a0:a0 + block]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) + body
                   + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)
gz_one = zlib.compressobj(1, zlib.DEFLATED, 31); gz_one = gz_one.compress(sub) + gz_one.flush()
gz_blk = _bgzf(sub)
z0 = time.perf_counter(); t1_ = rio.gunzip(gz_one); z1 = time.perf_counter(); t2_ = rio.gunzip(gz_blk); z2 = time.perf_counter()
assert t1_.tobytes() == sub and t2_.tobytes() == sub
gz_line = ("gunzip of %.2f GB of FASTQ text: one gzip member %.2f s (%.2f GB/s of text, one thread); BGZF blocks %.3f s (%.1f GB/s of text, all host threads)"
           % (len(sub) / 1e9, z1 - z0, len(sub) / 1e9 / (z1 - z0), z2 - z1, len(sub) / 1e9 / (z2 - z1)))
os.remove(fq)
print(gz_line)
print("FASTQ file path, records found on the GPU: read %.2f s, rb_graph_add_fastq %.2f s -> %.2f G k-mers/s end to end; from an mmap of the file: %.2f s -> %.2f G k-mers/s"
      % (u1 - u0, u2 - u1, km / (u2 - u0) / 1e9, m1 - m0, km / (m1 - m0) / 1e9))
print("FASTQ file path (%.2f GB of text): read %.2f s (%.1f GB/s), rb_fastq_split %.2f s (%.1f GB/s), rb_graph_add_reads %.2f s -> %.2f G k-mers/s end to end"
      % (fq_bytes / 1e9, t1 - t0, fq_bytes / 1e9 / (t1 - t0), t2 - t1, fq_bytes / 1e9 / (t2 - t1), t3 - t2, km / (t3 - t0) / 1e9))
print("host ASCII path (seq+qual %.2f GB over PCIe, %d reads per call): %.2f G k-mers/s (%.2f s for %d k-mers)"
      % (2 * seq.size / 1e9, chunk, km / dt / 1e9, dt, km))
print("same reads, batch already resident in HBM: %.2f G k-mers/s" % (st.kmers / dr / 1e9))
