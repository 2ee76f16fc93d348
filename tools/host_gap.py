#!/usr/bin/env python3
"""Where the host-resident leg's extra milliseconds go (bench.py `host_resident` against the HBM-resident step), config 2 on one GPU:
  A  resident:      clear + addBatch x 2
  B  as benched:    prefetchPacked x 2 + clear + addPacked x 2               (every byte uploaded inside the timed region; HOST_GAP_CLEAR_FIRST=1: the clear first)
  C  pre-uploaded:  prefetchPacked x 2, wait for the copies, THEN time clear + addPacked x 2   (the packed path without its link time)
  D  clear alone
  E  file 1 pre-uploaded, file 2 uploaded inside the timed region beside file 1's insert
B - A = what the upload costs the step; C - A = what the packed path costs without the link (bookkeeping, offset kernels, the sub-batch plan
from pinned offsets); B - C = waiting for data + contention with the copies.  RB_HOST_TIMING=1 adds the library's own per-call lines."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rna-bloom_amd")]
import torch
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nk = 450_000_000 * pairs // 50_000_000
size = N.lib.rb_expected_size(nk, 0.01, 2)
b = ReadBatch.synthetic(pairs, 64_000_000 * pairs // 50_000_000, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED)
g = BloomFilterDeBruijnGraph(size, size, size, 2, 2, 2, 25, False, True, rngSeed=1)
g.setReadPairedKmerDistance(115)
LATE = bool(os.environ.get("HOST_GAP_FILES_LATE"))     # pack the files after the first resident steps (bench.py's order) instead of before
files = [] if LATE else [(b.downloadPacked(0, pairs), False), (b.downloadPacked(pairs, pairs), True)]
sync = torch.cuda.synchronize
if os.environ.get("HOST_GAP_PROFILE"):            # bench.py runs its host-resident leg with the stage events on
    g.profileEnable(True)


def timed(fn, before=None):
    out = []
    for _ in range(reps + 1):
        if before:
            before()
        sync(); t0 = time.perf_counter(); fn(); sync(); out.append((time.perf_counter() - t0) * 1e3)
    return out[1:]


def resident():
    g.clearAllBf()
    g.addBatch(b, storeReadPairedKmers=True, first=0, n=pairs)
    g.addBatch(b, reverseComplement=True, storeReadPairedKmers=True, first=pairs, n=pairs)


def packed():
    if os.environ.get("HOST_GAP_CLEAR_FIRST"):
        g.clearAllBf()
    for ph, _ in files:
        g.prefetchPacked(ph)
    if not os.environ.get("HOST_GAP_CLEAR_FIRST"):
        g.clearAllBf()                  # (as bench.py does: the clear waits for its memsets; the uploads are under way meanwhile)
    for ph, rc in files:
        g.addPacked(ph, reverseComplement=rc, storeReadPairedKmers=True)


def pre():
    for ph, _ in files:
        g.prefetchPacked(ph)


def packed_pre():
    g.clearAllBf()
    for ph, rc in files:
        g.addPacked(ph, reverseComplement=rc, storeReadPairedKmers=True)


def pre1():
    g.prefetchPacked(files[0][0])


def packed_pre1():                      # file 1 is there already, file 2 travels inside the timed region beside file 1's insert
    g.clearAllBf()
    g.prefetchPacked(files[1][0])
    for ph, rc in files:
        g.addPacked(ph, reverseComplement=rc, storeReadPairedKmers=True)


fmt = lambda v: " / ".join("%.1f" % x for x in v)
if os.environ.get("HOST_GAP_LINK_FIRST"):         # what bench.py did before the timed steps until round 6's last day: the link alone through a packed stream
    from rnabloom.graph import PackedStream
    chunk = min(pairs, 12_500_000)
    ps = PackedStream(chunk, max(ph.words_before(min(r0 + chunk, ph.n_reads)) - ph.words_before(r0) for ph, _ in files for r0 in range(0, ph.n_reads, chunk)), device=g.device)
    sync(); t0 = time.perf_counter()
    for ph, _ in files:
        for r0 in range(0, ph.n_reads, chunk):
            ps.begin(ph, r0, min(chunk, ph.n_reads - r0)); ps.finish()
    print("link alone %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    ps.close()
A = timed(resident); print("A resident                 ", fmt(A), flush=True)
if LATE:
    files = [(b.downloadPacked(0, pairs), False), (b.downloadPacked(pairs, pairs), True)]
B = timed(packed); print("B packed, as benched       ", fmt(B), flush=True)
C = timed(packed_pre, before=pre); print("C packed, pre-uploaded     ", fmt(C), flush=True)
D = timed(g.clearAllBf); print("D clear alone              ", fmt(D), flush=True)
E = timed(packed_pre1, before=pre1); print("E packed, file 1 pre-uploaded", fmt(E), flush=True)
A2 = timed(resident); print("A resident again           ", fmt(A2), flush=True)
def two(fn):
    fn(); sync(); t0 = time.perf_counter(); fn(); fn(); sync(); return (time.perf_counter() - t0) * 500
print("two steps back to back, no sync between: resident %.1f  packed %.1f  resident %.1f  packed %.1f" % (two(resident), two(packed), two(resident), two(packed)), flush=True)
m = lambda v: sorted(v)[len(v) // 2]
print("medians: A %.1f  B %.1f  C %.1f  D %.1f  E %.1f   B - A = %.1f   C - A = %.1f   B - C = %.1f   E - A = %.1f (file 2's upload beside the insert)   B - E = %.1f (waiting for file 1's head)" % (m(A + A2), m(B), m(C), m(D), m(E), m(B) - m(A + A2), m(C) - m(A + A2), m(B) - m(C), m(E) - m(A + A2), m(B) - m(E)))
