"""BASELINE config 5 shape: ~2 kb noisy long reads, k=35 stage-1 insert (no pairs, R/RNABloom.java:1313-1316),
minimizers (m=13,w=15) and order-3 strobemers (k=11, wMin=12, wMax=61; R/util/SeqSubsampler.java:360-363)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import rbo
from rnabloom import _native as N
from rnabloom import graph as G


def long_reads(n, seed, mean=2000):
    rng = np.random.default_rng(seed)
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 60000)]
    reads = []
    for _ in range(n):
        L = int(np.clip(rng.lognormal(np.log(mean), 0.5), 20, 12000))
        a = int(rng.integers(0, len(genome) - L))
        r = genome[a:a + L].copy()
        err = rng.random(L) < 0.05                       # ONT-like substitutions
        r[err] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(err.sum()))]
        r[rng.random(L) < 0.002] = ord("N")
        dele = rng.random(L) < 0.01                      # deletions
        reads.append(r[~dele].tobytes())
    return reads


def test_long_read_insert_k35_bit_exact():
    reads = long_reads(300, 1)
    seq, _, off = rbo.pack_reads(reads)
    sizes = (3_000_017, 6_000_011, 10_007)
    og = rbo.Graph(*sizes, 2, 2, 2, 35, False, False, 4)
    gg = G.BloomFilterDeBruijnGraph(*sizes, 2, 2, 2, 35, False, False, rngSeed=4, maxBatchKmers=200_000)
    so = og.add_reads(seq, None, off, 3, 0)
    sg = gg.addReads(seq, None, off, 3)
    assert sg.kmers == so.kmers > 400_000
    assert (gg.exportFilter(N.DBGBF) == og.dbgbf_bytes()).all()
    assert (gg.exportFilter(N.CBF) == og.cbf_bytes()).all()
    # counts along a read
    ko, f, r, c = gg.getKmers(reads[:5])
    for i in range(5):
        ef, er, ec = og.get_kmers(reads[i])
        assert (f[ko[i]:ko[i + 1]] == ef).all() and (r[ko[i]:ko[i + 1]] == er).all() and (c[ko[i]:ko[i + 1]] == ec).all()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_minimizers_match_oracle(mode):
    reads = long_reads(40, 2) + [b"ACGT", b"", b"ACGTACGTACGTACGTACGTACGTACGTACG"]
    mo, h, p = G.minimizers(reads, 13, 15, mode)
    for i, s in enumerate(reads):
        eh, ep = rbo.minimizers(s, 13, 15, mode)
        a, b = mo[i], mo[i + 1]
        assert b - a == len(eh)
        assert (h[a:b] == eh).all()
        hv, _ = rbo.hash_region(s, 13, 1, mode) if len(s) >= 13 else (np.zeros((0, 1), np.uint64), None)
        for q in range(0, b - a, 37):                     # position attains the minimum, leftmost
            win = hv[q:q + 15, 0].astype(np.int64)
            assert p[a + q] == q + int(np.argmin(win))


def test_strobemers_match_oracle():
    reads = long_reads(40, 3) + [b"ACGT" * 10, b""]
    so, h, s, e = G.strobemers(reads, 11, 3, 12, 61)
    tot = 0
    for i, sq in enumerate(reads):
        eh, es, ee = rbo.strobemers(sq, 11, 3, 12, 61)
        a, b = so[i], so[i + 1]
        assert b - a == len(eh)
        assert (h[a:b] == eh).all() and (s[a:b] == es).all() and (e[a:b] == ee).all()
        tot += len(eh)
    assert tot > 10000


def test_long_reads_at_scale():
    """6 000 long reads (~12 M bases): the k = 35 insert of the whole set against the oracle, minimizers and order-3
    strobemers of the whole set in one call each, checked read by read at the beginning and at the end of the launch
    (the device is full: a defect that needs several wavefronts of a kernel per SIMD shows only at this size)."""
    reads = long_reads(6000, 9)
    seq, _, off = rbo.pack_reads(reads)
    sizes = (200_000_033, 200_000_033, 10_007)
    og = rbo.Graph(*sizes, 2, 2, 2, 35, False, False, 4)
    gg = G.BloomFilterDeBruijnGraph(*sizes, 2, 2, 2, 35, False, False, rngSeed=4)
    so = og.add_reads(seq, None, off, 3, 0)
    sg = gg.addReads(seq, None, off, 3)
    assert sg.kmers == so.kmers > 9_000_000
    assert np.array_equal(gg.exportFilter(N.DBGBF), og.dbgbf_bytes())
    assert np.array_equal(gg.exportFilter(N.CBF), og.cbf_bytes())
    sample = list(range(0, 25)) + list(range(len(reads) - 25, len(reads)))
    for mode in (0, 1):
        mo, h, p = G.minimizers(reads, 13, 15, mode)
        assert mo[-1] > 9_000_000
        for i in sample:
            eh, ep = rbo.minimizers(reads[i], 13, 15, mode)
            a, b = mo[i], mo[i + 1]
            assert b - a == len(eh) and (h[a:b] == eh).all() and (p[a:b] == ep).all()
    so_, h, s, e = G.strobemers(reads, 11, 3, 12, 61)
    assert so_[-1] > 8_000_000
    for i in sample:
        eh, es, ee = rbo.strobemers(reads[i], 11, 3, 12, 61)
        a, b = so_[i], so_[i + 1]
        assert b - a == len(eh) and (h[a:b] == eh).all() and (s[a:b] == es).all() and (e[a:b] == ee).all()
