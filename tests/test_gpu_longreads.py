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


# ---- the other sketch iterators and SeqSubsampler's hashing halves (a20) ----
EDGE_READS = [b"ACGT" * 10, b"", b"ACGTACGTAC", b"A" * 200, b"ACGTTGCA" * 40]


@pytest.mark.parametrize("canonical,slide", [(False, False), (False, True), (True, False)])
@pytest.mark.parametrize("n,wmin,wmax", [(3, 12, 61), (2, 12, 50), (4, 5, 9)])
def test_randstrobes_match_oracle(canonical, slide, n, wmin, wmax):
    reads = long_reads(30, 11) + EDGE_READS
    so, h, pos, _ = G.randstrobes(reads, 11, n, wmin, wmax, canonical=canonical, slide=slide)
    tot = 0
    for i, sq in enumerate(reads):
        eh, ep = rbo.randstrobes(sq, 11, n, wmin, wmax, canonical=canonical, slide=slide)
        a, b = so[i], so[i + 1]
        assert b - a == len(eh)
        assert (h[a:b] == eh).all() and (pos[a:b] == ep).all()
        tot += len(eh)
    assert tot > 20000


@pytest.mark.parametrize("canonical", [False, True])
def test_strobe3_match_oracle(canonical):
    reads = long_reads(30, 12) + EDGE_READS + [rbo_seq for rbo_seq in (b"ACGTGCTAGCTAGGATC" * 9, b"ACGTGCTAGCTAGGATC" * 3)]
    so, h, pos, _ = G.strobe3(reads, 11, 12, 61, canonical=canonical)
    tot = 0
    for i, sq in enumerate(reads):
        eh, ep = rbo.strobe3(sq, 11, 12, 61, canonical)
        a, b = so[i], so[i + 1]
        assert b - a == len(eh)
        assert (h[a:b] == eh).all() and (pos[a:b] == ep).all()
        tot += len(eh)
    assert tot > 20000


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_next_minimizers_and_minimizer_sets_match_oracle(mode):
    reads = long_reads(40, 13) + EDGE_READS + [b"ACGTACGTACGTACGTACGTACGTACG", b"ACGTACGTACGTACGTACGTACGTACGT", b"ACGTACGTACGTA", b"ACGTACGTACGT"]
    k, w = 13, 15
    mo, h, p = G.nextMinimizers(reads, k, w, mode)
    rng = np.random.default_rng(5)
    stale = rng.integers(0, 1 << 63, len(reads), dtype=np.int64).astype(np.uint64) * np.uint64(2)    # spans the sign bit
    so, sv = G.getMinimizers(reads, k, w, mode)
    so2, sv2 = G.getMinimizers(reads, k, w, mode, stale=stale)
    for i, s in enumerate(reads):
        eh, ep = rbo.minimizers_next(s, k, w, mode)
        a, b = mo[i], mo[i + 1]
        assert b - a == len(eh) and (h[a:b] == eh).all() and (p[a:b] == ep).all()
        es = rbo.minimizer_set(s, k, w, mode)
        assert (sv[so[i]:so[i + 1]] == es).all() and so[i + 1] - so[i] == len(es)
        es2 = rbo.minimizer_set(s, k, w, mode, stale=int(stale[i]))
        assert (sv2[so2[i]:so2[i + 1]] == es2).all() and so2[i + 1] - so2[i] == len(es2)
    assert mo[-1] > 3000 and so[-1] > 3000


@pytest.mark.parametrize("canonical", [False, True])
def test_subsampler_hashing_halves(canonical):
    """SeqSubsampler.strobemerBased / kmerBased: hash every strobemer / k-mer pair of a read and ask the counting filter
    for its multiplicity (R/util/SeqSubsampler.java:389-395, 176-179, 266-268) — one call, counts looked up on the device"""
    from rnabloom.bloom import CountingBloomFilter
    reads = long_reads(25, 14)
    k = 11
    cbf = CountingBloomFilter(400_009, 2, k, rngSeed=3)
    ocbf = rbo.Graph(64, 400_009, 0, 1, 2, 1, k, True, False, 3)
    # fill with the strobemers of the first reads (several times each), as the subsampler does when it keeps a read
    so, h, s, e = G.strobemers(reads[:10], k, 3, k + 1, 2 * k)
    for _ in range(3):
        cbf.increment(h)
        for x in h: ocbf.add_count_only(rbo.ntm64(int(x), k, 2))
    assert (cbf.toBytes() == ocbf.cbf_bytes()).all()

    def ocount(x):           # CountingBloomFilter.getCount(long) on the oracle's counters (:235-251 + MiniFloat.toFloat :40-45)
        raw = ocbf.cbf_bytes()
        b = min(int(raw[(int(v) >> 1) % 400_009]) for v in rbo.ntm64(int(x), k, 2))
        return float(b) if b <= 7 else float(((b & 7) | 8) << ((b >> 3) - 1))
    so, h, s, e = G.strobemers(reads, k, 3, k + 1, 2 * k)
    so2, h2, pos2, cnt = G.randstrobes(reads, k, 3, k + 1, 2 * k, slide=True, counts_from=cbf)
    assert (so2 == so).all() and (h2 == h).all() and (pos2[:, 0] == s).all() and (pos2[:, 2] + k - 1 == e).all()
    exp = np.array([ocount(x) for x in h], np.float32)
    assert (cnt == exp).all() and (cnt >= 3).sum() > 100 and (cnt == 0).sum() > 100
    # k-mer pairs (gap 1), stranded and canonical
    po, ph, pc = G.kmerPairHashes(reads, k, k + 1, canonical=canonical, counts_from=cbf)
    for i, sq in enumerate(reads):
        assert (ph[po[i]:po[i + 1]] == rbo.kmer_pair_hashes(sq, k, k + 1, canonical)).all()
    assert (pc == np.array([ocount(x) for x in ph], np.float32)).all()


def test_nbits_files_decode_on_the_device_and_insert_like_ascii(tmp_path):
    """.nbits (R/io/NucleotideBits{Reader,Writer}.java) -> packed batch by a GPU bit permute; same filters as from ASCII"""
    from rnabloom import io as RIO
    rng = np.random.default_rng(21)
    reads = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(L)).tolist()) for L in list(range(0, 70)) + [150] * 40 + [999, 1000, 1001, 4097]]
    seq, _, off = rbo.pack_reads(reads)
    RIO.writeNbits(tmp_path / "r.nbits", seq, off)
    data = (tmp_path / "r.nbits").read_bytes()
    batch, used = RIO.batchFromNbits(data)
    assert used == len(data) and batch.n_reads == len(reads)
    s2, o2 = batch.download()
    assert (o2 == off).all() and (s2 == seq).all()
    # a truncated file: the last record is not returned (NucleotideBitsReader.next -> null), the rest is
    b2, used2 = RIO.batchFromNbits(data[:-3])
    assert b2.n_reads == len(reads) - 1 and used2 == len(data) - (4 + (4097 + 3) // 4)
    b3, _ = RIO.batchFromNbits(data, max_reads=5)
    assert b3.n_reads == 5
    # insert from the .nbits batch == insert from ASCII
    sizes = (600_011, 900_007, 64)
    ga = G.BloomFilterDeBruijnGraph(*sizes, 2, 2, 1, 25, False, False, rngSeed=1)
    gb = G.BloomFilterDeBruijnGraph(*sizes, 2, 2, 1, 25, False, False, rngSeed=1)
    sa = ga.addReads(seq, None, off, 3)
    sb = gb.addBatch(batch)
    assert sa.kmers == sb.kmers > 10000
    assert (ga.exportFilter(N.DBGBF) == gb.exportFilter(N.DBGBF)).all() and (ga.exportFilter(N.CBF) == gb.exportFilter(N.CBF)).all()


def test_sketch_calls_accept_packed_arrays_and_existing_outputs():
    """minimizers / strobemers / getKmers from packed (sequence, offsets) arrays into result arrays that exist already
    give what the list-of-bytes forms return"""
    reads = long_reads(60, 11)
    seq = np.frombuffer(b"".join(reads), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    mo, h, p = G.minimizers(reads, 13, 15, 1)
    oh, op = np.zeros(h.size + 7, np.uint64), np.zeros(h.size + 7, np.int64)
    mo2, h2, p2 = G.minimizers((seq, off), 13, 15, 1, out=(oh, op))
    assert (mo2 == mo).all() and (h2 == h).all() and (p2 == p).all() and h2.base is oh
    so, sh, ss, se = G.strobemers(reads, 11, 3, 12, 61)
    bh, bs, be = np.zeros(sh.size, np.uint64), np.zeros(sh.size, np.int32), np.zeros(sh.size, np.int32)
    so2, sh2, ss2, se2 = G.strobemers((seq, off), 11, 3, 12, 61, out=(bh, bs, be))
    assert (so2 == so).all() and (sh2 == sh).all() and (ss2 == ss).all() and (se2 == se).all()
    gg = G.BloomFilterDeBruijnGraph(3_000_017, 6_000_011, 10_007, 2, 2, 2, 35, False, False, rngSeed=4)
    gg.addReads(seq, None, off, 3)
    ko, f, r, c = gg.getKmers(reads)
    of, orr, oc = np.zeros(f.size, np.uint64), np.zeros(f.size, np.uint64), np.zeros(f.size, np.float32)
    ko2, f2, r2, c2 = gg.getKmers(reads, out=(of, orr, oc))
    assert (ko2 == ko).all() and (f2 == f).all() and (r2 == r).all() and (c2 == c).all() and c.max() > 1
