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
    reads = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(L)).tolist()) for L in list(range(1, 70)) + [150] * 40 + [999, 1000, 1001, 4097]]
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
    # an empty sequence ends the iteration, as in the reference: fin.read(new byte[0]) returns 0, which is not > 0, so
    # NucleotideBitsReader.next() returns null there (R/io/NucleotideBitsReader.java:39-47)
    s3, _, o3 = rbo.pack_reads([b"ACGTA", b"", b"GGATTCA"])
    RIO.writeNbits(tmp_path / "e.nbits", s3, o3)
    b4, used4 = RIO.batchFromNbits((tmp_path / "e.nbits").read_bytes())
    assert b4.n_reads == 1 and used4 == 4 + 2 and bytes(b4.download()[0]) == b"ACGTA"
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


def _nasty(seed, n=14, lo=90, hi=260):
    """reads with letters outside ACGTU (reverse-strand seed by `ch & 7`, R/bloom/hash/NTHash.java:30, 133-166), homopolymers
    and short tandem repeats (windows full of equal hashes: every tie rule matters)"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        s = acgt[rng.integers(0, 4, L)].copy()
        for _ in range(int(rng.integers(0, 4)) * max(1, L // 250)):
            a = int(rng.integers(0, L - 40)); unit = acgt[rng.integers(0, 4, int(rng.integers(1, 4)))]
            rep = np.tile(unit, 40)[:int(rng.integers(20, 40))]
            s[a:a + rep.size] = rep[:max(0, min(rep.size, L - a))]
        if i % 2:
            for ch in b"KMSWYIELOQDRBHVNXacgtu":
                for _ in range(max(1, L // 400)):
                    if rng.random() < 0.5: s[int(rng.integers(0, L))] = ch
        out.append(s.tobytes())
    return out + [b"A" * 120, b"ACACACACAC" * 14, b"ACGT", b""]


def test_three_way_sketches_hip_c_oracle_python_oracle():
    """HIP == oracle/rb_oracle_sketch.c == oracle/rbo_py.py (written from the Java a second time, in another shape: hashes from
    scratch, brute-force argmin, a literal LongRollingWindow) on reads where the tie rules and the `ch & 7` lookup matter"""
    from oracle import rbo_py as P
    reads = _nasty(3)
    for mode in (0, 1, 2):
        mo, h, p = G.minimizers(reads, 13, 15, mode)
        nmo, nh, npos = G.nextMinimizers(reads, 13, 15, mode)
        smo, sv = G.getMinimizers(reads, 13, 15, mode)
        for i, s in enumerate(reads):
            want = P.minimizers(s, 13, 15, mode)
            eh, ep = rbo.minimizers(s, 13, 15, mode)
            assert [int(x) for x in h[mo[i]:mo[i + 1]]] == [a for a, _ in want] == [int(x) for x in eh], (i, mode)
            assert [int(x) for x in p[mo[i]:mo[i + 1]]] == [b for _, b in want] == [int(x) for x in ep], (i, mode)
            assert list(zip([int(x) for x in nh[nmo[i]:nmo[i + 1]]], [int(x) for x in npos[nmo[i]:nmo[i + 1]]])) == P.minimizers_next(s, 13, 15, mode)
            assert [int(x) for x in sv[smo[i]:smo[i + 1]]] == P.minimizer_set(s, 13, 15, mode)
    so, sh, ss, se = G.strobemers(reads, 11, 3, 12, 31)
    for i, s in enumerate(reads):
        assert list(zip([int(x) for x in sh[so[i]:so[i + 1]]], [int(x) for x in ss[so[i]:so[i + 1]]], [int(x) for x in se[so[i]:so[i + 1]]])) == P.strobemer_intervals(s, 11, 3, 12, 31)
    for canonical, slide in ((False, False), (False, True), (True, False)):
        ro, rh, rp, _ = G.randstrobes(reads, 11, 3, 5, 20, canonical=canonical, slide=slide)
        for i, s in enumerate(reads):
            want = P.randstrobes(s, 11, 3, 5, 20, canonical)
            assert [int(x) for x in rh[ro[i]:ro[i + 1]]] == [a for a, _ in want], (i, canonical, slide)
            assert [list(map(int, r)) for r in rp[ro[i]:ro[i + 1]]] == [b for _, b in want]
    for canonical in (False, True):
        to, th, tp, _ = G.strobe3(reads, 11, 6, 25, canonical=canonical)
        po, ph, _ = G.kmerPairHashes(reads, 13, 9, canonical=canonical)
        for i, s in enumerate(reads):
            want = P.strobe3(s, 11, 6, 25, canonical)
            assert [int(x) for x in th[to[i]:to[i + 1]]] == [a for a, _ in want], (i, canonical)
            assert [list(map(int, r)) for r in tp[to[i]:to[i + 1]]] == [b for _, b in want]
            assert [int(x) for x in ph[po[i]:po[i + 1]]] == P.kmer_pair_hashes(s, 13, 9, canonical)


def test_get_kmers_reproduces_the_reverse_seed_of_letters_outside_acgtu():
    """getKmers on sequences holding K M S W Y I E L O Q D ...: the reference's reverse hash takes the complement's seed through
    `ch & 7` (NTHash.java:30, 133-166), so such a letter hashes as a base on the reverse strand and as nothing on the forward
    strand; forward hash, reverse hash and count equal the oracle's and the from-scratch Python hashes"""
    from oracle import rbo_py as P
    reads = [r for r in _nasty(9, 20, 60, 200) if len(r) >= 25]
    assert any(ch in r for r in reads for ch in b"KMSWYIE")
    for stranded in (False, True):
        og = rbo.Graph(300_007, 500_009, 64, 2, 2, 1, 25, stranded, False, 2)
        gg = G.BloomFilterDeBruijnGraph(300_007, 500_009, 64, 2, 2, 1, 25, stranded, False, rngSeed=2)
        seq, _, off = rbo.pack_reads(reads)
        og.add_reads(seq, None, off, 3, 0); gg.addReads(seq, None, off, 3)
        ko, f, r, c = gg.getKmers(reads)
        differs = 0
        for i, s in enumerate(reads):
            ef, er, ec = og.get_kmers(s)
            pf, pr = P.kmer_hashes(s, 25)
            a, b = ko[i], ko[i + 1]
            assert [int(x) for x in f[a:b]] == pf == [int(x) for x in ef], i
            if not stranded:
                assert [int(x) for x in r[a:b]] == pr == [int(x) for x in er], i
                # the validity-bit-only packing of round 2 gave these letters seed 0 on both strands
                zr = P.kmer_hashes(bytes(ch if ch in b"ACGTUacgtu" else ord("N") for ch in s), 25)[1]
                differs += sum(x != y for x, y in zip(pr, zr))
            assert (c[a:b] == ec).all()
        assert stranded or differs > 50


@pytest.mark.parametrize("simple", [False, True])
def test_tile_kernels_on_long_tie_heavy_reads(simple, monkeypatch):
    """the tile kernels (LDS-staged hashes, several tiles per read) and the one-thread-per-output kernels they replace, on reads
    of several thousand bases with homopolymers and tandem repeats: equal to the C oracle; minimizer positions in runs of tied
    windows are replayed from the tie-free window in front of them"""
    if simple: monkeypatch.setenv("RB_SKETCH_SIMPLE", "1")
    reads = _nasty(21, 10, 2500, 6000) + long_reads(6, 5, mean=3000)
    for mode in (0, 1):
        mo, h, p = G.minimizers(reads, 13, 15, mode)
        for i, s in enumerate(reads):
            eh, ep = rbo.minimizers(s, 13, 15, mode)
            assert (h[mo[i]:mo[i + 1]] == eh).all() and (p[mo[i]:mo[i + 1]] == ep).all(), (i, mode)
    so, sh, ss, se = G.strobemers(reads, 11, 3, 12, 61)
    ro, rh, rp, _ = G.randstrobes(reads, 11, 3, 12, 61, canonical=True)
    to, th, tp, _ = G.strobe3(reads, 11, 12, 61, canonical=True)
    uo, uh, up, _ = G.strobe3(reads, 11, 12, 61, canonical=False)
    for i, s in enumerate(reads):
        oh, os_, oe = rbo.strobemers(s, 11, 3, 12, 61)
        assert (sh[so[i]:so[i + 1]] == oh).all() and (ss[so[i]:so[i + 1]] == os_).all() and (se[so[i]:so[i + 1]] == oe).all(), i
        eh, ep = rbo.randstrobes(s, 11, 3, 12, 61, canonical=True)
        assert (rh[ro[i]:ro[i + 1]] == eh).all() and (rp[ro[i]:ro[i + 1]] == ep).all(), i
        eh, ep = rbo.strobe3(s, 11, 12, 61, canonical=True)
        assert (th[to[i]:to[i + 1]] == eh).all() and (tp[to[i]:to[i + 1]] == ep).all(), i
        eh, ep = rbo.strobe3(s, 11, 12, 61, canonical=False)
        assert (uh[uo[i]:uo[i + 1]] == eh).all() and (up[uo[i]:uo[i + 1]] == ep).all(), i
