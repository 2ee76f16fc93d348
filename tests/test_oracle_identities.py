"""Algebraic identities of SURVEY.md Appendix A.8 + filter/graph semantics of the oracle."""
import numpy as np
import pytest

from oracle import rbo
from rnabloom import synth

RNG = np.random.default_rng(7)


def rand_seq(n, alphabet=b"ACGT"):
    return bytes(np.frombuffer(alphabet, np.uint8)[RNG.integers(0, len(alphabet), n)])


def revcomp(s):
    return bytes(synth.revcomp(np.frombuffer(s, np.uint8)))


def s64(x):
    x = int(x)
    return x - (1 << 64) if x >> 63 else x


@pytest.mark.parametrize("k", [5, 25, 35, 64, 65, 100])
def test_rolled_equals_scratch_and_revcomp(k):
    L = rbo.lib()
    s = rand_seq(k + 80)
    h, fr = rbo.hash_region(s, k, 3, rbo.CANON)
    rc = revcomp(s)
    hrc, frrc = rbo.hash_region(rc, k, 3, rbo.CANON)
    for i in range(len(h)):
        w = s[i:i + k]
        assert int(fr[i, 0]) == L.rbo_ntp64(w, k)                     # (ii)
        assert int(fr[i, 1]) == L.rbo_ntp64rc(w, k)
        assert L.rbo_ntp64(revcomp(w), k) == int(fr[i, 1])            # (iii)
        assert s64(h[i, 0]) == min(s64(fr[i, 0]), s64(fr[i, 1]))      # signed canonical min
    assert (h[::-1] == hrc).all()                                     # (iii) canonical invariant
    assert (fr[::-1, 0] == frrc[:, 1]).all() and (fr[::-1, 1] == frrc[:, 0]).all()
    # (vii) stranded RC iterator over read == forward iterator over revcomp(read), reversed
    hr, _ = rbo.hash_region(s, k, 2, rbo.RC)
    hf, _ = rbo.hash_region(rc, k, 2, rbo.FWD)
    assert (hr[::-1] == hf).all()


def test_case_and_u_equivalence():
    s = rand_seq(200)
    for alt in (s.lower(), s.replace(b"T", b"U"), s.lower().replace(b"t", b"u")):
        for mode in (rbo.FWD, rbo.CANON, rbo.RC):
            a, _ = rbo.hash_region(s, 25, 2, mode)
            b, _ = rbo.hash_region(alt, 25, 2, mode)
            assert (a == b).all()


def test_multihash_and_combine():
    M = (1 << 64) - 1
    for _ in range(50):
        b = int(RNG.integers(0, 1 << 63)) * 2 + int(RNG.integers(0, 2))
        k = int(RNG.integers(1, 200))
        hv = rbo.ntm64(b, k, 5)
        assert int(hv[0]) == b
        for i in range(1, 5):
            t = (b * ((i ^ ((k * 0x90b45d39fb6da1fa) & M)) & M)) & M
            assert int(hv[i]) == t ^ (t >> 27)
        a = int(RNG.integers(0, 1 << 63))
        assert rbo.lib().rbo_combine(a, b) == a ^ ((b + 0xFFFFFFFF9E3779B9 + ((a << 6) & M) + (b >> 1 >> 1)) & M)


@pytest.mark.parametrize("canonical", [0, 1])
@pytest.mark.parametrize("k", [25, 35, 64])
def test_neighbors_and_variants(canonical, k):
    L = rbo.lib()
    s = rand_seq(k)
    f, r = L.rbo_ntp64(s, k), L.rbo_ntp64rc(s, k)
    for direction in (0, 1):
        co = s[0] if direction == 0 else s[-1]
        nf, nr, nh = rbo.neighbors(f, r, co, k, 2, canonical, direction)
        for i, c in enumerate(b"ACGT"):
            t = s[1:] + bytes([c]) if direction == 0 else bytes([c]) + s[:-1]
            assert int(nf[i]) == L.rbo_ntp64(t, k)                    # (vi)
            if canonical:
                assert int(nr[i]) == L.rbo_ntp64rc(t, k)
            ref, _ = rbo.hash_region(t, k, 2, rbo.CANON if canonical else rbo.FWD)
            assert (nh[i] == ref[0]).all()
    for side in (0, 1):
        co = s[0] if side == 0 else s[-1]
        for c in b"ACGT":
            t = bytes([c]) + s[1:] if side == 0 else s[:-1] + bytes([c])
            vf, vr, vh = rbo.variant(f, r, co, c, k, 2, canonical, side)
            assert vf == L.rbo_ntp64(t, k)
            if canonical:
                assert vr == L.rbo_ntp64rc(t, k)
            ref, _ = rbo.hash_region(t, k, 2, rbo.CANON if canonical else rbo.FWD)
            assert (vh == ref[0]).all()


def test_pair_hash_invariants():
    k, d = 25, 40
    s = rand_seq(150)
    p, l, r = rbo.hash_pairs_region(s, k, 2, d, rbo.CANON)
    h, fr = rbo.hash_region(s, k, 2, rbo.CANON)
    assert len(p) == 150 - k - d + 1
    L = rbo.lib()
    for i in range(len(p)):
        a = L.rbo_combine(int(fr[i, 0]), int(fr[i + d, 0]))
        b = L.rbo_combine(int(fr[i + d, 1]), int(fr[i, 1]))
        base = a if s64(a) < s64(b) else b
        assert (p[i] == rbo.ntm64(base, k, 2)).all()
        assert (l[i] == h[i]).all() and (r[i] == h[i + d]).all()
    prc, _, _ = rbo.hash_pairs_region(revcomp(s), k, 2, d, rbo.CANON)  # (viii)
    assert (p[::-1] == prc).all()
    pf, _, _ = rbo.hash_pairs_region(revcomp(s), k, 2, d, rbo.FWD)
    pr, _, _ = rbo.hash_pairs_region(s, k, 2, d, rbo.RC)
    assert (pr[::-1] == pf).all()


def test_minifloat():
    L = rbo.lib()
    exp = {0: 0, 7: 7, 8: 8, 15: 15, 16: 16, 17: 18, 23: 30, 24: 32, 127: 245760.0}
    for b, v in exp.items():
        assert L.rbo_minifloat_to_float(b) == v
    for b in range(0, 16):
        assert L.rbo_minifloat_increment(b, 12345) == b + 1           # deterministic below 16
    assert L.rbo_minifloat_increment(127, 0) == 127                   # saturates
    for b in (16, 23, 24, 40, 126):
        s = (b >> 3) - 1
        assert L.rbo_minifloat_increment(b, 0) == b + 1
        assert L.rbo_minifloat_increment(b, 1) == b
        assert L.rbo_minifloat_increment(b, 1 << s) == b + 1
    # rng is a pure function of (seed, ordinal, pos) and roughly uniform
    v = np.array([L.rbo_rng31(5, o, p) for o in range(200) for p in range(20)])
    assert v.max() < (1 << 31) and abs((v & 1).mean() - 0.5) < 0.05
    assert L.rbo_rng31(5, 3, 4) == L.rbo_rng31(5, 3, 4) != L.rbo_rng31(6, 3, 4)


def test_expected_size():
    L = rbo.lib()
    assert L.rbo_expected_size(1000000, 0.01, 2) == int(np.ceil(1000000 * 18.982443385784062))
    assert abs(L.rbo_expected_size(10 ** 9, 0.01, 1) / 1e9 - 99.49916470861523) < 1e-6
    assert abs(L.rbo_expected_size(10 ** 9, 0.01, 3) / 1e9 - 12.364166878733823) < 1e-6


def test_bloom_layout_and_lookup_then_add():
    L = rbo.lib()
    size = 1003
    b = L.rbo_bloom_new(size, 2)
    h = np.array([2 * 17, 2 * (size + 9)], np.uint64)               # idx 17 and 9
    assert L.rbo_bloom_lookup_then_add(b, h.ctypes.data) == 0
    assert L.rbo_bloom_lookup_then_add(b, h.ctypes.data) == 1
    n = rbo.C.c_int64()
    p = L.rbo_bloom_bytes(b, rbo.C.byref(n))
    raw = np.ctypeslib.as_array(rbo.C.cast(p, rbo.C.POINTER(rbo.C.c_uint8)), (n.value,))
    assert n.value == (size + 7) // 8
    assert raw[1] == (1 << 1) and raw[2] == (1 << 1) and raw.sum() == 4   # LSB-first bits 9, 17
    same = np.array([2 * 5, 2 * 5], np.uint64)                      # both probes on one bit
    assert L.rbo_bloom_lookup_then_add(b, same.ctypes.data) == 0    # found = old(bit) && true
    assert L.rbo_bloom_popcount(b) == 3
    assert abs(L.rbo_bloom_fpr(b) - np.float32((3 / size) ** 2)) < 1e-9
    L.rbo_bloom_free(b)


def test_cbf_conservative_update_order_dependence():
    # keys->probes {(1,0),(3,0),(3,3)}, multiplicities {3,1,3}, 4 counters
    L = rbo.lib()

    def run(order):
        c = L.rbo_cbf_new(4, 2)
        keys = {"a": (1, 0), "b": (3, 0), "c": (3, 3)}
        for kname in order:
            h = np.array([2 * x for x in keys[kname]], np.uint64)
            L.rbo_cbf_increment(c, h.ctypes.data, 0)
        n = rbo.C.c_int64()
        p = L.rbo_cbf_bytes(c, rbo.C.byref(n))
        out = tuple(np.ctypeslib.as_array(rbo.C.cast(p, rbo.C.POINTER(rbo.C.c_uint8)), (4,)).tolist())
        L.rbo_cbf_free(c)
        return out

    finals = {run(o) for o in ("aaabccc", "cccbaaa", "acacacb", "bcccaaa", "abccaca")}
    assert len(finals) >= 2


def test_graph_add_counts():
    g = rbo.Graph(100003, 1000003, 100003, k=25)
    h = rbo.ntm64(0x1234567890ABCDEF, 25, 2)
    assert g.get_count(h) == 0 and not g.contains(h)
    for n in range(1, 20):
        g.add(h)
        assert g.contains(h)
        if n <= 17:
            assert g.get_count(h) == n                                # (x) count == multiplicity <= 17


def test_segmentation_regex_semantics():
    k = 5
    seq = b"ACGTANACGTACGTTTNNACG" + b"acgun" + b"ACGTAC"
    se = rbo.segments(seq, None, k, 3)
    assert se.tolist() == [[0, 5], [6, 16], [18, 25], [26, 32]]   # lower case and U are valid
    qual = bytearray(b"I" * len(seq))
    qual[8] = ord("#")            # PHRED 2 < 3 splits the [6,16) run into [6,8) (too short) and [9,16)
    qual[2] = ord("$")            # PHRED 3 is acceptable at q=3
    se = rbo.segments(seq, bytes(qual), k, 3)
    assert se.tolist() == [[0, 5], [9, 16], [18, 25], [26, 32]]
    assert rbo.segments(b"ACGT", None, k, 3).tolist() == []
    assert rbo.segments(b"", None, k, 3).tolist() == []


def test_add_reads_equals_per_kmer_api():
    d = synth.generate_pairs(300, G=20000, err=0.01, n_rate=0.002, seed=11)
    seq, off = synth.flat(d["left"])
    qual, _ = synth.flat(d["lqual"])
    k, dist = 25, 115
    args = dict(dbg_h=2, cbf_h=2, pk_h=2, k=k, stranded=False, use_read_pairs=True, rng_seed=3)
    g1 = rbo.Graph(200003, 1600003, 50021, **args)
    g1.set_read_pair_distance(dist)
    st = g1.add_reads(seq, qual, off, 3, rbo.STORE_READ_PAIRS)
    g2 = rbo.Graph(200003, 1600003, 50021, **args)
    nk = npairs = 0
    for i in range(len(off) - 1):
        s = bytes(seq[off[i]:off[i + 1]]); q = bytes(qual[off[i]:off[i + 1]])
        for a, b in rbo.segments(s, q, k, 3):
            h, _ = rbo.hash_region(s, k, 2, rbo.CANON, a, b)
            for row in h:
                g2.add(row)
            nk += len(h)
            p, _, _ = rbo.hash_pairs_region(s, k, 2, dist, rbo.CANON, a, b)
            for row in p:
                g2.add_read_pair(row)
            npairs += len(p)
    assert st.kmers == nk and st.pairs == npairs and st.reads == 300
    assert (g1.dbgbf_bytes() == g2.dbgbf_bytes()).all()
    assert (g1.rpkbf_bytes() == g2.rpkbf_bytes()).all()
    # counters agree wherever the RNG is not consulted (the per-k-mer API keys it differently)
    c1, c2 = g1.cbf_bytes(), g2.cbf_bytes()
    assert ((c1 == c2) | ((c1 > 16) & (c2 > 16)) | ((c1 >= 16) & (c2 >= 16))).all()


def test_get_kmers_and_neighbors_graph():
    d = synth.generate_pairs(200, G=5000, err=0.0, n_rate=0.0, seed=5)
    seq, off = synth.flat(d["left"])
    g = rbo.Graph(100003, 800003, 10007, k=25, use_read_pairs=False)
    g.add_reads(seq, None, off, 3, 0)
    s = bytes(seq[:150])
    f, r, c = g.get_kmers(s)
    assert (c >= 1).all()
    s2 = s[:60] + b"N" + s[61:]
    f2, r2, c2 = g.get_kmers(s2)
    assert (c2[36:61] == 0).all() and (c2[:36] == c[:36]).all() and (c2[61:] == c[61:]).all()
    nf, nr, cnt = g.neighbors(f[10], r[10], s[10], 0)
    i = b"ACGT".index(s[35:36])
    assert nf[i] == f[11] and nr[i] == r[11] and cnt[i] == c[11]
    pf, pr, pc = g.neighbors(f[10], r[10], s[34], 1)
    j = b"ACGT".index(s[9:10])
    assert pf[j] == f[9] and pr[j] == r[9] and pc[j] == c[9]


def test_minimizers_window_min():
    s = rand_seq(400)
    k, w = 13, 15
    mh, mp = rbo.minimizers(s, k, w, rbo.CANON)
    h, _ = rbo.hash_region(s, k, 1, rbo.CANON)
    sv = np.array([s64(x) for x in h[:, 0]])
    assert len(mh) == len(sv) - w + 1
    for p in range(len(mh)):
        win = sv[p:p + w]
        assert s64(mh[p]) == win.min()
        assert sv[mp[p]] == win.min() and p <= mp[p] < p + w


def test_strobemers_reference_shape():
    s = rand_seq(600)
    k, n, wmin, wmax = 11, 3, 12, 61
    sh, ss, se = rbo.strobemers(s, k, n, wmin, wmax)
    nk = len(s) - k + 1
    assert len(sh) == nk - wmax * (n - 2) - wmin
    h, _ = rbo.hash_region(s, k, 1, rbo.FWD)
    hv = [int(x) for x in h[:, 0]]
    L = rbo.lib()
    for p in (0, 7, len(sh) - 1):
        cur = hv[p]
        last = p
        for st in range(n - 1):
            lo = p + st * wmax + wmin
            hi = min(p + st * wmax + wmax, nk)
            best = min(range(lo, hi), key=lambda i: (L.rbo_combine(cur, hv[i]), -i))
            cur = L.rbo_combine(cur, hv[best]); last = best
        assert int(sh[p]) == cur and ss[p] == p and se[p] == last + k - 1
    assert len(rbo.strobemers(s[:100], k, n, wmin, wmax)[0]) == 0


def test_oracle_fold_is_the_digest_rb_filter_fold_computes():
    """rbo_fold (in place, for filters too large to copy) == rnabloom.graph.fold_bytes over the same bytes — the numpy form the GPU
    suite pins rb_filter_fold to (tests/test_gpu_config3.py)"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rna-bloom_amd", "rnabloom"))
    rng = np.random.default_rng(5)
    og = rbo.Graph(300_017, 240_011, 100_003, 2, 2, 2, 25, False, True, 9)
    og.set_read_pair_distance(60)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 150 * 300)]
    og.add_reads(seq, None, np.arange(0, 150 * 301, 150, dtype=np.int64), 3, rbo.STORE_READ_PAIRS)

    def fold_bytes(data):           # the text of rnabloom.graph.fold_bytes (that module needs the HIP library to import)
        a = np.ascontiguousarray(data, np.uint8)
        pad = (-a.size) % 4
        if pad: a = np.concatenate([a, np.zeros(pad, np.uint8)])
        w = a.view("<u4").astype(np.uint64)
        nz = np.nonzero(w)[0]
        with np.errstate(over="ignore"):
            z = nz.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + w[nz]
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return int((z ^ (z >> np.uint64(31))).sum(dtype=np.uint64))
    assert og.folds() == tuple(fold_bytes(x) for x in (og.dbgbf_bytes(), og.cbf_bytes(), og.rpkbf_bytes()))
    assert all(og.folds())
