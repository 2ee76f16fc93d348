"""The oracle's sketch iterators (oracle/rb_oracle_sketch.c) against brute-force Python restatements of the same
reference statements and against the identities the reference's own main() methods rely on.  CPU only."""
import numpy as np
import pytest

from oracle import rbo

M64 = (1 << 64) - 1
SEQ = (b"TCGAATCCGTCTGATGCCTGACTGTAGCTGCGACTGATCGTAGCTAGCGACGAGCAGTCGCCCCATCGTACGTAGTCATGCATGCATGCATGCAGTACTATCTGCACACATGA"
       b"TGCATGCAATCTATATATTTTTATAT")     # the sequence of StrobeHashIterator.main / CanonicalStrobe3HashIterator.main


def comb(a, b):          # HashFunction.combineHashValues R/bloom/hash/HashFunction.java:260-263
    return (a ^ ((b + 0xFFFFFFFF9E3779B9 + ((a << 6) & M64) + (b >> 2)) & M64)) & M64


def signed(x):
    return x - (1 << 64) if x >> 63 else x


def revcomp(s):
    return s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))


def rand_seq(n, seed):
    rng = np.random.default_rng(seed)
    return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tolist())


def fwd(seq, k):
    h, _ = rbo.hash_region(seq, k, 1, rbo.FWD if hasattr(rbo, "FWD") else 0)
    return [int(x) for x in h[:, 0]]


def fr(seq, k):
    _, f = rbo.hash_region(seq, k, 1, 1)
    return [int(x) for x in f[:, 0]], [int(x) for x in f[:, 1]]


@pytest.mark.parametrize("n,wmin,wmax", [(2, 12, 50), (3, 12, 50), (3, 12, 30), (4, 5, 9)])
def test_randstrobes_next_and_get(n, wmin, wmax):
    k = 11
    seq = SEQ + rand_seq(300, 5)
    f = fwd(seq, k)
    nk = len(f)
    for slide in (False, True):
        h, pos = rbo.randstrobes(seq, k, n, wmin, wmax, slide=slide)
        assert len(h) == (nk - wmax * (n - 2) - wmin if nk > wmax * (n - 1) else 0)
        for p in range(0, len(h), 7):
            sh, ps = f[p], [p]
            for s in range(n - 1):
                lo, hi = p + s * wmax + wmin, min(p + s * wmax + wmax, nk)
                best, bk, bh = lo, f[lo], comb(sh, f[lo])
                for i in range(lo + 1, hi):
                    if slide and f[i] == bk:
                        best = i
                    else:
                        h2 = comb(sh, f[i])
                        if bh >= h2:
                            best, bk, bh = i, f[i], h2
                sh = bh; ps.append(best)
            assert int(h[p]) == sh and list(pos[p]) == ps


def test_canonical_randstrobes_properties():
    k, n, wmin, wmax = 11, 3, 12, 50
    seq = SEQ + rand_seq(500, 9)
    f, r = fr(seq, k)
    h, pos = rbo.randstrobes(seq, k, n, wmin, wmax, canonical=True)
    hf, posf = rbo.randstrobes(seq, k, n, wmin, wmax)
    assert (pos == posf).all()                       # strobes are chosen on the forward hashes
    for p in range(0, len(h), 5):
        rs = r[pos[p][n - 1]]
        for s in range(n - 2, -1, -1):
            rs = comb(r[pos[p][s]], rs)
        assert signed(int(h[p])) == min(signed(int(hf[p])), signed(rs))       # Math.min: signed (:107)
    # (no reverse-complement identity exists for this class: the strobes are chosen on the forward strand only, so the
    # hash sets of a sequence and of its reverse complement need not meet — CanonicalStrobeHashIterator.main only prints them)


@pytest.mark.parametrize("canonical", [False, True])
def test_strobe3(canonical):
    k, wmin, wmax = 11, 12, 50
    seq = SEQ + rand_seq(400, 3)
    f, r = fr(seq, k)
    nk = len(f)
    h, pos = rbo.strobe3(seq, k, wmin, wmax, canonical)
    mn = wmax if canonical else wmin
    mx = nk - 1 - (wmax if canonical else wmin)
    assert len(h) == max(0, mx + 1 - mn)

    def arg(vals, strict):            # first index of the unsigned minimum (strict: keep the first; else the last)
        best = 0
        for i in range(1, len(vals)):
            if (vals[best] > vals[i]) if strict else (vals[best] >= vals[i]):
                best = i
        return best
    for p in range(mn, mx + 1, 3):
        lo1 = max(0, p - wmax + 1)
        c1 = [comb(f[i], f[p]) for i in range(lo1, p - wmin + 1)]
        i1 = arg(c1, True); h1 = c1[i1]
        c3 = [comb(h1, f[i]) for i in range(p + wmin, min(p + wmax, nk))]
        i3 = arg(c3, not canonical); h3 = c3[i3]
        exp_h, exp_p = h3, [lo1 + i1, p, p + wmin + i3]
        if canonical:
            d3 = [comb(r[i], r[p]) for i in range(p + wmin, min(p + wmax, nk))]
            j3 = arg(d3, False); rh3 = d3[j3]
            d1 = [comb(rh3, r[i]) for i in range(lo1, p - wmin + 1)]
            j1 = arg(d1, True); rh1 = d1[j1]
            if h3 > rh1:
                exp_h, exp_p = rh1, [lo1 + j1, p, p + wmin + j3]
        assert int(h[p - mn]) == exp_h and list(pos[p - mn]) == exp_p
    # short sequences: numKmers <= 2*wMin -> no strobemers; canonical needs 2*wMax k-mers to produce any
    assert len(rbo.strobe3(seq[:k + 2 * wmin - 1], k, wmin, wmax, canonical)[0]) == 0
    assert len(rbo.strobe3(seq[:k + 2 * wmax - 1], k, wmin, wmax, True)[0]) == 0


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_minimizer_set_and_next(mode):
    k, w = 13, 15
    for seed, n in ((1, 400), (2, 27), (3, 28), (4, 20), (5, 12)):
        seq = rand_seq(n, seed)
        nk = n - k + 1
        hv = [signed(int(x)) for x in rbo.hash_region(seq, k, 1, mode)[0][:, 0]] if nk > 0 else []
        got = [signed(int(x)) for x in rbo.minimizer_set(seq, k, w, mode)]
        if nk <= w:
            assert got == [min([0] + hv)]                       # stale seed of a fresh iterator (GraphUtils.java:2480-2488)
            st = -(1 << 62)
            assert [signed(int(x)) for x in rbo.minimizer_set(seq, k, w, mode, stale=st & M64)] == [min([st] + hv)]
        else:
            assert got == sorted(set(min(hv[i:i + w]) for i in range(nk - w + 1)))
        nh, npos = rbo.minimizers_next(seq, k, w, mode)
        wh, wp = rbo.minimizers(seq, k, w, mode)
        exp = [(int(wh[i]), int(wp[i])) for i in range(len(wh)) if i == 0 or wp[i] != wp[i - 1]]
        assert [(int(a), int(b)) for a, b in zip(nh, npos)] == exp
        assert all(npos[i] < npos[i + 1] for i in range(len(npos) - 1))


@pytest.mark.parametrize("canonical", [False, True])
def test_kmer_pair_hashes(canonical):
    k, shift = 11, 12
    seq = rand_seq(200, 7)
    f, r = fr(seq, k)
    got = rbo.kmer_pair_hashes(seq, k, shift, canonical)
    assert len(got) == len(f) - shift
    for i in range(len(got)):
        pf = comb(f[i], f[i + shift])
        exp = pf if not canonical else (pf if signed(pf) <= signed(comb(r[i + shift], r[i])) else comb(r[i + shift], r[i]))
        assert int(got[i]) == exp
    if canonical:                                               # invariant under reverse complement (pair order flips)
        back = rbo.kmer_pair_hashes(revcomp(seq), k, shift, True)
        assert [int(x) for x in back] == [int(x) for x in got[::-1]]


# ---- three-way check, CPU leg: the C restatement against the independent Python one (oracle/rbo_py.py: hashes from scratch,
#      brute-force argmin, a literal LongRollingWindow) on reads with everything that makes the rules matter: letters outside
#      ACGTU (reverse-strand seed by `ch & 7`), runs of one base and short tandem repeats (equal k-mer hashes: ties) ----
def nasty_reads(seed, n=14):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(90, 260))
        s = acgt[rng.integers(0, 4, L)].copy()
        for _ in range(int(rng.integers(0, 4))):                 # homopolymers / tandem repeats: windows full of equal hashes
            a = int(rng.integers(0, L - 40)); unit = acgt[rng.integers(0, 4, int(rng.integers(1, 4)))]
            rep = np.tile(unit, 40)[:int(rng.integers(20, 40))]
            s[a:a + rep.size] = rep[:max(0, min(rep.size, L - a))]
        if i % 2:                                                # IUPAC and other letters, lower case, U
            for ch in b"KMSWYIELOQDRBHVNXacgtu":
                if rng.random() < 0.5: s[int(rng.integers(0, L))] = ch
        out.append(s.tobytes())
    return out + [b"A" * 120, b"ACACACACAC" * 14, b"ACGT", b""]


def test_python_restatement_agrees_with_the_c_oracle():
    from oracle import rbo_py as P
    for s in nasty_reads(3):
        for mode in (0, 1, 2):
            eh, ep = rbo.minimizers(s, 13, 15, mode)
            got = P.minimizers(s, 13, 15, mode)
            assert [int(x) for x in eh] == [h for h, _ in got] and [int(x) for x in ep] == [p for _, p in got], (s, mode)
            nh, npos = rbo.minimizers_next(s, 13, 15, mode)
            assert list(zip([int(x) for x in nh], [int(x) for x in npos])) == P.minimizers_next(s, 13, 15, mode)
            assert [int(x) for x in rbo.minimizer_set(s, 13, 15, mode, stale=7)] == P.minimizer_set(s, 13, 15, mode, stale=7)
        oh, os_, oe = rbo.strobemers(s, 11, 3, 12, 31)
        assert list(zip([int(x) for x in oh], [int(x) for x in os_], [int(x) for x in oe])) == P.strobemer_intervals(s, 11, 3, 12, 31)
        for canonical, slide in ((False, False), (False, True), (True, False)):
            rh, rp = rbo.randstrobes(s, 11, 3, 5, 20, canonical=canonical, slide=slide)
            want = P.randstrobes(s, 11, 3, 5, 20, canonical)
            assert [int(x) for x in rh] == [h for h, _ in want] and [list(map(int, r)) for r in rp] == [p for _, p in want], (s, canonical, slide)
        for canonical in (False, True):
            th, tp = rbo.strobe3(s, 11, 6, 25, canonical=canonical)
            want = P.strobe3(s, 11, 6, 25, canonical)
            assert [int(x) for x in th] == [h for h, _ in want] and [list(map(int, r)) for r in tp] == [p for _, p in want], (s, canonical)
            assert [int(x) for x in rbo.kmer_pair_hashes(s, 13, 9, canonical)] == P.kmer_pair_hashes(s, 13, 9, canonical)
