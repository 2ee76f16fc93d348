"""The smaller pieces of the reference's filter / graph API: PairedKeysBloomFilter,
BloomFilter.getOptimalSize, CountingBloomFilter.incrementAndGet / getBloomFilter, destroy*() of the graph's filters,
restorePkbf with another size, Kmer.has* / getNum*, Float.toString in the .desc files."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import rbo
from rnabloom import _native as N
from rnabloom.bloom import BloomFilter, CountingBloomFilter, PairedKeysBloomFilter
from rnabloom.graph import BloomFilterDeBruijnGraph, _java_float


def hashes(n, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 1 << 63, n, dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, n).astype(np.uint64)


def test_paired_keys_bloom_filter_is_a_bloom_filter_on_pair_hashes():
    L = rbo.lib()
    size, nh, k = 300_007, 3, 25
    pk = PairedKeysBloomFilter(size, nh, k)
    ob = L.rbo_bloom_new(size, nh)
    h = hashes(40_000, 1)
    h[1000:1200] = h[:200]                                           # repeats inside one call: array order decides
    exp = np.zeros(h.size, bool)
    for i, x in enumerate(h):
        hv = rbo.ntm64(int(x), k, nh)
        exp[i] = bool(L.rbo_bloom_lookup_then_add(ob, rbo._p(hv)))  # PairedKeysBloomFilter.lookupThenAdd :160-170
    got = pk.lookupThenAdd(h)
    assert (got == exp).all() and exp[1000:1200].all() and not exp[:200].all()
    n = C.c_int64()
    p = L.rbo_bloom_bytes(ob, C.byref(n))
    assert (pk.toBytes() == np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,))).all()
    assert pk.lookup(h).all() and pk.getNumhash() == nh
    more = hashes(5000, 2)
    pk.add(more)                                                     # :133-141
    assert pk.lookup(more).all()
    fpr = pk.getFPR()
    pop = pk.getPopCount()
    assert fpr == np.float32((pop / size) ** nh)
    assert pk.getOptimalSize(0.01) == BloomFilter.getExpectedSize(pop, 0.01, nh)        # :218-226
    assert PairedKeysBloomFilter.getExpectedSize(10 ** 6, 0.01, 2) == 18982444
    fresh = BloomFilter(1000, 2, 25)
    assert fresh.getOptimalSize(0.01) == 1000                        # popcount unknown (-1): the size itself
    pk.empty()
    assert pk.getPopCount() == pop and pk.getFPR() == 0 and pk.getPopCount() == 0     # the cached count moves with getFPR() only
    L.rbo_bloom_free(ob)


def test_increment_and_get_matches_oracle_in_array_order():
    L = rbo.lib()
    size, nh, k, seed = 5003, 2, 11, 9
    cbf = CountingBloomFilter(size, nh, k, rngSeed=seed)
    oc = L.rbo_cbf_new(size, nh)
    rng = np.random.default_rng(3)
    h = hashes(60, 4)[rng.integers(0, 60, 6000)]                     # few keys, many repeats: counters climb past 16
    ordinal = 0
    for chunk in np.array_split(h, 7):
        got = cbf.incrementAndGet(chunk)
        exp = np.zeros(chunk.size, np.float32)
        for i, x in enumerate(chunk):
            hv = rbo.ntm64(int(x), k, nh)
            exp[i] = L.rbo_cbf_increment_and_get(oc, rbo._p(hv), L.rbo_rng31(seed, ordinal, 0))
            ordinal += 1
        assert (got == exp).all()
    n = C.c_int64()
    p = L.rbo_cbf_bytes(oc, C.byref(n))
    raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,)).copy()
    assert (cbf.toBytes() == raw).all() and raw.max() > 20
    # getBloomFilter(minCov) :328-338
    val = np.array([float(b) if b <= 7 else float(((b & 7) | 8) << ((b >> 3) - 1)) for b in range(128)], np.float32)
    for min_cov in (1, 3, 17, 40):
        bf = cbf.getBloomFilter(min_cov)
        assert (bf.toBytes() == np.packbits(val[raw] >= min_cov, bitorder="little")).all()
        assert bf.device == cbf.device and bf.getNumHash() == nh
    L.rbo_cbf_free(oc)


def test_destroy_filters_and_restore_pkbf_with_another_size(tmp_path):
    g = BloomFilterDeBruijnGraph(100_003, 200_003, 30_011, 2, 2, 2, 25, False, True)
    g.setReadPairedKmerDistance(20)
    g.initializePairKmersBloomFilter(50_021, 3); g.setFragPairedKmerDistance(30)
    h = hashes(3000, 5)
    g.add(h); g.addReadSingleKmerPair(h[:500]); g.addFragmentSingleKmerPair(h[500:900])
    path = tmp_path / "rnabloom.graph"
    g.save(path); g.savePkbf(path)
    assert (tmp_path / "rnabloom.graph.fpkbf.desc").read_text().startswith("size:50021\nnumhash:3\nfpr:")
    # another graph whose live fpkbf has a different size and numhash: restorePkbf rebuilds it from the files (:341-350)
    g2 = BloomFilterDeBruijnGraph(100_003, 200_003, 30_011, 2, 2, 2, 25, False, True)
    g2.initializePairKmersBloomFilter(7001, 1)
    g2.restorePkbf(path)
    assert g2.filterSize(N.FPKBF)[0] == 50_021 and g2.filterSize(N.FPKBF)[2] == 3 and g2.getPkbfNumHash() == 3
    assert (g2.exportFilter(N.FPKBF) == g.exportFilter(N.FPKBF)).all()
    assert g2.lookupFragmentKmerPair(h[500:900]).all()
    # destroy: the views disappear, save() no longer writes the filter, the memory is gone
    g.destroyRpkbf()
    assert g.getRpkbf() is None and g.getFpkbf() is not None
    g.destroyFpkbf()
    assert g.getFpkbf() is None
    with pytest.raises(N.NativeError): g.lookupReadKmerPair(h[:5])
    g.initializePairKmersBloomFilter(1009, 2)                        # :352-359 creates it again
    assert g.filterSize(N.FPKBF)[0] == 1009 and g.popcount(N.FPKBF) == 0
    assert g.contains(h).all()
    g.destroyCbf()
    with pytest.raises(N.NativeError): g.getCount(h[:5])
    with pytest.raises(N.NativeError): g.add(h[:5])
    assert g.contains(h).all()                                       # dbgbf is still there
    g.destroyDbgbf()
    with pytest.raises(N.NativeError): g.contains(h[:5])
    # fromFile keeps the metadata of the file
    g3 = BloomFilterDeBruijnGraph.fromFile(path)
    assert g3.getPkbfNumHash() == 3 and g3.getMaxNumHash() == 2 and g3.getFragPairedKmerDistance() == 30
    assert (g3.exportFilter(N.FPKBF) == g2.exportFilter(N.FPKBF)).all()


def test_java_float_to_string():
    for x, want in ((1e-5, "1.0E-5"), (0.01, "0.01"), (0.5, "0.5"), (1.0, "1.0"), (3.0e-4, "3.0E-4"), (0.001, "0.001"),
                    (1.0e7, "1.0E7"), (123456.7, "123456.7"), (0.0, "0.0"), (2.5e-10, "2.5E-10"), (0.33333334, "0.33333334")):
        assert _java_float(np.float32(x)) == want


def test_kmer_neighbour_predicates():
    from rnabloom import synth
    d = synth.generate_pairs(600, G=2500, err=0.003, n_rate=0.0, seed=77)
    og = rbo.Graph(300_007, 400_009, 64, 2, 2, 1, 25, False, False, 3)
    gg = BloomFilterDeBruijnGraph(300_007, 400_009, 64, 2, 2, 1, 25, False, False, rngSeed=3)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    og.add_reads(s, q, off, 3, 0); gg.addReads(s, q, off, 3)
    branching = 0
    for ri in range(3, 40):
        read = bytes(s[off[ri]:off[ri + 1]])
        if b"N" in read: continue
        f, r, _ = og.get_kmers(read)
        first = np.frombuffer(read[:f.size], np.uint8); last = np.frombuffer(read[24:24 + f.size], np.uint8)
        nsucc = gg.getNumSuccessors(f, r, first); npred = gg.getNumPredecessors(f, r, last)
        for i in range(0, f.size, 5):
            _, _, oc = og.neighbors(f[i], r[i], int(first[i]), 0)
            assert nsucc[i] == int((oc > 0).sum())                      # graph.contains(hVals) <=> getCount > 0
            _, _, oc = og.neighbors(f[i], r[i], int(last[i]), 1)
            assert npred[i] == int((oc > 0).sum())
        assert (gg.hasSuccessors(f, r, first) == (nsucc > 0)).all() and (gg.hasAtLeastXPredecessors(f, r, last, 2) == (npred >= 2)).all()
        branching += int((nsucc >= 2).sum() + (npred >= 2).sum())
    assert branching > 0


@pytest.mark.parametrize("stranded", [False, True])
def test_naive_extension_and_extend_once_match_the_restatement(stranded):
    """GraphUtils.naiveExtendRight / Left (three forms each) and greedyExtendRightOnce / LeftOnce against step-by-step
    restatements over the oracle graph (oracle/rbo.py::naive_extend, ::greedy_extend)"""
    from rnabloom import synth
    d = synth.generate_pairs(1500, G=6000, err=0.004, n_rate=5e-4, seed=58)
    og = rbo.Graph(400_009, 700_001, 64, 2, 2, 1, 25, stranded, False, 3)
    gg = BloomFilterDeBruijnGraph(400_009, 700_001, 64, 2, 2, 1, 25, stranded, False, rngSeed=3)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    og.add_reads(s, q, off, 3, 0); gg.addReads(s, q, off, 3)
    rng = np.random.default_rng(4)
    reads = [bytes(s[off[i]:off[i + 1]]) for i in range(200)]
    seeds, frags = [], []
    for sq in reads:
        if b"N" in sq: continue
        p = int(rng.integers(0, len(sq) - 60))
        seeds.append(sq[p:p + 25]); frags.append(sq[p:p + 60])
    seeds += [b"ACGTNACGTACGTACGTACGTACGT", b"A" * 25]
    frags += [b"", b"A" * 30]
    reasons = set()
    for direction in (0, 1):
        for mode, kw in ((1, dict(bound=12)), (1, dict(bound=0)), (2, dict(bound=20)), (0, dict(cap=30)), (1, dict(bound=40, minKmerCov=2.0))):
            sd = seeds if direction == 0 else [f[-25:] if len(f) >= 25 else sdd for f, sdd in zip(frags, seeds)]
            got, why = gg.naiveExtend(sd, direction, mode, terminators=frags if mode == 0 else None, **kw)
            for i, km in enumerate(sd):
                eb, er = rbo.naive_extend(og, km, direction, mode, bound=kw.get("bound", 0), min_cov=kw.get("minKmerCov", 1.0),
                                          terminators=frags[i] if mode == 0 else b"", cap=kw.get("cap", 4096))
                assert (got[i], int(why[i])) == (eb, er), (direction, mode, kw, i, km)
                reasons.add(er)
    assert {0, 1, 2, 3, 4, 5}.issubset(reasons), reasons
    # a dead end that also has a back branch: the reference computes the neighbours first and never enters its loop
    # (R/util/GraphUtils.java:6789-6791), so the walk ends for lack of neighbours (reason 0), not at the back branch (1)
    acgt, rng2, dead = b"ACGT", np.random.default_rng(77), []
    for _ in range(40):
        x = bytes(np.frombuffer(acgt, np.uint8)[rng2.integers(0, 4, 25)])
        xl = bytes([acgt[(acgt.index(x[0]) + 1) % 4]]) + x[1:]          # left variant: back branch of a right walk
        xr = x[:-1] + bytes([acgt[(acgt.index(x[-1]) + 1) % 4]])        # right variant: back branch of a left walk
        three = np.frombuffer(x + xl + xr, np.uint8)
        og.add_reads(three, np.full(75, 73, np.uint8), np.array([0, 25, 50, 75], np.int64), 3, 0)
        gg.addReads(three, np.full(75, 73, np.uint8), np.array([0, 25, 50, 75], np.int64), 3)
        dead.append(x)
    hit = 0
    for direction in (0, 1):
        got, why = gg.naiveExtend(dead, direction, 1, bound=5)
        for i, km in enumerate(dead):
            eb, er = rbo.naive_extend(og, km, direction, 1, bound=5)
            assert (got[i], int(why[i])) == (eb, er), (direction, i, km)
            hit += er == 0 and eb == b""
    assert hit >= 40
    # greedyExtendRightOnce / LeftOnce = one step of the greedy extension
    clean = [km for km in seeds if b"N" not in km]
    for direction in (0, 1):
        for la in (0, 3):
            bases, cnt = gg.greedyExtendOnce(clean, direction, la)
            for i in range(0, len(clean), 7):
                eb, ec = rbo.greedy_extend(og, clean[i], direction, la, 1, stranded=stranded)
                assert bases[i] == eb and (not eb or float(cnt[i]) == ec[0])
