"""GPU parity tests: the HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.
Bit-exact bar: hash values, dbgbf / rpkbf / fpkbf bytes, cbf bytes, popcounts, counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import rbo
from rnabloom import _native as N
from rnabloom import synth
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch


def make_reads(n_pairs, G, err, n_rate, seed, sigma=2.0, uniform=False):
    d = synth.generate_pairs(n_pairs, G=G, err=err, n_rate=n_rate, seed=seed, sigma=sigma, uniform_expr=uniform)
    ls, off = synth.flat(d["left"]); lq, _ = synth.flat(d["lqual"])
    rs, _ = synth.flat(d["right"]); rq, _ = synth.flat(d["rqual"])
    return (ls, lq, off), (rs, rq, off)


def ragged_reads(seed, n=400):
    rng = np.random.default_rng(seed)
    reads, quals = [], []
    for i in range(n):
        L = int(rng.choice([0, 1, 10, 24, 25, 26, 31, 32, 33, 57, 64, 65, 100, 150, 151, 300, 1000]))
        s = np.frombuffer(b"ACGTacgtUuNRY", np.uint8)[rng.choice(13, L, p=[.2, .2, .2, .2, .03, .03, .03, .03, .01, .01, .03, .02, .01])]
        q = np.where(rng.random(L) < 0.03, ord("#"), ord("I")).astype(np.uint8)
        q[rng.random(L) < 0.01] = ord("$")
        reads.append(s.tobytes()); quals.append(q.tobytes())
    return reads, quals


@pytest.mark.parametrize("k", [1, 5, 25, 32, 33, 35, 64, 65, 130])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_nthash_batch_matches_iterators(k, mode):
    reads, quals = ragged_reads(k * 3 + mode)
    b = ReadBatch.from_reads(reads, quals, 3)
    h0, rd, ps = b.nthash(k, mode, with_positions=True)
    exp_h, exp_r, exp_p = [], [], []
    for i, (s, q) in enumerate(zip(reads, quals)):
        if len(s) < k:
            continue
        for a, e in rbo.segments(s, q, k, 3):
            h, _ = rbo.hash_region(s, k, 1, mode, a, e)
            exp_h.append(h[:, 0]); exp_r.append(np.full(len(h), i)); exp_p.append(np.arange(a, a + len(h)))
    exp_h = np.concatenate(exp_h) if exp_h else np.zeros(0, np.uint64)
    assert len(h0) == len(exp_h)
    assert (h0 == exp_h).all()
    if len(h0):
        assert (rd == np.concatenate(exp_r)).all() and (ps == np.concatenate(exp_p)).all()


def test_batch_roundtrip_download():
    reads, quals = ragged_reads(99, 100)
    b = ReadBatch.from_reads(reads, None)
    seq, off = b.download()
    for i, s in enumerate(reads):
        got = bytes(seq[off[i]:off[i + 1]])
        exp = bytes(c if c in b"ACGT" else (ord("T") if c in b"Uu" else (c - 32 if c in b"acgt" else ord("N"))) for c in s)
        assert got == exp


def graph_pair(dbg_bits, cbf_bytes, pk_bits, k=25, stranded=False, dbg_h=2, cbf_h=2, pk_h=2, seed=7, pairs=True,
               max_batch=0, group_bits=0):
    og = rbo.Graph(dbg_bits, cbf_bytes, pk_bits, dbg_h, cbf_h, pk_h, k, stranded, pairs, seed)
    gg = BloomFilterDeBruijnGraph(dbg_bits, cbf_bytes, pk_bits, dbg_h, cbf_h, pk_h, k, stranded, pairs,
                                  rngSeed=seed, maxBatchKmers=max_batch, groupBits=group_bits)
    return og, gg


def assert_same_state(og, gg, pairs=True):
    assert (gg.exportFilter(N.DBGBF) == og.dbgbf_bytes()).all(), "dbgbf differs"
    c_g, c_o = gg.exportFilter(N.CBF), og.cbf_bytes()
    bad = np.nonzero(c_g != c_o)[0]
    assert bad.size == 0, "cbf differs at %d bytes, first %s: gpu %s oracle %s" % (bad.size, bad[:5], c_g[bad[:5]], c_o[bad[:5]])
    if pairs:
        assert (gg.exportFilter(N.RPKBF) == og.rpkbf_bytes()).all(), "rpkbf differs"
    pc = og.popcounts()
    assert gg.popcount(N.DBGBF) == pc[0] and gg.popcount(N.CBF) == pc[1]
    if pairs:
        assert gg.popcount(N.RPKBF) == pc[2]


@pytest.mark.parametrize("case", [
    # (n_pairs, G, err, sizes(dbg,cbf,pk), max_batch)   small filters => many collisions/conflicts
    dict(n=2000, G=30000, err=0.002, sizes=(400_003, 3_000_017, 90_001), mb=0),
    dict(n=2000, G=30000, err=0.002, sizes=(60_013, 200_003, 9_001), mb=0),          # heavy collisions
    dict(n=3000, G=8000, err=0.0, sizes=(300_007, 2_000_003, 70_001), mb=50_000),    # deep coverage, many sub-batches
    dict(n=1500, G=50000, err=0.01, sizes=(1_000_003, 8_000_009, 200_003), mb=0),
])
def test_stage1_paired_end_bit_exact(case):
    (ls, lq, off), (rs, rq, _) = make_reads(case["n"], case["G"], case["err"], 1e-3, seed=case["n"] + case["G"])
    og, gg = graph_pair(*case["sizes"], max_batch=case["mb"])
    d = 150 - 25 - 10
    og.set_read_pair_distance(d); gg.setReadPairedKmerDistance(d)
    # forward file first, then the reverse-complemented file (R/RNABloom.java:1301-1311)
    so = og.add_reads(ls, lq, off, 3, rbo.STORE_READ_PAIRS)
    sg = gg.addReads(ls, lq, off, 3, storeReadPairedKmers=True)
    assert (sg.kmers, sg.pairs, sg.reads) == (so.kmers, so.pairs, so.reads + so.reads_skipped)
    assert_same_state(og, gg)
    so = og.add_reads(rs, rq, off, 3, rbo.STORE_READ_PAIRS | rbo.REVCOMP)
    sg = gg.addReads(rs, rq, off, 3, reverseComplement=True, storeReadPairedKmers=True)
    assert (sg.kmers, sg.pairs) == (so.kmers, so.pairs)
    assert_same_state(og, gg)
    assert gg.getOpOrdinal() == 2 * case["n"]
    fo = og.fprs()
    assert gg.getDbgbfFPR() == fo[0] and gg.getCbfFPR() == fo[1] and gg.getRpkbfFPR() == fo[2]


@pytest.mark.parametrize("cbf_bytes,cbf_h,mb,old_rule", [(2_003, 2, 0, False), (2_003, 4, 20_000, False), (40_009, 3, 0, False), (313, 2, 7_000, False),
                                                         (40_009, 3, 0, True)])
def test_shared_counters_everywhere(monkeypatch, cbf_bytes, cbf_h, mb, old_rule):
    """counting filters so small that nearly every run shares counters with others, chains of sharing included: the ordered
    set (runs that can reach a shared counter somebody else can reach, closed over runs whose own bound no longer holds —
    k_cs_writers / k_cs_order) must give the sequential result; deep coverage takes the counters far into the
    probabilistic range, heavy runs and many sub-batches included.  old_rule: every run with a shared counter is replayed."""
    if old_rule: monkeypatch.setenv("RB_ORDER_ALL_SHARED", "1")
    (ls, lq, off), (rs, rq, _) = make_reads(2500, 6000, 0.002, 1e-3, seed=cbf_bytes + cbf_h)
    og, gg = graph_pair(500_009, cbf_bytes, 50_021, cbf_h=cbf_h, max_batch=mb)
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    og.add_reads(ls, lq, off, 3, rbo.STORE_READ_PAIRS); gg.addReads(ls, lq, off, 3, storeReadPairedKmers=True)
    assert_same_state(og, gg)
    og.add_reads(rs, rq, off, 3, rbo.STORE_READ_PAIRS | rbo.REVCOMP); gg.addReads(rs, rq, off, 3, reverseComplement=True, storeReadPairedKmers=True)
    assert_same_state(og, gg)
    assert og.cbf_bytes().max() >= 40


@pytest.mark.parametrize("stranded", [False, True])
@pytest.mark.parametrize("h", [(1, 1, 1), (3, 2, 1), (2, 4, 3)])
def test_stage1_hash_counts_and_strandedness(stranded, h):
    (ls, lq, off), (rs, rq, _) = make_reads(800, 20000, 0.003, 1e-3, seed=31)
    og, gg = graph_pair(500_009, 1_000_003, 50_021, k=31, stranded=stranded, dbg_h=h[0], cbf_h=h[1], pk_h=h[2])
    og.set_read_pair_distance(60); gg.setReadPairedKmerDistance(60)
    og.add_reads(ls, lq, off, 3, rbo.STORE_READ_PAIRS)
    gg.addReads(ls, lq, off, 3, storeReadPairedKmers=True)
    og.add_reads(rs, rq, off, 3, rbo.STORE_READ_PAIRS | rbo.REVCOMP)
    gg.addReads(rs, rq, off, 3, reverseComplement=True, storeReadPairedKmers=True)
    assert_same_state(og, gg)


def test_high_multiplicity_probabilistic_regime():
    # 4 kb "transcriptome" at ~500x: most counters go far beyond 16, exercising the shared RNG,
    # the wave-per-k-mer heavy path and ordered conflict replay
    (ls, lq, off), _ = make_reads(6000, 4000, 0.001, 1e-3, seed=77, uniform=True)
    og, gg = graph_pair(100_003, 150_001, 20_011)
    og.add_reads(ls, lq, off, 3, 0)
    st = gg.addReads(ls, lq, off, 3)
    assert st.conflict_ops > 0
    assert_same_state(og, gg, pairs=False)
    assert og.cbf_bytes().max() > 24


@pytest.mark.parametrize("group_bits", [64, 20, 8, 1])
def test_grouping_prefix_does_not_change_results(group_bits):
    # fewer grouping bits => the same hash appears as several runs ("split runs"); results must not move
    (ls, lq, off), _ = make_reads(2500, 6000, 0.002, 1e-3, seed=13)
    og, gg = graph_pair(150_001, 400_009, 20_011, group_bits=group_bits)
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    og.add_reads(ls, lq, off, 3, rbo.STORE_READ_PAIRS)
    gg.addReads(ls, lq, off, 3, storeReadPairedKmers=True)
    assert_same_state(og, gg)
    assert og.cbf_bytes().max() > 16


def test_noop_prefilter_many_sub_batches():
    # deep coverage + tiny sub-batches: the hot-k-mer cache is filled by early sub-batches and drops
    # provably ineffective occurrences of later ones; results must not move
    (ls, lq, off), (rs, rq, _) = make_reads(8000, 3000, 0.001, 1e-3, seed=91, uniform=True)
    og, gg = graph_pair(100_003, 170_003, 20_011, max_batch=40_000)
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    so = og.add_reads(ls, lq, off, 3, rbo.STORE_READ_PAIRS)
    sg = gg.addReads(ls, lq, off, 3, storeReadPairedKmers=True)
    assert sg.kmers == so.kmers and sg.pairs == so.pairs
    assert_same_state(og, gg)
    so = og.add_reads(rs, rq, off, 3, rbo.STORE_READ_PAIRS | rbo.REVCOMP)
    sg = gg.addReads(rs, rq, off, 3, reverseComplement=True, storeReadPairedKmers=True)
    assert sg.kmers == so.kmers
    assert_same_state(og, gg)
    assert og.cbf_bytes().max() > 40
    # incrementIfPresent path goes through the same prefilter
    so = og.add_reads(ls, lq, off, 3, rbo.COUNT_IF_PRESENT)
    sg = gg.addReads(ls, lq, off, 3, incrementIfPresent=True)
    assert sg.kmers == so.kmers
    assert_same_state(og, gg)
    # clearing must invalidate the cache
    og.clear(); gg.clearAllBf()
    og.add_reads(rs[: off[500]], rq[: off[500]], off[:501], 3, 0)
    gg.addReads(rs[: off[500]], rq[: off[500]], off[:501], 3)
    assert_same_state(og, gg, pairs=False)


def test_batch_partition_independence():
    (ls, lq, off), _ = make_reads(1500, 10000, 0.002, 1e-3, seed=5)
    _, g1 = graph_pair(200_003, 900_001, 20_011)
    _, g2 = graph_pair(200_003, 900_001, 20_011, max_batch=20_000)
    g1.addReads(ls, lq, off, 3)
    b = ReadBatch.from_ascii(ls, lq, off, 3)
    for first in range(0, 1500, 333):
        g2.addBatch(b, first=first, n=min(333, 1500 - first))
    for w in (N.DBGBF, N.CBF):
        assert (g1.exportFilter(w) == g2.exportFilter(w)).all()


def test_fasta_and_empty_inputs():
    og, gg = graph_pair(100_003, 300_007, 10_007, pairs=False)
    reads = [b"", b"ACGT", b"ACGTNNNN" * 10, b"A" * 200, b"ACGTTGCAAGGCTTAGCATCGATCGATTAGC" * 5, b"N" * 100]
    seq, _, off = rbo.pack_reads(reads)
    so = og.add_reads(seq, None, off, 3, 0)
    sg = gg.addReads(seq, None, off, 3)
    assert sg.kmers == so.kmers
    assert_same_state(og, gg, pairs=False)
    sg = gg.addReads(np.zeros(0, np.uint8), None, np.zeros(1, np.int64), 3)
    assert sg.kmers == 0 and sg.reads == 0
    assert_same_state(og, gg, pairs=False)


def test_per_hash_ops_match_oracle_sequence():
    rng = np.random.default_rng(3)
    og, gg = graph_pair(20_011, 30_011, 5_003)
    pool = rng.integers(0, 1 << 63, 300, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    for rnd, (op_o, op_g) in enumerate([("add", "add"), ("add_if_absent", "addIfAbsent"),
                                        ("add_count_if_present", "addCountIfPresent"), ("add", "add"),
                                        ("add_count_only", "addCountOnly"), ("add_dbg_only", "addDbgOnly"),
                                        ("add_if_absent", "addIfAbsent"), ("add_count_if_present", "addCountIfPresent")]):
        hs = pool[rng.integers(0, len(pool), 4000)]
        for h in hs:
            getattr(og, op_o)(rbo.ntm64(int(h), 25, 2))
        getattr(gg, op_g)(hs)
        assert_same_state(og, gg)
    q = pool[:200]
    exp_c = np.array([og.get_count(rbo.ntm64(int(h), 25, 2)) for h in q], np.float32)
    exp_b = np.array([og.contains(rbo.ntm64(int(h), 25, 2)) for h in q])
    assert (gg.getCount(q) == exp_c).all() and (gg.contains(q) == exp_b).all()
    ph = rng.integers(0, 1 << 62, 500, dtype=np.int64).astype(np.uint64)
    for h in ph[:250]:
        og.add_read_pair(rbo.ntm64(int(h), 25, 2))
    gg.addReadSingleKmerPair(ph[:250])
    assert (gg.exportFilter(N.RPKBF) == og.rpkbf_bytes()).all()
    exp = np.array([og.lookup_read_pair(rbo.ntm64(int(h), 25, 2)) for h in ph])
    assert (gg.lookupReadKmerPair(ph) == exp).all()


def test_queries_get_kmers_and_neighbors():
    (ls, lq, off), _ = make_reads(1500, 12000, 0.002, 1e-3, seed=21)
    for stranded in (False, True):
        og, gg = graph_pair(300_007, 2_000_003, 10_007, stranded=stranded, pairs=False)
        og.add_reads(ls, lq, off, 3, 0); gg.addReads(ls, lq, off, 3)
        reads = [bytes(ls[off[i]:off[i + 1]]) for i in range(0, 60)] + [b"ACGT", b"", b"ACGTN" * 30]
        ko, f, r, c = gg.getKmers(reads)
        for i, s in enumerate(reads):
            ef, er, ec = og.get_kmers(s)
            a, b = ko[i], ko[i + 1]
            assert b - a == len(ef)
            assert (f[a:b] == ef).all() and (c[a:b] == ec).all()
            if not stranded:
                assert (r[a:b] == er).all()
        s = reads[0]
        ef, er, _ = og.get_kmers(s)
        idx = np.arange(0, len(ef), 7)
        for direction in (0, 1):
            ch = np.array([s[i] if direction == 0 else s[i + 24] for i in idx], np.uint8)
            f4, r4, c4 = gg.getNeighbors(ef[idx], er[idx], ch, direction)
            for j, i in enumerate(idx):
                of, orr, oc = og.neighbors(ef[i], er[i], int(ch[j]), direction)
                assert (f4[j] == of).all() and (c4[j] == oc).all()
                if not stranded:
                    assert (r4[j] == orr).all()


def test_variants_match_oracle():
    (ls, lq, off), _ = make_reads(600, 9000, 0.002, 0.0, seed=8)
    for stranded in (False, True):
        og, gg = graph_pair(200_003, 900_001, 10_007, stranded=stranded, pairs=False)
        og.add_reads(ls, lq, off, 3, 0); gg.addReads(ls, lq, off, 3)
        s = bytes(ls[off[3]:off[4]])
        ef, er, _ = og.get_kmers(s)
        idx = np.arange(0, len(ef), 11)
        for side, direction in ((0, 2), (1, 3)):
            ch = np.array([s[i] if side == 0 else s[i + 24] for i in idx], np.uint8)
            f4, r4, c4 = gg.getNeighbors(ef[idx], er[idx], ch, direction)
            for j, i in enumerate(idx):
                for b, base in enumerate(b"ACGT"):
                    vf, vr, vh = rbo.variant(int(ef[i]), int(er[i]), int(ch[j]), base, 25, 2, not stranded, side)
                    assert int(f4[j, b]) == vf and c4[j, b] == og.get_count(vh)
                    if not stranded:
                        assert int(r4[j, b]) == vr


def test_export_import_roundtrip_and_errors():
    og, gg = graph_pair(100_003, 300_007, 10_007)
    (ls, lq, off), _ = make_reads(300, 5000, 0.0, 0.0, seed=2)
    gg.setReadPairedKmerDistance(115)
    gg.addReads(ls, lq, off, 3, storeReadPairedKmers=True)
    snap = [gg.exportFilter(w) for w in (N.DBGBF, N.CBF, N.RPKBF)]
    assert [len(s) for s in snap] == [(100_003 + 7) // 8, 300_007, (10_007 + 7) // 8]
    gg.clearAllBf()
    assert gg.popcount(N.DBGBF) == 0 and gg.popcount(N.CBF) == 0 and gg.getOpOrdinal() == 0
    for w, s in zip((N.DBGBF, N.CBF, N.RPKBF), snap):
        gg.importFilter(w, s)
        assert (gg.exportFilter(w) == s).all()
    with pytest.raises(N.NativeError):
        gg.importFilter(N.CBF, snap[0])                 # wrong size
    with pytest.raises(N.NativeError):
        gg.popcount(N.FPKBF)                            # fragment pair filter not initialised
    with pytest.raises(N.NativeError):
        BloomFilterDeBruijnGraph(0, 10, 10, 2, 2, 2, 25, False, True)
    gg.initializePairKmersBloomFilter(5003, 2)
    assert gg.popcount(N.FPKBF) == 0


def test_synthetic_generator_consistency():
    b = ReadBatch.synthetic(5000, 1 << 16, seed=9)
    assert b.n_reads == 10000
    seq, off = b.download()
    assert (np.diff(off) == 150).all()
    frac_n = (seq == ord("N")).mean()
    assert 0.0003 < frac_n < 0.003
    og, gg = graph_pair(2_000_003, 16_000_057, 400_009)
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    og.add_reads(seq, None, off, 3, rbo.STORE_READ_PAIRS)
    gg.addBatch(b, storeReadPairedKmers=True)
    assert_same_state(og, gg)


def test_synthetic_slices_are_the_whole_set():
    """rank slices (pair_offset/total_pairs) of the device generator reassemble the single-batch read set"""
    whole = ReadBatch.synthetic(3000, 1 << 16, seed=11)
    seq, off = whole.download()
    reads = seq.reshape(-1, 150)
    for G in (2, 3):
        per = 3000 // G
        for r in range(G):
            part = ReadBatch.synthetic(per, 1 << 16, seed=11, pair_offset=r * per, total_pairs=3000)
            ps, _ = part.download()
            pr = ps.reshape(-1, 150)
            assert (pr[:per] == reads[r * per:(r + 1) * per]).all(), "left reads differ"
            assert (pr[per:] == reads[3000 + r * per:3000 + (r + 1) * per]).all(), "right reads differ"
    with pytest.raises(N.NativeError):
        ReadBatch.synthetic(10, 1 << 16, pair_offset=5, total_pairs=12)


def test_cold_subbatch_is_split_when_survivors_exceed_the_bound():
    """a sub-batch may span 2x max_batch windows; when (cold cache) more than max_batch of them survive
    the prefilter it is halved and redone — results stay exact and every k-mer is counted once"""
    d = synth.generate_pairs(1200, G=60000, err=0.002, n_rate=1e-3, seed=33)      # low coverage: nothing to drop
    og, gg = graph_pair(900_001, 1_200_007, 50_021, max_batch=20_000)
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    o_st = og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS)
    st = gg.addReads(s, q, off, 3, storeReadPairedKmers=True)
    assert_same_state(og, gg)
    assert st.kmers == o_st.kmers and st.reads == 1200


@pytest.mark.parametrize("env", [{}, {"RB_NO_MPF": "1"}, {"RB_NO_RAMP": "1"}, {"RB_SERIAL": "1"},
                                 {"RB_MPF": "8"}, {"RB_NPF": "8", "RB_NO_MPF": "1"}, {"RB_MPF_M": "9"},
                                 {"RB_READ_LANES": "0"}, {"RB_SPARSE_EMIT": "0"}, {"RB_READ_LANES": "0", "RB_NO_MPF": "1"},
                                 {"RB_EMIT_RESUME": "0"}, {"RB_EMIT_RESUME": "0", "RB_SPARSE_EMIT": "0"},
                                 {"RB_FIRST_SETTER_TABLE": "1"}, {"RB_NO_CS_FILTER": "1"},
                                 {"RB_GROUP_ORDERED": "1"}, {"RB_GROUP_PREFETCH": "0"}, {"RB_GROUP_FIX": "0"}, {"RB_GROUP_FIX": "2"},
                                 {"RB_GROUP_T": "18"}, {"RB_GROUP_T": "18", "RB_GROUP_WIDE_LDS": "0"},
                                 {"RB_FILTER_PIPE": "1"}, {"RB_FILTER_PIPE": "0"}, {"RB_RAGGED_LANES": "0"},
                                 {"RB_GROUP_IDX": "1"}, {"RB_GROUP_IDX": "0"}, {"RB_GROUP_IDX": "1", "RB_GROUP_T": "18"}, {"RB_GROUP_IDX": "1", "RB_GROUP_T": "3"},
                                 {"RB_FT_FILTER": "1"}, {"RB_FT_FILTER": "0"}, {"RB_GROUP_CLASSES": "0"},
                                 {"RB_SWEEP": "1"}, {"RB_SWEEP": "1", "RB_GROUP_T": "18"}, {"RB_SWEEP": "1", "RB_GROUP_T": "3"}, {"RB_SWEEP": "1", "RB_GROUP_T": "11"},
                                 {"RB_SWEEP": "1", "RB_SERIAL": "1"}, {"RB_SWEEP": "1", "RB_GROUP_ORDERED": "1"}, {"RB_SWEEP": "1", "RB_NO_MPF": "1"},
                                 {"RB_SWEEP": "1", "RB_FT_FILTER": "1"}, {"RB_SWEEP": "1", "RB_PF_SKIP": "2"}, {"RB_SWEEP": "0"},
                                 {"RB_SMALL_COMPONENT_OPS": "1"}, {"RB_SMALL_COMPONENT_OPS": "1000000"}, {"RB_RELEASE_EARLY": "1"},
                                 {"RB_ASCII_PIECE": "30000"}, {"RB_ASCII_PIECE": "151"}, {"RB_ASCII_PIECE": "70000", "RB_SERIAL": "1"}, {"RB_ASCII_CHUNKED": "1"}, {"RB_NO_INGEST_POOL": "1"}])
def test_pipeline_switches_do_not_change_results(monkeypatch, env):
    """every scheduling / cache switch of the insert path (minimizer- vs hash-bucketed cache, tiny caches that
    thrash, minimizer length, cold-start ramp, producer one sub-batch ahead, serialised streams, one word or one
    read per lane in the prefilter, dense or sparse emit pass) is a pure performance choice: the filters must
    come out identical to the oracle's"""
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    d = synth.generate_pairs(2500, G=4000, err=0.002, n_rate=1e-3, seed=77, uniform_expr=True)
    og, gg = graph_pair(300_007, 400_009, 60_013, max_batch=15_000)
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    for name, rc in (("left", False), ("right", True)):
        s, off = synth.flat(d[name]); q, _ = synth.flat(d[name[0] + "qual"])
        og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
        st = gg.addReads(s, q, off, 3, reverseComplement=rc, storeReadPairedKmers=True)
        assert_same_state(og, gg)
    assert og.cbf_bytes().max() > 30          # deep into the probabilistic counter range
    assert st.sorted_kmers < st.kmers         # and the prefilter is doing something


@pytest.mark.parametrize("sizes", [(300_007, 300_007), (300_007, 400_009), (5_000_011, 5_000_011), (64 * 4096 * 8, 64 * 4096 * 8 + 1)])
@pytest.mark.parametrize("T", ["2", "7", "12"])
def test_swept_bloom_bit_stage_with_oversized_buckets_and_shared_words(monkeypatch, sizes, T):
    """the swept Bloom-bit stage (csrc/rb_group.hip sweep_bits_device, forced with RB_SWEEP=1): index ranges that share words with their
    neighbours (a few bits to a few thousand words per range), ranges longer than one 64 KB round, filters of equal and of different
    entries (the grouping then goes by the Bloom filter's indices), and — prefilter off, one read 6000 times — a bucket too large for
    LDS whose runs are appended out of range order: the filters come out as the oracle's"""
    monkeypatch.setenv("RB_SWEEP", "1"); monkeypatch.setenv("RB_GROUP_T", T); monkeypatch.setenv("RB_NO_MPF", "1")
    d = synth.generate_pairs(1500, G=6000, err=0.004, n_rate=1e-3, seed=int(T) + sizes[0] % 97, uniform_expr=True)
    og, gg = graph_pair(sizes[0], sizes[1], 60_013, max_batch=0)
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    reads = np.concatenate([d["left"], np.repeat(d["left"][3:4], 6000, axis=0), d["right"]])
    quals = np.concatenate([d["lqual"], np.repeat(d["lqual"][3:4], 6000, axis=0), d["rqual"]])
    for rep in range(2):                      # second pass: every k-mer present, every bit found set before the sub-batch
        s, off = synth.flat(reads); q, _ = synth.flat(quals)
        og.add_reads(s, q, off, 3, 0)
        gg.addReads(s, q, off, 3)
        assert_same_state(og, gg, pairs=False)
    assert og.cbf_bytes().max() > 40


@pytest.mark.parametrize("claims", ["0", "1"])
@pytest.mark.parametrize("T", ["3", "9"])
@pytest.mark.parametrize("size", [60_013, 200_003, 3_000_017])
def test_swept_stage_reads_unmet_counters_without_claiming_them(monkeypatch, size, T, claims):
    """round 5: behind the swept stage a run none of whose Bloom probes met another probe of the sub-batch reads its counters with plain loads
    instead of claiming them (filters of EQUAL size: probe j of the counting filter has probe j's Bloom-bit index; k_probe_h2<., true>).  Filters
    so small that most probes do meet somebody and counters are shared all over (60 013 entries), and sparser ones where most do not; deep
    coverage (counters well into the probabilistic range), many sub-batches, no oversized bucket (one would switch the shortcut off), both
    files of a library and a second pass in which every bit is found set; RB_SWEEP_CLAIMS=1 (everybody claims, round 4) gives the same bytes."""
    monkeypatch.setenv("RB_SWEEP", "1"); monkeypatch.setenv("RB_SWEEP_CLAIMS", claims); monkeypatch.setenv("RB_GROUP_T", T)     # (T: index ranges per sub-batch — the sweep needs at least one partition bit)
    (ls, lq, off), (rs, rq, roff) = make_reads(3000, 8000, 0.003, 1e-3, seed=size % 89)
    og, gg = graph_pair(size, size, 30_011, max_batch=40_000)
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    for rep in range(2):
        for seq, qual, o, rc in ((ls, lq, off, False), (rs, rq, roff, True)):
            og.add_reads(seq, qual, o, 3, rbo.REVCOMP if rc else 0)
            gg.addReads(seq, qual, o, 3, reverseComplement=rc)
            assert_same_state(og, gg, pairs=False)
    assert og.cbf_bytes().max() > 40


@pytest.mark.parametrize("k", [25, 35])
@pytest.mark.parametrize("sweep", ["0", "1"])
@pytest.mark.parametrize("pairs", [False, True])
def test_reads_without_repeats_stop_asking_the_prefilter_cache(monkeypatch, k, sweep, pairs):
    """where the cache drops nothing (reads off a genome far larger than the reads cover: every k-mer new) the window walk against it is
    skipped for 15 sub-batches at a time (csrc/rb_graph.hip add_range, RB_PF_SKIP=2: sub-batches of any size count) — the unfiltered emit
    path and the filtered one alternate inside one call, with and without the swept stage, with and without the paired k-mers' walker on its
    side stream; then the same reads again (every k-mer present now, still nothing for the cache to drop: counters of 1 and 2 move at every
    sighting)"""
    monkeypatch.setenv("RB_PF_SKIP", "2"); monkeypatch.setenv("RB_SWEEP", sweep)
    rng = np.random.default_rng(99 + k)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    genome = acgt[rng.integers(0, 4, 4_000_000, dtype=np.uint8)]
    lens = rng.integers(60, 400, 6000)
    starts = rng.integers(0, genome.size - 400, 6000)
    seq = np.concatenate([genome[a:a + L] for a, L in zip(starts, lens)])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    og, gg = graph_pair(20_000_003, 20_000_003, 2_000_003, k=k, pairs=pairs, max_batch=20_000)
    if pairs:
        og.set_read_pair_distance(40); gg.setReadPairedKmerDistance(40)
    tot = []
    for rep in range(2):
        o_st = og.add_reads(seq, None, off, 3, rbo.STORE_READ_PAIRS if pairs else 0)
        st = gg.addReads(seq, None, off, 3, storeReadPairedKmers=pairs)
        assert_same_state(og, gg, pairs=pairs)
        assert st.kmers == o_st.kmers
        tot.append((st.kmers, st.sorted_kmers))
    assert tot[0][1] == tot[0][0], tot


@pytest.mark.parametrize("tries", ["1", "4"])
def test_counting_filter_placement_trial_leaves_the_filter_clear(monkeypatch, capfd, tries):
    """a counting filter of 1 GB and more is placed by a trial (csrc/rb_graph.hip: alloc_best_placed: up to RB_ALLOC_TRIES
    allocations, each timed with random XOR pairs, the fastest kept): the XORs must cancel — the filter starts clear and ends
    as the oracle's, whichever allocation won"""
    monkeypatch.setenv("RB_ALLOC_TRIES", tries); monkeypatch.setenv("RB_ALLOC_DEBUG", "1")
    d = synth.generate_pairs(1500, G=20000, err=0.002, n_rate=1e-3, seed=5, uniform_expr=True)
    og, gg = graph_pair(3_000_017, (1 << 30) + 12_345, 400_009)
    err = capfd.readouterr().err
    assert err.count("cbf allocation") == int(tries), err
    assert gg.popcount(N.CBF) == 0
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS)
    gg.addReads(s, q, off, 3, storeReadPairedKmers=True)
    assert_same_state(og, gg)


@pytest.mark.parametrize("env", [{}, {"RB_RAGGED_LANES": "0"}, {"RB_FILTER_PIPE": "0"}, {"RB_NO_MPF": "1"}, {"RB_EMIT_RESUME": "0"}])
@pytest.mark.parametrize("k", [25, 31])
def test_trimmed_reads_take_a_read_per_lane(monkeypatch, env, k):
    """reads of DIFFERENT lengths that all fit ten packed words (trimmed 150-base reads: 4 or 5 words, and every edge: empty, shorter
    than k, exactly k, 32 / 33 / 64 / 65 / 256 / 257 / 320 bases) go through the read-per-lane prefilter with the lane's read found through the
    word offsets; many sub-batches, a hot cache, qualities that cut reads into segments, both files of a pair"""
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    d = synth.generate_pairs(2600, G=4000, L=320, err=0.002, n_rate=1e-3, seed=91, uniform_expr=True, frag_mean=450.0, frag_sd=30.0)
    rng = np.random.default_rng(6)
    edges = [0, 1, k - 1, k, k + 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 150, 255, 256, 257, 288, 289, 319, 320]
    og, gg = graph_pair(300_007, 400_009, 60_013, k=k, max_batch=15_000)
    og.set_read_pair_distance(90); gg.setReadPairedKmerDistance(90)
    for name, rc in (("left", False), ("right", True)):
        reads, quals = d[name], d[name[0] + "qual"]
        n = reads.shape[0]
        lens = np.where(rng.random(n) < 0.6, 150, rng.integers(100, 151, n))
        lens[rng.integers(0, n, 200)] = rng.choice(edges, 200)
        keep = np.arange(320)[None, :] < lens[:, None]
        s, q = reads[keep], quals[keep]
        off = np.zeros(n + 1, np.int64); np.cumsum(lens, out=off[1:])
        og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
        st = gg.addReads(s, q, off, 3, reverseComplement=rc, storeReadPairedKmers=True)
        assert_same_state(og, gg)
    assert og.cbf_bytes().max() > 30 and st.sorted_kmers < st.kmers


@pytest.mark.parametrize("k,stranded,trimmed", [(25, False, False), (25, True, False), (25, False, True), (31, False, True), (17, False, False),
                                                (35, False, True), (47, True, True), (63, False, True)])
def test_the_prefilter_walkers_agree_word_for_word(monkeypatch, capfd, k, stranded, trimmed):
    """round 5: the default prefilter walker fetches its cache buckets cooperatively (csrc/rb_batch.hip k_filter_reads_coop: the lanes that need a
    bucket post it, groups of four lanes fetch it, the helpers store it into the asker's LDS image a step later).  A broken fetch there cannot be
    seen in the filters — a lookup in the wrong image misses, the window is kept, the result stays exact — so with RB_FILTER_CHECK the library
    runs the walker of rounds 3-4 (and for k <= 31 the one that fetches in place) over the same words and fails the call on any difference in a
    word's count, keep mask or resume state: canonical / forward / reverse-complement hashing, uniform and ragged reads (crowded first windows,
    lanes that end early), the wide (32 <= k <= 63) walkers, many sub-batches on a hot cache"""
    monkeypatch.setenv("RB_FILTER_CHECK", "2")
    L = 320 if trimmed else 150
    d = synth.generate_pairs(2600, G=4000, L=L, err=0.002, n_rate=1e-3, seed=17 + k, uniform_expr=True, frag_mean=450.0, frag_sd=30.0)
    rng = np.random.default_rng(k)
    og, gg = graph_pair(300_007, 400_009, 60_013, k=k, stranded=stranded, max_batch=15_000)
    og.set_read_pair_distance(60); gg.setReadPairedKmerDistance(60)
    for name, rc in (("left", False), ("right", True)):
        reads, quals = d[name], d[name[0] + "qual"]
        n = reads.shape[0]
        if trimmed:
            lens = np.where(rng.random(n) < 0.6, 150, rng.integers(100, 151, n))
            lens[rng.integers(0, n, 200)] = rng.choice([0, 1, k - 1, k, k + 1, 32, 33, 64, 65, 150, 256, 257, 319, 320], 200)
        else:
            lens = np.full(n, L)
        keep = np.arange(L)[None, :] < lens[:, None]
        s, q = reads[keep], quals[keep]
        off = np.zeros(n + 1, np.int64); np.cumsum(lens, out=off[1:])
        og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
        st = gg.addReads(s, q, off, 3, reverseComplement=rc, storeReadPairedKmers=True)        # (raises if the walkers disagree)
        assert_same_state(og, gg)
    err = capfd.readouterr().err
    checked = [ln for ln in err.splitlines() if "filter check" in ln]
    assert len(checked) >= 4 and all(ln.endswith(" 0 differ") for ln in checked), err[-2000:]
    assert st.sorted_kmers < st.kmers


@pytest.mark.parametrize("k", [9, 12, 16, 17, 20, 26, 28, 29, 31])
def test_small_and_boundary_k_through_the_prefiltered_path(k):
    """k = 31 uses all 16 slots of the minimizer ring (k - m + 1 = 16), k <= 16 makes the minimizer the k-mer
    itself, k = 9 is below the minimizer cache's range... all must stay exact"""
    d = synth.generate_pairs(1500, G=1500, err=0.002, n_rate=1e-3, seed=100 + k, uniform_expr=True)
    og, gg = graph_pair(150_001, 200_003, 30_011, k=k, max_batch=20_000)
    og.set_read_pair_distance(100); gg.setReadPairedKmerDistance(100)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS)
    st = gg.addReads(s, q, off, 3, storeReadPairedKmers=True)
    assert_same_state(og, gg)
    assert st.sorted_kmers < st.kmers


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ragged_reads_fuzz(seed):
    """ragged read lengths (shorter than k up to several words), N runs, low-quality stretches, FASTA and FASTQ,
    both strands' files, many small sub-batches: the prefiltered path (minimizer ring across word boundaries,
    per-read word counts that differ) must stay exact"""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, 3000, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seqs, quals = [], []
    for _ in range(2500):
        L = int(rng.choice([5, 24, 25, 26, 31, 32, 33, 63, 64, 65, 97, 150, 151, 260]))
        a = int(rng.integers(0, genome.size - L)) if L < genome.size else 0
        s = acgt[genome[a:a + L]].copy()
        q = np.full(L, ord("I"), np.uint8)
        if rng.random() < 0.3 and L > 8:                       # an N run or a low-quality stretch somewhere
            p, n = int(rng.integers(0, L - 4)), int(rng.integers(1, 5))
            if rng.random() < 0.5: s[p:p + n] = ord("N")
            else: q[p:p + n] = ord("#")
        if rng.random() < 0.1: s = np.char.lower(s.view("S1")).view(np.uint8)   # lower case is legal (R/util/SeqUtils.java)
        seqs.append(s); quals.append(q)
    seq = np.concatenate(seqs); qual = np.concatenate(quals)
    off = np.concatenate([[0], np.cumsum([x.size for x in seqs])]).astype(np.int64)
    for use_qual in (True, False):
        og, gg = graph_pair(120_011, 170_003, 20_011, max_batch=9_000)
        og.set_read_pair_distance(40); gg.setReadPairedKmerDistance(40)
        for rc in (False, True):
            og.add_reads(seq, qual if use_qual else None, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
            gg.addReads(seq, qual if use_qual else None, off, 3, reverseComplement=rc, storeReadPairedKmers=True)
            assert_same_state(og, gg)


@pytest.mark.parametrize("k,stranded", [(32, False), (33, False), (35, True), (40, True), (47, False), (48, False), (62, False), (63, False), (64, False), (64, True)])
def test_wide_k_takes_the_prefiltered_path(k, stranded, monkeypatch):
    """32 <= k <= 64: the word-per-lane walkers with three words of input and 128 bits of history per lane
    (WordWalk<true>), hash-bucketed hot-k-mer cache, masked / sparse emit — ragged reads with N runs and
    low-quality stretches, both files, deep coverage so that the cache really drops occurrences, a
    count-if-present pass, many small sub-batches; and the same through the unfiltered generic kernels"""
    rng = np.random.default_rng(7000 + k)
    genome = rng.integers(0, 4, 2500, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seqs, quals = [], []
    for _ in range(3000):
        L = int(rng.choice([20, k - 1, k, k + 1, 64, 65, 96, 97, 128, 150, 151, 260, 301]))
        a = int(rng.integers(0, genome.size - L))
        s = acgt[genome[a:a + L]].copy()
        q = np.full(L, ord("I"), np.uint8)
        if rng.random() < 0.3 and L > 8:
            p, n = int(rng.integers(0, L - 4)), int(rng.integers(1, 5))
            if rng.random() < 0.5: s[p:p + n] = ord("N")
            else: q[p:p + n] = ord("#")
        seqs.append(s); quals.append(q)
    seq = np.concatenate(seqs); qual = np.concatenate(quals)
    off = np.concatenate([[0], np.cumsum([x.size for x in seqs])]).astype(np.int64)
    # (round 3: reads of up to 320 bases at k <= 63 take the read-per-lane prefilter with the minimizer-bucketed cache, the bucket of a
    # k-mer being that of the minimizer of its middle 21 (k odd) / 20 (k even) bases; RB_WIDE_MPF=0: the hash-bucketed cache and the one-word kernels as before)
    for wide, wmpf in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("RB_WIDE_PREFILTER", wide); monkeypatch.setenv("RB_WIDE_MPF", wmpf)
        og, gg = graph_pair(150_001, 200_003, 30_011, k=k, stranded=stranded, max_batch=15_000)
        og.set_read_pair_distance(30); gg.setReadPairedKmerDistance(30)
        for rc in (False, True):
            o_st = og.add_reads(seq, qual, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
            st = gg.addReads(seq, qual, off, 3, reverseComplement=rc, storeReadPairedKmers=True)
            assert_same_state(og, gg)
            assert st.kmers == o_st.kmers
            assert (st.sorted_kmers < st.kmers) == (wide == "1")     # the cache drops occurrences only on the prefiltered path
        og.add_reads(seq, qual, off, 3, rbo.COUNT_IF_PRESENT)
        gg.addReads(seq, qual, off, 3, incrementIfPresent=True)
        assert_same_state(og, gg)
        assert og.cbf_bytes().max() > 30


@pytest.mark.parametrize("words,k", [(1, 25), (2, 25), (5, 25), (5, 31), (8, 21), (9, 25)])
def test_uniform_word_count_fuzz(words, k):
    """every read has the same number of 32-base words but not the same length (the last word is partly filled),
    with N runs and low-quality stretches: batches of up to 8 words per read take the one-read-per-lane prefilter
    (k_filter_reads: words preloaded into registers, per-word counts / keep masks flushed at word boundaries,
    trailing words without a window start zero-filled), 9 words fall back to one word per lane — same state"""
    rng = np.random.default_rng(1000 * words + k)
    genome = rng.integers(0, 4, 4000, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seqs, quals = [], []
    for _ in range(3000):
        L = int(rng.integers(32 * (words - 1) + 1, 32 * words + 1))
        a = int(rng.integers(0, genome.size - L))
        s = acgt[genome[a:a + L]].copy()
        q = np.full(L, ord("I"), np.uint8)
        if rng.random() < 0.3 and L > 8:
            p, n = int(rng.integers(0, L - 4)), int(rng.integers(1, 5))
            if rng.random() < 0.5: s[p:p + n] = ord("N")
            else: q[p:p + n] = ord("#")
        seqs.append(s); quals.append(q)
    seq = np.concatenate(seqs); qual = np.concatenate(quals)
    off = np.concatenate([[0], np.cumsum([x.size for x in seqs])]).astype(np.int64)
    og, gg = graph_pair(150_001, 200_003, 30_011, k=k, max_batch=15_000)
    og.set_read_pair_distance(40); gg.setReadPairedKmerDistance(40)
    for rc in (False, True):
        og.add_reads(seq, qual, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
        st = gg.addReads(seq, qual, off, 3, reverseComplement=rc, storeReadPairedKmers=True)
        assert_same_state(og, gg)
    assert st.sorted_kmers < st.kmers or words == 1        # the prefilter did drop occurrences


@pytest.mark.parametrize("general", [False, True])
def test_read_pairs_at_scale(monkeypatch, general):
    """200 000 synthetic 150-base reads against the oracle: enough wavefronts that every SIMD of the device holds
    several of one kernel — the size at which the first paired-k-mer kernel set ~2 % of its bits at wrong
    positions (optimised build of k_pairs_insert; smaller inputs were exact).  Both paired-k-mer kernels: one read
    per lane (k_pairs_reads) and the general one (k_pairs_insert, compiled unoptimised)."""
    if general:
        monkeypatch.setenv("RB_PAIRS_GENERAL", "1")
    n = 200_000
    batch = ReadBatch.synthetic(n // 2, 2_000_000, 150, 300, 30, 0.002, 1e-3, 2.0, seed=99, device=0)
    seq, off = batch.download(0, n)
    og, gg = graph_pair(300_000_007, 300_000_007, 300_000_007, max_batch=0)
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    og.add_reads(seq, None, off, 3, rbo.STORE_READ_PAIRS)
    st = gg.addBatch(batch, storeReadPairedKmers=True, first=0, n=n)
    assert st.pairs > 1_000_000
    assert_same_state(og, gg)


@pytest.mark.parametrize("seen", ["default", "0", "4", "12"])
@pytest.mark.parametrize("stranded", [False, True])
def test_seen_pair_cache_never_changes_the_read_pair_filter(monkeypatch, seen, stranded):
    """k_pairs_reads behind its seen-pair cache (BitFilter::seen, round 5): deep coverage, so most pairs are sightings of pairs that are in
    the filter already and are skipped on the cache's word — the filter must equal the oracle's after every call, with the cache at its
    default size, switched off, with 16 buckets (every lookup evicts) and with 4096; after clearRpkbf (the cache has to forget), after an
    import of other bytes (likewise), after a second pass over the same reads (every pair known), and with N bases / low qualities that
    cut pairs out of the middle of reads."""
    if seen != "default":
        monkeypatch.setenv("RB_PAIR_SEEN", seen)
    (ls, lq, off), (rs, rq, roff) = make_reads(6000, 9000, 0.003, 2e-3, seed=77)        # ~80x coverage of a 9 kb transcriptome
    og, gg = graph_pair(400_009, 3_000_017, 250_007, stranded=stranded, max_batch=200_000)
    og.set_read_pair_distance(115); gg.setReadPairedKmerDistance(115)
    def both(seq, qual, o, rc):
        og.add_reads(seq, qual, o, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
        st = gg.addReads(seq, qual, o, 3, reverseComplement=rc, storeReadPairedKmers=True)
        assert (gg.exportFilter(N.RPKBF) == og.rpkbf_bytes()).all(), "rpkbf differs"
        return st
    st = both(ls, lq, off, False)
    assert st.pairs > 30_000
    both(rs, rq, roff, True)
    both(ls, lq, off, False)                                     # every pair is known now
    assert_same_state(og, gg)
    # the filter is emptied: a cache that still vouched for its pairs would leave the bits unset
    gg.clearRpkbf()
    assert gg.popcount(N.RPKBF) == 0
    og_b = rbo.Graph(400_009, 3_000_017, 250_007, 2, 2, 2, 25, stranded, True, 7)
    og_b.set_read_pair_distance(115)
    og_b.add_reads(rs, rq, roff, 3, rbo.STORE_READ_PAIRS | rbo.REVCOMP)
    gg.addPairs(ReadBatch.from_ascii(rs, rq, roff, 3), reverseComplement=True)
    assert (gg.exportFilter(N.RPKBF) == og_b.rpkbf_bytes()).all(), "rpkbf differs after clearRpkbf"
    # other bytes are imported: the same
    other = np.zeros_like(og_b.rpkbf_bytes()); other[::7] = 0x21
    gg.importFilter(N.RPKBF, other)
    gg.addPairs(ReadBatch.from_ascii(rs, rq, roff, 3), reverseComplement=True)
    assert (gg.exportFilter(N.RPKBF) == (og_b.rpkbf_bytes() | other)).all(), "rpkbf differs after an import"


@pytest.mark.parametrize("stranded", [False, True])
def test_max_cov_walks_match_oracle(stranded):
    """rb_graph_walk: batched greedy maximum-coverage walks (Kmer.getMaxCovSuccessor / getMaxCovPredecessor in the loop
    of GraphUtils.getMaxCoveragePath) against the step-by-step restatement over the oracle graph: appended bases, counts,
    length and stop reason, both directions, several coverage thresholds, with and without a target, seeds with N."""
    (ls, lq, off), _ = make_reads(2000, 6000, 0.004, 1e-3, seed=31)
    og, gg = graph_pair(400_009, 2_000_003, 10_007, stranded=stranded, pairs=False)
    og.add_reads(ls, lq, off, 3, 0); gg.addReads(ls, lq, off, 3)
    rng = np.random.default_rng(5)
    seeds, targets = [], []
    for _ in range(400):
        r = int(rng.integers(0, len(off) - 1)); p = int(rng.integers(0, 150 - 25 - 40))
        s = bytes(ls[off[r] + p: off[r] + p + 25])
        seeds.append(s)
        q = p + int(rng.integers(1, 40))
        targets.append(bytes(ls[off[r] + q: off[r] + q + 25]) if rng.random() < 0.7 else bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 25)))
    seeds[7] = seeds[7][:10] + b"N" + seeds[7][11:]
    seeds[8] = seeds[8].lower()
    reasons = set()
    for direction in (0, 1):
        for min_cov, bound, use_t in ((1.0, 60, True), (2.0, 35, False), (4.0, 10, True), (1.0, 1, True)):
            tg = targets if use_t else None
            if direction == 1 and tg is not None:      # walking left: aim at a k-mer to the left of the seed instead
                tg = [bytes(ls[off[0] + 5: off[0] + 30])] * len(seeds)
            bases, f, r, c, ln, reason = gg.walkMaxCov(seeds, direction, bound, min_cov, tg)
            for i, s in enumerate(seeds):
                eb, ec, er = rbo.walk_max_cov(og, s, direction, bound, min_cov, tg[i] if tg is not None else None, stranded=stranded)
                assert int(reason[i]) == er and int(ln[i]) == len(eb), (direction, min_cov, i, int(reason[i]), er, int(ln[i]), len(eb))
                assert bytes(bases[i, :ln[i]]) == eb and (c[i, :ln[i]] == np.array(ec, np.float32)).all()
                reasons.add(er)
    assert reasons >= {0, 1, 3, 4}


def test_max_cov_walk_meets_its_own_path():
    """a tandem repeat of period 30: the walk comes back to the first k-mer it appended after 30 steps and stops there
    (reason 2, GraphUtils.java:1613-1615 `leftPathKmers.contains(best)`), in both directions; the seed itself is not
    part of the visited set, so a period-1 step count of 30 — not 29 — is what the reference does"""
    rng = np.random.default_rng(2)
    unit = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 30))
    read = unit * 6
    og, gg = graph_pair(200_003, 400_009, 10_007, pairs=False)
    seq = np.frombuffer(read, np.uint8); off = np.array([0, len(read)], np.int64)
    og.add_reads(seq, None, off, 3, 0); gg.addReads(seq, None, off, 3)
    for direction in (0, 1):
        seed = read[40:65]
        bases, f, r, c, ln, reason = gg.walkMaxCov([seed], direction, 100, 1.0)
        eb, ec, er = rbo.walk_max_cov(og, seed, direction, 100, 1.0)
        assert (int(ln[0]), int(reason[0])) == (len(eb), er) == (30, 2)
        assert bytes(bases[0, :30]) == eb


def test_get_max_coverage_paths_match_oracle():
    """rnabloom.graphutils.getMaxCoveragePaths (two batched rb_graph_walk calls + the bookkeeping of
    GraphUtils.getMaxCoveragePath) against the statement-by-statement restatement over the oracle graph: pairs taken
    from the same read at various distances (connected from the left, only from the right, through an intersection of
    the two walks, not at all), random pairs, low bounds."""
    from rnabloom.graphutils import getMaxCoveragePaths, isLowComplexityShort
    (ls, lq, off), _ = make_reads(2500, 5000, 0.006, 0.0, seed=41)
    og, gg = graph_pair(300_007, 1_500_007, 10_007, pairs=False)
    og.add_reads(ls, lq, off, 3, 0); gg.addReads(ls, lq, off, 3)
    rng = np.random.default_rng(9)
    # a fork right after `left`: X is followed by Y five times and by Z once, left = the last k-mer of X, right inside Z.
    # The walk from the left takes the Y branch; the walk from the right comes back along Z and arrives at left itself.
    X, Y, Z = (bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 40)) for _ in range(3))
    fork = [X + Y] * 5 + [X + Z]
    fs = np.frombuffer(b"".join(fork), np.uint8); fo = np.arange(0, 81 * len(fork), 80, dtype=np.int64)
    og.add_reads(fs, None, fo, 3, 0); gg.addReads(fs, None, fo, 3)
    lefts, rights = [X[15:40]], [(X + Z)[50:75]]
    for _ in range(500):
        r = int(rng.integers(0, len(off) - 1)); p = int(rng.integers(0, 60))
        d = int(rng.integers(1, 60))
        lefts.append(bytes(ls[off[r] + p: off[r] + p + 25]))
        if rng.random() < 0.85:
            rights.append(bytes(ls[off[r] + p + d: off[r] + p + d + 25]))
        else:
            r2 = int(rng.integers(0, len(off) - 1))
            rights.append(bytes(ls[off[r2] + 3: off[r2] + 28]))
    kinds = set()
    trace = []
    for bound, min_cov in ((70, 1.0), (20, 2.0), (8, 1.0)):
        got = getMaxCoveragePaths(gg, lefts, rights, bound, min_cov)
        for i in range(len(lefts)):
            exp = rbo.get_max_coverage_path(og, lefts[i], rights[i], bound, min_cov, low_complexity=isLowComplexityShort, trace=trace)
            assert got[i] == exp, (bound, min_cov, i)
            kinds.add(None if exp is None else (len(exp) > 0))
    assert kinds == {None, True, False}
    assert set(trace) == {"from the left", "from the right", "walks meet"}, set(trace)
    assert isLowComplexityShort(b"A" * 25) and isLowComplexityShort(b"AC" * 12 + b"A") and not isLowComplexityShort(lefts[0])


@pytest.mark.parametrize("stranded", [False, True])
def test_greedy_extend_with_lookahead_matches_oracle(stranded):
    """rb_graph_greedy_extend (GraphUtils.greedyExtendRight / Left: per step the candidate whose depth-first lookahead
    paths have the best minimum coverage, ties to the larger count) against the statement-by-statement restatement over
    the oracle graph — a graph with plenty of branches (high error rate, small filters => false-positive neighbours),
    lookahead 0..6, both directions."""
    (ls, lq, off), _ = make_reads(1500, 3000, 0.02, 0.0, seed=57)
    og, gg = graph_pair(150_001, 600_011, 10_007, stranded=stranded, pairs=False)
    og.add_reads(ls, lq, off, 3, 0); gg.addReads(ls, lq, off, 3)
    rng = np.random.default_rng(15)
    seeds = []
    for _ in range(160):
        r = int(rng.integers(0, len(off) - 1)); p = int(rng.integers(0, 120))
        seeds.append(bytes(ls[off[r] + p: off[r] + p + 25]))
    branched = 0
    for direction in (0, 1):
        for lookahead, bound in ((3, 40), (5, 25), (0, 10), (1, 10), (6, 12)):
            bases, c, ln, reason = gg.greedyExtend(seeds, direction, lookahead, bound)
            for i, s in enumerate(seeds):
                eb, ec = rbo.greedy_extend(og, s, direction, lookahead, bound)
                assert int(ln[i]) == len(eb) and bytes(bases[i, :ln[i]]) == eb, (direction, lookahead, i)
                assert (c[i, :ln[i]] == np.array(ec, np.float32)).all()
                assert int(reason[i]) == (3 if len(eb) == bound else 0)
            if lookahead == 3:
                plain = gg.walkMaxCov(seeds, direction, bound, 1.0, hashes=False)[0]
                branched += int((plain != bases).any(axis=1).sum())
    assert branched > 0          # the lookahead changed at least one decision of the plain maximum-count walk


@pytest.mark.parametrize("stranded", [False, True])
def test_pair_only_and_fragment_workers_match_oracle(stranded):
    """rb_graph_add_pairs (PairedKmersToGraphWorker: paired k-mers only, with and without existingKmersOnly) and
    rb_graph_add_fragments (FragmentsToGraphWorker: dbgbf-only k-mers + read pairs + fragment pairs) against the same
    loops written over the oracle's per-hash entry points, hash values from the oracle's iterators."""
    k, read_d, frag_d, mode = 25, 60, 150, (0 if stranded else 1)
    (ls, lq, off), _ = make_reads(700, 4000, 0.004, 1e-3, seed=77)
    og, gg = graph_pair(300_007, 900_001, 200_003, stranded=stranded)
    og.set_read_pair_distance(read_d); gg.setReadPairedKmerDistance(read_d)
    og.init_fragment_pairs(250_007, 2, frag_d); gg.initializePairKmersBloomFilter(250_007, 2); gg.setFragPairedKmerDistance(frag_d)
    # half of the reads go in as ordinary reads (so that some k-mers exist), then pairs of ALL reads, existing k-mers only
    half = 350
    og.add_reads(ls[:off[half]], lq[:off[half]], off[:half + 1], 3, 0); gg.addReads(ls[:off[half]], lq[:off[half]], off[:half + 1], 3)
    batch = ReadBatch.from_ascii(ls, None, off, 3)
    reads = [bytes(ls[off[i]:off[i + 1]]) for i in range(len(off) - 1)]

    def oracle_pairs(seqs, d, add, existing):
        n = 0
        for s in seqs:
            for a, b in rbo.segments(s, None, k, 3):
                if b - a < k + d:
                    continue
                p, l, r = rbo.hash_pairs_region(s, k, 2, d, mode, int(a), int(b))
                for i in range(p.shape[0]):
                    if existing and not (og.contains(l[i]) and og.contains(r[i])):
                        continue
                    add(p[i]); n += 1
        return n

    n_exist = oracle_pairs(reads, read_d, og.add_read_pair, True)
    st = gg.addPairs(batch, N.RPKBF, existingKmersOnly=True)
    assert st.pairs == n_exist > 1000
    assert_same_state(og, gg)
    n_all = oracle_pairs(reads, read_d, og.add_read_pair, False)
    st = gg.addPairs(batch, N.RPKBF)
    assert st.pairs == n_all > n_exist
    assert_same_state(og, gg)
    # fragments: ACGT-only sequences of 60..400 bases
    rng = np.random.default_rng(4)
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 20_000)]
    frags = []
    for _ in range(500):
        L = int(rng.integers(60, 400)); a = int(rng.integers(0, genome.size - L))
        frags.append(genome[a:a + L].tobytes())
    fseq = np.frombuffer(b"".join(frags), np.uint8); foff = np.concatenate([[0], np.cumsum([len(f) for f in frags])]).astype(np.int64)
    fb = ReadBatch.from_ascii(fseq, None, foff, 3)
    for load in (False, True):
        n_k = n_p = 0
        for s in frags:
            hv, _ = rbo.hash_region(s, k, 2, mode)
            for i in range(hv.shape[0]):
                og.add_dbg_only(hv[i])
            n_k += hv.shape[0]
            if load and len(s) >= k + read_d:
                p, _, _ = rbo.hash_pairs_region(s, k, 2, read_d, mode)
                for i in range(p.shape[0]):
                    og.add_read_pair(p[i])
                n_p += p.shape[0]
                if len(s) >= k + frag_d:
                    p, _, _ = rbo.hash_pairs_region(s, k, 2, frag_d, mode)
                    for i in range(p.shape[0]):
                        og.add_fragment_pair(p[i])
                    n_p += p.shape[0]
        st = gg.addFragments(fb, loadPairedKmers=load)
        assert (st.kmers, st.pairs) == (n_k, n_p)
        assert_same_state(og, gg)
        assert (gg.exportFilter(N.FPKBF) == og.fpkbf_bytes()).all()
    assert gg.popcount(N.FPKBF) > 1000


@pytest.mark.parametrize("which,name", [(N.DBGBF, "dbgbf"), (N.RPKBF, "rpkbf")])
def test_lookup_then_add_in_array_order(which, name):
    """rb_filter_lookup_then_add: BloomFilter.lookupThenAdd over an array must answer as the sequential loop does — an
    element is 'found' when earlier elements of the same array already set all of its bits.  A tiny filter (most
    answers come from collisions inside the array), repeats, three rounds on the same filter."""
    import ctypes as C
    rng = np.random.default_rng(8)
    og, gg = graph_pair(20_011, 30_011, 5_003)
    ob = getattr(og.L, "rbo_graph_" + name)(og.g)
    pool = rng.integers(0, 1 << 63, 3000, dtype=np.int64).astype(np.uint64)
    seen_true = seen_false = 0
    for rnd in range(3):
        hs = pool[rng.integers(0, len(pool), 6000)]
        exp = np.empty(hs.size, bool)
        for i, h in enumerate(hs):
            hv = rbo.ntm64(int(h), 25, 2)
            exp[i] = bool(og.L.rbo_bloom_lookup_then_add(C.c_void_p(ob), hv.ctypes.data_as(C.c_void_p)))
        got = gg.lookupThenAdd(which, hs)
        assert (got == exp).all(), (rnd, int((got != exp).sum()))
        seen_true += int(exp.sum()); seen_false += int((~exp).sum())
        assert (gg.exportFilter(which) == getattr(og, name + "_bytes")()).all()
    assert seen_true > 1000 and seen_false > 1000


def test_standalone_filter_classes():
    """rnabloom.bloom.BloomFilter / CountingBloomFilter (the reference's stand-alone filter classes) against the oracle's
    stand-alone rbo_bloom and against an oracle graph's counting filter: add, lookup, lookupThenAdd, pop count, FPR,
    bytes, counts after many increments of a small key set (probabilistic range), getBloomFilter(minCount)."""
    import ctypes as C
    from rnabloom.bloom import BloomFilter, CountingBloomFilter, PairedKeysBloomFilter
    rng = np.random.default_rng(12)
    L = rbo.lib()
    size, h, k = 50_021, 3, 25
    ob = L.rbo_bloom_new(size, h)
    bf = BloomFilter(size, h, k)
    keys = rng.integers(0, 1 << 63, 4000, dtype=np.int64).astype(np.uint64)
    hv = lambda x: rbo.ntm64(int(x), k, h)
    for x in keys[:2500]:
        L.rbo_bloom_add(C.c_void_p(ob), hv(x).ctypes.data_as(C.c_void_p))
    bf.add(keys[:2500])
    exp = np.array([bool(L.rbo_bloom_lookup(C.c_void_p(ob), hv(x).ctypes.data_as(C.c_void_p))) for x in keys])
    assert (bf.lookup(keys) == exp).all() and exp[:2500].all() and not exp.all()
    order = keys[rng.integers(2000, 4000, 3000)]
    exp = np.array([bool(L.rbo_bloom_lookup_then_add(C.c_void_p(ob), hv(x).ctypes.data_as(C.c_void_p))) for x in order])
    assert (bf.lookupThenAdd(order) == exp).all()
    assert bf.getPopCount() == -1                                    # BloomFilter.java:48: unknown until getFPR() counts
    assert np.float32(bf.getFPR()) == np.float32(L.rbo_bloom_fpr(C.c_void_p(ob))) and bf.getPopCount() == L.rbo_bloom_popcount(C.c_void_p(ob))
    n = C.c_int64(); p = L.rbo_bloom_bytes(C.c_void_p(ob), C.byref(n))
    assert (bf.toBytes() == np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,))).all()
    assert BloomFilter.getExpectedSize(1_000_000, 0.01, 2) == L.rbo_expected_size(1_000_000, 0.01, 2) == PairedKeysBloomFilter.getExpectedSize(1_000_000, 0.01, 2)
    L.rbo_bloom_free(C.c_void_p(ob)); bf.destroy()
    # counting filter: the oracle graph's cbf under addCountOnly is the same object (same generator: seed + op ordinal)
    csize = 30_011
    og = rbo.Graph(64, csize, 0, 1, 2, 1, k, True, False, 5)
    cbf = CountingBloomFilter(csize, 2, k, rngSeed=5)
    small = keys[:300]
    for rnd in range(4):
        seq = small[rng.integers(0, small.size, 5000)]
        for x in seq:
            og.add_count_only(rbo.ntm64(int(x), k, 2))
        cbf.increment(seq)
        assert (cbf.toBytes() == og.cbf_bytes()).all()
    assert og.cbf_bytes().max() > 24
    exp = np.array([L.rbo_cbf_get_count(C.c_void_p(L.rbo_graph_cbf(og.g)), rbo.ntm64(int(x), k, 2).ctypes.data_as(C.c_void_p)) for x in keys[:600]], np.float32)
    assert (cbf.getCount(keys[:600]) == exp).all()
    raw = og.cbf_bytes()
    to_float = lambda b: float(b) if b <= 7 else float(((b & 7) | 8) * 2.0 ** ((b >> 3) - 1))     # MiniFloat.toFloat, R/util/MiniFloat.java:40-45
    pops = []
    for min_count in (20, 120):
        want = np.packbits(np.array([to_float(int(b)) >= min_count for b in raw]), bitorder="little")
        bmin = cbf.getBloomFilter(min_count)
        assert (bmin.toBytes() == want).all()
        bmin.getFPR(); pops.append(bmin.getPopCount()); bmin.destroy()
    cbf.getFPR()
    assert cbf.getPopCount() >= pops[0] > pops[1] >= 0
    cbf.destroy()


@pytest.mark.parametrize("stranded", [False, True])
def test_kmer_list_pair_helpers(stranded):
    """graphutils.containsAllPairedKmers / lookupAndAddAllPairedKmers (BloomFilterDeBruijnGraph.java:496-526 over lists of
    k-mers) for batches of sequences, against the same loops over the oracle's fragment pair filter; pair hash values
    (Kmer / CanonicalKmer.getKmerPairHashValue) against the oracle's paired iterator."""
    import ctypes as C
    from rnabloom.graphutils import containsAllPairedKmers, lookupAndAddAllPairedKmers, kmerPairHashValues
    k, d, mode = 25, 30, (0 if stranded else 1)
    og, gg = graph_pair(100_003, 200_003, 50_021, stranded=stranded)
    og.init_fragment_pairs(40_009, 2, d); gg.initializePairKmersBloomFilter(40_009, 2); gg.setFragPairedKmerDistance(d)
    rng = np.random.default_rng(6)
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 3000)]
    seqs = []
    for _ in range(300):
        L = int(rng.integers(30, 140)); a = int(rng.integers(0, genome.size - L))
        seqs.append(genome[a:a + L].tobytes())
    ko, f, r, _ = gg.getKmers(seqs[:5])
    for i in range(5):
        if len(seqs[i]) >= k + d:
            p, _, _ = rbo.hash_pairs_region(seqs[i], k, 1, d, mode)
            assert (kmerPairHashValues(f[ko[i]:ko[i + 1]], r[ko[i]:ko[i + 1]], d, stranded) == p[:, 0]).all()
    fp = og.L.rbo_graph_fpkbf(og.g)
    for rnd in range(2):
        exp_c, exp_l = [], []
        for s in seqs:                                     # containsAll on the state before this round's adds
            p = rbo.hash_pairs_region(s, k, 2, d, mode)[0] if len(s) >= k + d else np.zeros((0, 2), np.uint64)
            exp_c.append(bool(p.shape[0] > 0 and all(og.lookup_fragment_pair(row) for row in p)))
        assert containsAllPairedKmers(gg, seqs, d) == exp_c
        for s in seqs:
            p = rbo.hash_pairs_region(s, k, 2, d, mode)[0] if len(s) >= k + d else np.zeros((0, 2), np.uint64)
            found = True
            for row in p:
                found &= bool(og.L.rbo_bloom_lookup_then_add(C.c_void_p(fp), np.ascontiguousarray(row).ctypes.data_as(C.c_void_p)))
            exp_l.append(found)
        assert lookupAndAddAllPairedKmers(gg, seqs, d) == exp_l
        assert (gg.exportFilter(N.FPKBF) == og.fpkbf_bytes()).all()
        assert rnd == 0 or all(exp_l)
    assert any(exp_c) and not all(exp_c)


def test_greedy_extend_with_bloom_filter_gate():
    """the `bf` variants (greedyExtendRight(graph, source, lookahead, bound, bf): a neighbour must pass bf.lookup before its
    count is read, Kmer.java:257-299): the gate is a stand-alone rnabloom.bloom.BloomFilter holding the k-mers of a third
    of the reads; oracle: the same restatement with the oracle's stand-alone filter as the gate."""
    import ctypes as C
    from rnabloom.bloom import BloomFilter
    (ls, lq, off), _ = make_reads(1200, 3000, 0.01, 0.0, seed=67)
    og, gg = graph_pair(150_001, 600_011, 10_007, pairs=False)
    og.add_reads(ls, lq, off, 3, 0); gg.addReads(ls, lq, off, 3)
    third = 12                                            # the first 12 reads feed the gate: about a third of the genome
    gb = ReadBatch.from_ascii(ls[:off[third]], None, off[:third + 1], 3)
    h0 = gb.nthash(25, 1)
    L = rbo.lib()
    ob = L.rbo_bloom_new(400_009, 2)
    bf = BloomFilter(400_009, 2, 25)
    bf.add(h0)
    for x in np.unique(h0):
        L.rbo_bloom_add(C.c_void_p(ob), rbo.ntm64(int(x), 25, 2).ctypes.data_as(C.c_void_p))
    n = C.c_int64(); p = L.rbo_bloom_bytes(C.c_void_p(ob), C.byref(n))
    assert (bf.toBytes() == np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value,))).all()
    gate = lambda h: bool(L.rbo_bloom_lookup(C.c_void_p(ob), rbo.ntm64(int(h), 25, 2).ctypes.data_as(C.c_void_p)))
    rng = np.random.default_rng(3)
    seeds = [bytes(ls[off[r] + p: off[r] + p + 25]) for r, p in zip(rng.integers(0, 1200, 120), rng.integers(0, 120, 120))]
    shorter = 0
    for direction in (0, 1):
        bases, c, ln, reason = gg.greedyExtend(seeds, direction, 4, 30, bf=bf)
        free = gg.greedyExtend(seeds, direction, 4, 30)[2]
        for i, s in enumerate(seeds):
            eb, ec = rbo.greedy_extend(og, s, direction, 4, 30, gate=gate)
            assert int(ln[i]) == len(eb) and bytes(bases[i, :ln[i]]) == eb and (c[i, :ln[i]] == np.array(ec, np.float32)).all()
        shorter += int((ln < free).sum())
    assert shorter > 10                                    # the gate did cut walks short
    L.rbo_bloom_free(C.c_void_p(ob)); bf.destroy()


def test_get_kmers_min_coverage_variant():
    """graphutils.getKmersMinCoverage against the object-identity-faithful restatement of
    HashFunction.getKmers(seq, numHash, graph, minCoverage) over the oracle graph: sequences with several runs above the
    threshold (the reference keeps the first run only if no later closed run exists, and never looks at a run that is
    still open at the end)."""
    from rnabloom.graphutils import getKmersMinCoverage
    (ls, lq, off), _ = make_reads(1500, 4000, 0.01, 0.0, seed=91)
    og, gg = graph_pair(300_007, 1_200_007, 10_007, pairs=False)
    og.add_reads(ls, lq, off, 3, 0); gg.addReads(ls, lq, off, 3)
    rng = np.random.default_rng(2)
    genome_like = [bytes(ls[off[i]:off[i + 1]]) for i in range(200)]
    seqs = genome_like + [b"ACGT" * 10, b"", genome_like[0][:30] + b"N" + genome_like[1][:60]]
    kinds = set()
    for min_cov in (2.0, 6.0, 20.0):
        got = getKmersMinCoverage(gg, seqs, min_cov)
        for i, s in enumerate(seqs):
            exp = rbo.get_kmers_min_coverage(og, s, min_cov)
            st, n, cnt = got[i]
            assert n == len(exp) and (n == 0 or (st == exp[0][0] and (cnt == np.array([x[1] for x in exp], np.float32)).all())), (min_cov, i)
            kinds.add("empty" if n == 0 else ("from the start" if st == 0 else "later run"))
    assert kinds == {"empty", "from the start", "later run"}


def test_reference_facade_and_graph_files(tmp_path):
    """the per-element conveniences of the reference's facade (getKmer, contains(String), isValidSeq,
    getLeft/RightVariants, getSuccessors/Predecessors, addReadPairedKmers / addFragmentPairKmers) against the oracle
    graph, and the reference's file set (graph desc + .dbgbf/.cbf/.rpkbf/.fpkbf with their .desc): a saved graph comes
    back byte for byte and answers the same"""
    d = synth.generate_pairs(800, G=3000, err=0.002, n_rate=0.0, seed=4242)
    og, gg = graph_pair(400_009, 500_009, 80_021, max_batch=30_000)
    og.set_read_pair_distance(60); gg.setReadPairedKmerDistance(60)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS); gg.addReads(s, q, off, 3, storeReadPairedKmers=True)
    og.init_fragment_pairs(70_001, 2, 40); gg.initializePairKmersBloomFilter(70_001, 2); gg.setFragPairedKmerDistance(40)
    from rnabloom.graphutils import kmerPairHashValues
    reads = [bytes(s[off[i]:off[i + 1]]) for i in range(40)]
    rng = np.random.default_rng(5)
    hv = lambda x: rbo.ntm64(int(x), 25, 2)
    o_contains = lambda hs: np.array([og.contains(hv(x)) for x in hs], bool)
    # pair lists of whole sequences (read pairs into rpkbf, fragment pairs into fpkbf)
    for sq in reads[:10]:
        f, r, _ = og.get_kmers(sq)
        gg.addReadPairedKmers(f, r); gg.addFragmentPairKmers(f, r)
        for x in kmerPairHashValues(f, r, 60, False): og.add_read_pair(hv(x))
        for x in kmerPairHashValues(f, r, 40, False): og.add_fragment_pair(hv(x))
    assert (gg.exportFilter(N.RPKBF) == og.rpkbf_bytes()).all() and (gg.exportFilter(N.FPKBF) == og.fpkbf_bytes()).all()
    # k-mer strings: present ones, mutated ones
    kms = []
    for sq in reads[10:30]:
        p = int(rng.integers(0, len(sq) - 25)); km = bytearray(sq[p:p + 25])
        if b"N" in km: continue
        kms.append(bytes(km))
        km[int(rng.integers(0, 25))] = b"ACGT"[int(rng.integers(0, 4))]; kms.append(bytes(km))
    exp = []
    for km in kms:
        f, r, c = og.get_kmers(km)
        exp.append((int(f[0]), int(r[0]), float(c[0])))
        assert gg.getKmer(km) == exp[-1]
    canon = np.array([min(np.int64(np.uint64(e[0]).view(np.int64)), np.int64(np.uint64(e[1]).view(np.int64))) for e in exp]).view(np.uint64)
    assert (gg.containsSeq(kms) == o_contains(canon)).all()
    for km in kms[:12]:
        for pos, fn in ((0, gg.getLeftVariants), (24, gg.getRightVariants)):
            want = []
            for c in {65: b"CGT", 67: b"AGT", 71: b"ACT", 84: b"ACG"}[km[pos]]:
                v = km[:pos] + bytes([c]) + km[pos + 1:]
                f, r, _ = og.get_kmers(v)
                if og.contains(hv(np.array([min(f[0].view(np.int64), r[0].view(np.int64))]).view(np.uint64)[0])): want.append(v.decode())
            assert fn(km) == want
    seqs = reads[30:40] + [reads[31][:24], reads[32][:10] + b"ACGTACGTACGTACGTACGTACGTACGT" + reads[32][10:]]
    want = []
    for sq in seqs:
        f, r, _ = og.get_kmers(sq)
        hh = np.where(r.view(np.int64) < f.view(np.int64), r, f)
        want.append(bool(o_contains(hh).all()) if hh.size else True)
    assert gg.isValidSeq(seqs) == want and want[0] and not want[-1]
    f, r, _ = og.get_kmers(reads[12]); ch = np.frombuffer(reads[12][:f.size], np.uint8)
    f4, r4, c4, keep = gg.getSuccessors(f[:8], r[:8], ch[:8], 2.0)
    p4 = gg.getPredecessors(f[:8], r[:8], np.frombuffer(reads[12][24:32], np.uint8), 2.0)
    for i in range(8):
        of4, or4, oc4 = og.neighbors(f[i], r[i], int(ch[i]), 0)
        assert (f4[i] == of4).all() and (r4[i] == or4).all() and (c4[i] == oc4).all() and (keep[i] == (oc4 >= 2.0)).all()
        of4, or4, oc4 = og.neighbors(f[i], r[i], int(reads[12][24 + i]), 1)
        assert (p4[0][i] == of4).all() and (p4[1][i] == or4).all() and (p4[2][i] == oc4).all()
    # files
    path = str(tmp_path / "graph")
    gg.save(path); gg.savePkbf(path)
    desc = open(path).read().splitlines()
    assert desc == ["dbgbfCbfMaxNumHash:2", "stranded:false", "k:25", "readPairedKmersDistance:60", "fragmentPairedKmersDistance:40"]
    assert open(path + ".dbgbf.desc").read().startswith("size:400009\nnumhash:2\nfpr:")
    assert np.array_equal(np.fromfile(path + ".cbf", np.uint8), og.cbf_bytes())
    g2 = BloomFilterDeBruijnGraph.fromFile(path)
    for w in (N.DBGBF, N.CBF, N.RPKBF, N.FPKBF):
        assert np.array_equal(g2.exportFilter(w), gg.exportFilter(w))
    assert g2.k == 25 and not g2.stranded and g2.getReadPairedKmerDistance() == 60 and g2.getFragPairedKmerDistance() == 40
    assert (g2.containsSeq(kms) == gg.containsSeq(kms)).all() and g2.getKmer(kms[0]) == exp[0]
    g3 = BloomFilterDeBruijnGraph.fromFile(path, loadDbgBits=False)
    assert g3.popcount(N.DBGBF) == 0 and np.array_equal(g3.exportFilter(N.CBF), og.cbf_bytes())
    g3.updateFragmentKmerDistance(path); assert g3.getFragPairedKmerDistance() == 40
    # hash iterators over sequences
    h0, rd, ps = gg.getHashIterator(reads[:5])
    at = 0
    for i, sq in enumerate(reads[:5]):
        f, r, _ = og.get_kmers(sq)
        hh = np.where(r.view(np.int64) < f.view(np.int64), r, f)
        assert (h0[at:at + hh.size] == hh).all() and (rd[at:at + hh.size] == i).all() and (ps[at:at + hh.size] == np.arange(hh.size)).all()
        at += hh.size
    assert at == h0.size and (gg.getReverseComplementHashIterator(reads[:5])[0] == h0).all()     # canonical graph: same values
    # the graph's filters as objects
    assert gg.getDbgbfFPR() >= 0                                     # (getPopCount() is the count the last getFPR() remembered)
    assert gg.getDbgbf().getPopCount() == og.popcounts()[0] and gg.getCbf().getNumHash() == 2 and gg.getRpkbf().getSize() == 80_021
    assert (gg.getDbgbf().lookup(canon) == o_contains(canon)).all() and np.array_equal(gg.getFpkbf().toBytes(), og.fpkbf_bytes())
    assert (gg.getCbf().getCount(canon) == gg.getCbfCount(canon)).all() and gg.getFpkbfFPR() == gg.getPkbfFPR()
