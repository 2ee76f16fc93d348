"""Sharded (multi-GPU) engine on ONE GPU: G virtual ranks in one process (LoopbackCluster) must
reproduce the sequential oracle bit for bit, for every G.  Every rank walks the same read batch and
keeps the k-mers it owns, so the insertion order is the read order — the oracle consumes the same
reads in the same order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import rbo
from rnabloom import _native as N
from rnabloom import synth
from rnabloom.graph import ReadBatch
from rnabloom.sharded import LoopbackCluster


def check_filters(cl, og, pairs=True):
    assert (cl.exportFilter(N.DBGBF) == og.dbgbf_bytes()).all(), "dbgbf differs"
    if pairs:
        assert (cl.exportFilter(N.RPKBF) == og.rpkbf_bytes()).all(), "rpkbf differs"
    cg, co = cl.exportFilter(N.CBF), og.cbf_bytes()
    bad = np.nonzero(cg != co)[0]
    assert bad.size == 0, "cbf differs at %d bytes: %s gpu %s oracle %s" % (bad.size, bad[:6], cg[bad[:6]], co[bad[:6]])


@pytest.mark.parametrize("mode", ["replicated", "split"])
@pytest.mark.parametrize("G", [1, 2, 4, 8])
@pytest.mark.parametrize("sizes", [(400_003, 3_000_017, 90_001), (70_001, 250_007, 9_001)])
def test_loopback_matches_oracle(G, sizes, mode):
    d = synth.generate_pairs(2400, G=25000, err=0.003, n_rate=1e-3, seed=17 + G)
    og = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 9)
    cl = LoopbackCluster(G, *sizes, 2, 2, 2, 25, False, True, rngSeed=9, mode=mode)
    og.set_read_pair_distance(115); cl.setReadPairedKmerDistance(115)
    for name, rc in (("left", False), ("right", True)):
        s, off = synth.flat(d[name]); q, _ = synth.flat(d[name[0] + "qual"])
        og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
        cl.addBatch(ReadBatch.from_ascii(s, q, off, 3), 150, reverseComplement=rc, storeReadPairedKmers=True, reads_per_substep=577)
        check_filters(cl, og)
    pc = og.popcounts()
    assert (cl.popcount(N.DBGBF), cl.popcount(N.CBF), cl.popcount(N.RPKBF)) == pc
    assert sum(r.stats["conflict_ops"] for r in cl.ranks) > 0
    assert sum(r.stats["reads"] for r in cl.ranks) == 4800
    cl.destroy()


@pytest.mark.parametrize("mode", ["replicated", "split"])
@pytest.mark.parametrize("k", [25, 33])
def test_loopback_high_multiplicity(k, mode):
    """counts well into the probabilistic MiniFloat range, many sub-batches (the prefilter cache is hot);
    k = 33 takes the generic window-hash path (ownership + prefilter applied to the records)"""
    d = synth.generate_pairs(5000, G=3000, err=0.001, n_rate=1e-3, seed=3, uniform_expr=True)
    sizes = (100_003, 150_001, 20_011)
    og = rbo.Graph(*sizes, 2, 2, 2, k, False, True, 1)
    cl = LoopbackCluster(4, *sizes, 2, 2, 2, k, False, True, rngSeed=1, mode=mode)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    og.add_reads(s, q, off, 3, 0)
    cl.addBatch(ReadBatch.from_ascii(s, q, off, 3), 150, reads_per_substep=500)
    check_filters(cl, og, pairs=False)
    assert og.cbf_bytes().max() > 24
    kept = sum(r.stats["sorted_kmers"] for r in cl.ranks)
    assert kept < sum(r.stats["kmers"] for r in cl.ranks), "the prefilter dropped nothing"
    cl.destroy()


def test_loopback_stranded_count_if_present():
    """stranded hashing (no canonical minimum) + the addCountIfPresent pass over a pre-built dbgbf"""
    d = synth.generate_pairs(1500, G=6000, err=0.002, n_rate=1e-3, seed=5)
    sizes = (200_003, 300_007, 20_011)
    og = rbo.Graph(*sizes, 2, 2, 2, 25, True, False, 4)
    cl = LoopbackCluster(2, *sizes, 2, 2, 2, 25, True, False, rngSeed=4, mode="split")
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    b = ReadBatch.from_ascii(s, q, off, 3)
    og.add_reads(s, q, off, 3, 0)
    cl.addBatch(b, 150, reads_per_substep=400)
    check_filters(cl, og, pairs=False)
    og.add_reads(s, q, off, 3, rbo.COUNT_IF_PRESENT)
    from rnabloom.sharded import run_loopback, plan
    pos_bits, _ = plan(150, 25, 2)
    run_loopback([r.add_range(b, 0, b.n_reads, N.ADD_COUNT_IF_PRESENT, 400, pos_bits) for r in cl.ranks])
    check_filters(cl, og, pairs=False)
    cl.destroy()


def test_loopback_queries_match_single_gpu():
    """contains / getCount / counting-filter count / read-pair lookup on the sharded filters, asked from
    different ranks, against the single-GPU graph built from the same reads (itself oracle-exact)"""
    from rnabloom.graph import BloomFilterDeBruijnGraph
    d = synth.generate_pairs(1500, G=5000, err=0.002, n_rate=1e-3, seed=13)
    sizes = (300_007, 400_009, 50_021)
    gg = BloomFilterDeBruijnGraph(*sizes, 2, 2, 2, 25, False, True, rngSeed=2)
    cl = LoopbackCluster(4, *sizes, 2, 2, 2, 25, False, True, rngSeed=2)
    gg.setReadPairedKmerDistance(115); cl.setReadPairedKmerDistance(115)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    b = ReadBatch.from_ascii(s, q, off, 3)
    gg.addBatch(b, storeReadPairedKmers=True)
    cl.addBatch(b, 150, storeReadPairedKmers=True, reads_per_substep=300)
    reads = [bytes(s[off[i]:off[i + 1]]) for i in range(0, 40)]
    _, f, r, _ = gg.getKmers(reads)
    h_present = np.where(f.view(np.int64) < r.view(np.int64), f, r)          # canonical hashVals[0]
    rng = np.random.default_rng(5)
    h_random = rng.integers(0, 2 ** 63, 3000, dtype=np.uint64)
    allh = np.concatenate([h_present, h_random])
    parts = [allh[0::3], allh[1::3], np.zeros(0, np.uint64), allh[2::3]]     # rank 2 asks nothing
    for got, ref in ((cl.contains(parts), gg.contains), (cl.getCount(parts), gg.getCount), (cl.getCbfCount(parts), gg.getCbfCount),
                     (cl.lookupReadKmerPair(parts), gg.lookupReadKmerPair)):
        for p, g_ in zip(parts, got):
            assert g_.shape[0] == p.shape[0]
            if p.size:
                assert (g_ == ref(p)).all()
    assert cl.contains(parts)[0].any() and cl.getCount(parts)[0].max() > 1
    cl.destroy(); gg.destroy() if hasattr(gg, "destroy") else None


@pytest.mark.parametrize("env", [{"RB_SHARD_ORDER_ALL": "1"}, {"RB_SMALL_COMPONENT_OPS": "1"}, {"RB_SMALL_COMPONENT_OPS": "1000000"},
                                 {"RB_SHARD_ORDER_ALL": "1", "RB_SMALL_COMPONENT_OPS": "1"}])
@pytest.mark.parametrize("G", [2, 8])
def test_conflict_path_switches_do_not_change_results(monkeypatch, env, G):
    """which runs with a contested counter are replayed (the ordered set decided with the counters' owners, or every one of them as before
    round 6) and whether a lane or a wavefront replays a component are scheduling choices: few k-mers at high multiplicity in small
    filters — most runs share a counter — come out as the oracle's either way"""
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    d = synth.generate_pairs(3000, G=2500, err=0.002, n_rate=1e-3, seed=23, uniform_expr=True)
    sizes = (60_013, 40_009, 9_001)
    og = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 3)
    cl = LoopbackCluster(G, *sizes, 2, 2, 2, 25, False, True, rngSeed=3, mode="split")
    og.set_read_pair_distance(115); cl.setReadPairedKmerDistance(115)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS)
    cl.addBatch(ReadBatch.from_ascii(s, q, off, 3), 150, storeReadPairedKmers=True, reads_per_substep=700)
    check_filters(cl, og)
    assert sum(r.stats["conflict_ops"] for r in cl.ranks) > 1000 and og.cbf_bytes().max() > 24
    cl.destroy()


@pytest.mark.parametrize("overlap", [0, 1, 2])
def test_loopback_lookahead_modes(monkeypatch, overlap):
    """look-ahead hashing of the next sub-batch (replicated-hashing mode) is a scheduling choice only"""
    from rnabloom import sharded
    monkeypatch.setattr(sharded, "_OVERLAP", overlap)
    d = synth.generate_pairs(2000, G=4000, err=0.002, n_rate=1e-3, seed=41, uniform_expr=True)
    sizes = (200_003, 300_007, 40_009)
    og = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 6)
    cl = LoopbackCluster(2, *sizes, 2, 2, 2, 25, False, True, rngSeed=6, mode="replicated")
    og.set_read_pair_distance(115); cl.setReadPairedKmerDistance(115)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS)
    cl.addBatch(ReadBatch.from_ascii(s, q, off, 3), 150, storeReadPairedKmers=True, reads_per_substep=250)
    check_filters(cl, og)
    assert og.cbf_bytes().max() > 24
    cl.destroy()


@pytest.mark.parametrize("G,stranded", [(1, False), (2, False), (4, True), (8, False)])
def test_walks_on_a_sharded_graph(G, stranded):
    """ShardRank.walk (maximum-coverage walks over rb_shard_query exchanges, one exchange per step) must walk the sharded
    graph exactly as rb_graph_walk walks the single-GPU graph built from the same reads: bases, counts, lengths,
    reasons; ranks hold different numbers of seeds (one of them none), seeds with N, a loop."""
    from rnabloom.graph import BloomFilterDeBruijnGraph
    d = synth.generate_pairs(1500, G=4000, err=0.004, n_rate=1e-3, seed=77)
    sizes = (300_007, 1_200_007, 50_021)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    unit = bytes(np.random.default_rng(1).choice(np.frombuffer(b"ACGT", np.uint8), 30)) * 5      # a tandem repeat: walks loop
    s = np.concatenate([s, np.frombuffer(unit, np.uint8)]); q = np.concatenate([q, np.full(len(unit), ord("I"), np.uint8)])
    off = np.concatenate([off, [off[-1] + len(unit)]])
    g1 = BloomFilterDeBruijnGraph(*sizes, 2, 2, 2, 25, stranded, True, rngSeed=9)
    cl = LoopbackCluster(G, *sizes, 2, 2, 2, 25, stranded, True, rngSeed=9)
    g1.addReads(s, q, off, 3)
    cl.addBatch(ReadBatch.from_ascii(s, q, off, 3), 150, reads_per_substep=700)
    rng = np.random.default_rng(4)
    seeds = [bytes(s[off[r] + p: off[r] + p + 25]) for r, p in zip(rng.integers(0, 1500, 240), rng.integers(0, 120, 240))]
    seeds[3] = seeds[3][:8] + b"N" + seeds[3][9:]
    seeds[5] = unit[40:65]
    cuts = [0] + sorted(rng.integers(0, len(seeds), G - 1).tolist()) + [len(seeds)] if G > 1 else [0, len(seeds)]
    if G > 2:
        cuts[1] = cuts[0]                                   # rank 0 has no seeds at all
    per_rank = [seeds[cuts[i]:cuts[i + 1]] for i in range(G)]
    seen = set()
    for direction in (0, 1):
        for min_cov, bound in ((1.0, 45), (3.0, 12)):
            eb, _, _, ec, el, er = g1.walkMaxCov(seeds, direction, bound, min_cov)
            got = cl.walkMaxCov(per_rank, direction, bound, min_cov)
            for rk in range(G):
                bases, c, ln, reason = got[rk]
                a, b = cuts[rk], cuts[rk + 1]
                assert (ln == el[a:b]).all() and (reason == er[a:b]).all()
                m = np.arange(bound)[None, :] < ln[:, None]
                assert (bases[m] == eb[a:b][m]).all() and (c[m] == ec[a:b][m]).all()
            seen |= set(er.tolist())
    assert seen >= {0, 2, 3, 4}
    g1.destroy(); cl.destroy()


@pytest.mark.parametrize("G,stranded", [(1, False), (2, True), (4, False), (8, False)])
def test_get_kmers_and_neighbours_on_a_sharded_graph(G, stranded):
    """graph.getKmers and Kmer.getSuccessors / getPredecessors / variants on the sharded graph (hashes on the asking rank's GPU,
    counts from one query exchange) against the single-GPU graph built from the same reads; ranks ask about different
    sequences, one of them about none; sequences with N and shorter than k."""
    from rnabloom.graph import BloomFilterDeBruijnGraph
    d = synth.generate_pairs(1200, G=3000, err=0.004, n_rate=2e-3, seed=21)
    sizes = (250_007, 1_000_003, 50_021)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    g1 = BloomFilterDeBruijnGraph(*sizes, 2, 2, 2, 25, stranded, True, rngSeed=5)
    cl = LoopbackCluster(G, *sizes, 2, 2, 2, 25, stranded, True, rngSeed=5)
    g1.addReads(s, q, off, 3)
    cl.addBatch(ReadBatch.from_ascii(s, q, off, 3), 150, reads_per_substep=500)
    rng = np.random.default_rng(8)
    reads = [bytes(s[off[i]:off[i + 1]]) for i in rng.integers(0, 1200, 90)] + [b"ACGT", b"", b"ACGTNACGT" * 6]
    rnd = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 80))          # mostly absent k-mers
    reads.append(rnd)
    cuts = [0] + sorted(rng.integers(0, len(reads), G - 1).tolist()) + [len(reads)] if G > 1 else [0, len(reads)]
    if G > 2:
        cuts[1] = cuts[0]
    per_rank = [reads[cuts[i]:cuts[i + 1]] for i in range(G)]
    got = cl.getKmers(per_rank)
    allf, allr, allc = [], [], []
    for rk in range(G):
        ko, f, r, c = got[rk]
        eko, ef, er, ec = g1.getKmers(per_rank[rk])
        assert (ko == eko).all() and (f == ef).all() and (r == er).all() and (c == ec).all()
        allf.append(f); allr.append(r); allc.append(c)
    assert sum((c > 0).sum() for c in allc) > 1000 and sum((c == 0).sum() for c in allc) > 50
    # neighbours of the k-mers found above (first base of each k-mer as charOut for successors, a fixed one otherwise)
    for direction in (0, 1, 2, 3):
        frc = [(allf[rk][:300], allr[rk][:300], np.full(min(300, allf[rk].size), b"ACGT"[direction], np.uint8)) for rk in range(G)]
        res = cl.getNeighbors(frc, direction)
        for rk in range(G):
            ef4, er4, ec4 = g1.getNeighbors(*frc[rk], direction)
            f4, r4, c4 = res[rk]
            assert (f4 == ef4).all() and (r4 == er4).all() and (c4 == ec4).all()
    g1.destroy(); cl.destroy()


# ---- the exchange driver below the C ABI (csrc/rb_comm.hip: rb_shard_add_range) ----
@pytest.mark.parametrize("mode", ["replicated", "split"])
@pytest.mark.parametrize("G", [1, 2, 4, 8])
def test_native_driver_matches_oracle(G, mode):
    """the whole protocol inside the library — one host thread per virtual rank in rb_shard_add_range, exchanges through the
    loopback hub — against the sequential oracle, both files of a library, read pairs, many sub-batches"""
    sizes = (400_003, 3_000_017, 90_001)
    d = synth.generate_pairs(2400, G=25000, err=0.003, n_rate=1e-3, seed=31 + G)
    og = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 9)
    cl = LoopbackCluster(G, *sizes, 2, 2, 2, 25, False, True, rngSeed=9, mode=mode, native=True)
    og.set_read_pair_distance(115); cl.setReadPairedKmerDistance(115)
    for name, rc in (("left", False), ("right", True)):
        s, off = synth.flat(d[name]); q, _ = synth.flat(d[name[0] + "qual"])
        og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
        cl.addBatch(ReadBatch.from_ascii(s, q, off, 3), 150, reverseComplement=rc, storeReadPairedKmers=True, reads_per_substep=577)
        check_filters(cl, og)
    assert sum(r.stats["conflict_ops"] for r in cl.ranks) > 0 and sum(r.stats["reads"] for r in cl.ranks) == 4800
    cl.destroy()


@pytest.mark.parametrize("mode", ["replicated", "split"])
def test_native_driver_high_multiplicity_and_python_driver_agree(mode):
    """counters well into the probabilistic range, hot prefilter cache, look-ahead hashing: the native driver, the Python
    driver and the oracle end in the same filters"""
    d = synth.generate_pairs(5000, G=3000, err=0.001, n_rate=1e-3, seed=3, uniform_expr=True)
    sizes = (100_003, 150_001, 20_011)
    og = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 1)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    og.add_reads(s, q, off, 3, 0)
    b = ReadBatch.from_ascii(s, q, off, 3)
    got = []
    for native in (True, False):
        cl = LoopbackCluster(4, *sizes, 2, 2, 2, 25, False, True, rngSeed=1, mode=mode, native=native)
        cl.addBatch(b, 150, reads_per_substep=500)
        check_filters(cl, og, pairs=False)
        got.append([sum(r.stats[kk] for r in cl.ranks) for kk in ("kmers", "sorted_kmers", "conflict_ops", "distinct")])
        cl.destroy()
    assert got[0][0] == got[1][0] and og.cbf_bytes().max() > 24


def test_native_driver_over_rccl_world_1():
    """the RCCL transport (ncclSend / ncclRecv groups on the handle's stream; librccl through dlopen) with one rank: every
    exchange is a self-send, the rest of the path is what N processes run"""
    from rnabloom import sharded
    from rnabloom.sharded import NativeComm, ShardRank, add_range_native, plan

    class _OneRank:                       # what NativeComm.rccl needs from torch.distributed at world 1
        @staticmethod
        def get_rank(group=None): return 0
        @staticmethod
        def get_world_size(group=None): return 1
    sizes = (400_003, 3_000_017, 90_001)
    d = synth.generate_pairs(2000, G=20000, err=0.003, n_rate=1e-3, seed=5)
    og = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 9)
    og.set_read_pair_distance(115)
    comm = NativeComm.rccl(_OneRank, 0)
    for mode in ("replicated", "split"):
        og2 = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 9); og2.set_read_pair_distance(115)
        rk = ShardRank((*sizes, 2, 2, 2, 25, 0, 1, 0, 0, 9, 0), 0, 1, 0, mode)
        rk.set_read_pair_distance(115)
        pos_bits, _ = plan(150, 25, 1)
        for name, rc in (("left", False), ("right", True)):
            s, off = synth.flat(d[name]); q, _ = synth.flat(d[name[0] + "qual"])
            og2.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
            add_range_native(rk, comm, ReadBatch.from_ascii(s, q, off, 3), 0, 2000, N.ADD_STORE_READ_PAIRS | (N.ADD_REVCOMP if rc else 0), 450, pos_bits)
        assert (rk.local_filter(N.DBGBF) == og2.dbgbf_bytes()).all() and (rk.local_filter(N.CBF) == og2.cbf_bytes()).all()
        assert (rk.local_filter(N.RPKBF) == og2.rpkbf_bytes()).all()
        rk.destroy()
    comm.destroy()


@pytest.mark.parametrize("native", [False, True])
@pytest.mark.parametrize("routed_ranks", [(), (0, 1, 2, 3), (1, 2)])
def test_read_pair_filter_replicated_routed_and_mixed(monkeypatch, routed_ranks, native):
    """Round 4: a rank ORs the read pairs of its reads into a private full-size copy of rpkbf, merged into the owners' shards when the
    insert call ends (rb_shard_pairs_flush_*).  A rank without room for the copy keeps routing its pair indices (RB_SHARD_PAIRS=route),
    and the two kinds of rank work side by side: all replicated, all routed and a mix must leave the oracle's filter, after one call
    and after a second one into the same filters (the copies are cleared by the merge, the shards keep what they had)."""
    import os
    from rnabloom import sharded
    G, sizes = 4, (300_007, 900_001, 120_011)
    d = synth.generate_pairs(2000, G=30000, err=0.002, n_rate=1e-3, seed=23)
    og = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 3)
    cl = LoopbackCluster.__new__(LoopbackCluster)
    params = (*sizes, 2, 2, 2, 25, 0, 1, 0, 0, 3, 0)
    cl.k, cl.count, cl.max_batch = 25, G, sharded.default_batch_kmers(G, "split")
    cl.ranks = []
    for r in range(G):
        if r in routed_ranks: monkeypatch.setenv("RB_SHARD_PAIRS", "route")
        else: monkeypatch.delenv("RB_SHARD_PAIRS", raising=False)
        cl.ranks.append(sharded.ShardRank(params, r, G, 0, "split"))
    monkeypatch.delenv("RB_SHARD_PAIRS", raising=False)
    cl.comm = sharded.NativeComm.loopback(G) if native else None
    og.set_read_pair_distance(115); cl.setReadPairedKmerDistance(115)
    want_pairs = 0
    for name, rc in (("left", False), ("right", True)):
        s, off = synth.flat(d[name]); q, _ = synth.flat(d[name[0] + "qual"])
        want_pairs += og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0)).pairs
        cl.addBatch(ReadBatch.from_ascii(s, q, off, 3), 150, reverseComplement=rc, storeReadPairedKmers=True, reads_per_substep=600)
        check_filters(cl, og)
    assert cl.popcount(N.RPKBF) == og.popcounts()[2] > 10_000
    assert sum(r.stats["pairs"] for r in cl.ranks) == want_pairs > 20_000
    # clear and insert again: nothing of the first round is left in a rank's accumulation copy
    for r in cl.ranks: r.clear()
    og.clear(); og.set_read_pair_distance(115)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS)
    cl.addBatch(ReadBatch.from_ascii(s, q, off, 3), 150, storeReadPairedKmers=True, reads_per_substep=600)
    check_filters(cl, og)
    cl.destroy()
