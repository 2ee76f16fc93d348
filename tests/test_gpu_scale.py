"""Parity at a size where the device is full: 200 000+ reads (tens of millions of k-mers, hundreds of thousands of
lanes) against the CPU oracle, bit for bit.  The small parity tests keep every kernel within one wavefront per SIMD;
a defect that needs several wavefronts of one kernel on a SIMD, or more than 65536 threads, only shows here (the
first paired-k-mer kernel was exact on every small test and wrong from 66 000 reads on).  One case per code path:
the k <= 31 fast kernels, the generic kernels (k = 35), stranded + count-if-present, ragged reads, the per-hash
operations, and the sharded engine in both modes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import rbo
from rnabloom import _native as N
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch
from rnabloom.sharded import LoopbackCluster

BITS = 400_000_009          # sparse filters: almost every wrong bit stays visible


def same_state(og, g, pairs=True):
    assert np.array_equal(g.exportFilter(N.DBGBF), og.dbgbf_bytes()), "dbgbf differs"
    cg, co = g.exportFilter(N.CBF), og.cbf_bytes()
    bad = np.nonzero(cg != co)[0]
    assert bad.size == 0, "cbf differs at %d bytes, first %s" % (bad.size, bad[:5])
    if pairs:
        assert np.array_equal(g.exportFilter(N.RPKBF), og.rpkbf_bytes()), "rpkbf differs"


def synthetic(n_reads, genome=3_000_000, err=0.002, n_rate=1e-3, seed=5, read_len=150):
    batch = ReadBatch.synthetic(n_reads // 2, genome, read_len, 2 * read_len, 30, err, n_rate, 2.0, seed=seed, device=0)
    seq, off = batch.download(0, n_reads)
    return batch, seq, off


@pytest.mark.parametrize("k,stranded,dist,read_len", [(25, False, 115, 150), (31, True, 100, 150), (35, False, 90, 150), (64, True, 40, 150),
                                                      (25, False, 250, 300), (33, False, 60, 100)])
def test_insert_at_scale(k, stranded, dist, read_len):
    n = 200_000 if read_len <= 150 else 100_000
    batch, seq, off = synthetic(n, seed=k + read_len, read_len=read_len)
    og = rbo.Graph(BITS, BITS, BITS, 2, 2, 2, k, stranded, True, 3)
    g = BloomFilterDeBruijnGraph(BITS, BITS, BITS, 2, 2, 2, k, stranded, True, rngSeed=3)
    og.set_read_pair_distance(dist); g.setReadPairedKmerDistance(dist)
    og.add_reads(seq, None, off, 3, rbo.STORE_READ_PAIRS)
    st = g.addBatch(batch, storeReadPairedKmers=True, first=0, n=n)
    assert st.kmers > (read_len - k) * n // 2 and st.pairs > n
    same_state(og, g)
    # second pass over the same reads, reverse-complemented, counting only what is present
    og.add_reads(seq, None, off, 3, rbo.REVCOMP | rbo.COUNT_IF_PRESENT)
    g.addBatch(batch, reverseComplement=True, incrementIfPresent=True, first=0, n=n)
    same_state(og, g)
    g.destroy()


def test_ragged_reads_at_scale():
    rng = np.random.default_rng(11)
    genome = rng.integers(0, 4, 2_000_000, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    n = 150_000
    lens = rng.integers(20, 260, n)
    starts = rng.integers(0, genome.size - 260, n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    seq = np.empty(int(off[-1]), np.uint8)
    for i in range(n):
        seq[off[i]:off[i + 1]] = acgt[genome[starts[i]:starts[i] + lens[i]]]
    seq[rng.integers(0, seq.size, seq.size // 800)] = ord("N")
    og = rbo.Graph(BITS, BITS, BITS, 2, 2, 2, 25, False, True, 4)
    g = BloomFilterDeBruijnGraph(BITS, BITS, BITS, 2, 2, 2, 25, False, True, rngSeed=4)
    og.set_read_pair_distance(60); g.setReadPairedKmerDistance(60)
    og.add_reads(seq, None, off, 3, rbo.STORE_READ_PAIRS)
    st = g.addReads(seq, None, off, 3, storeReadPairedKmers=True)
    assert st.pairs > n
    same_state(og, g)
    g.destroy()


def _ntm64_rows(base, k):
    """hashVals[2] of NTHash.NTM64 for an array of base hashes (R/bloom/hash/NTHash.java:518-527; oracle rbo_ntm64)"""
    kmul = np.uint64((k * 0x90b45d39fb6da1fa) & 0xFFFFFFFFFFFFFFFF)
    t = base * (np.uint64(1) ^ kmul)
    out = np.empty((base.size, 2), np.uint64)
    out[:, 0] = base
    out[:, 1] = t ^ (t >> np.uint64(27))
    return out


def _oracle_each(og, fn_name, base, k=25, result=None):
    """one oracle call per hash, in order (the oracle's per-hash entry points take one k-mer's hashVals)"""
    import ctypes as C
    rows = _ntm64_rows(base, k)
    assert (rows[:5] == np.stack([rbo.ntm64(int(b), k, 2) for b in base[:5]])).all()
    fn = getattr(og.L, fn_name)
    addr = rows.ctypes.data
    if result is None:
        for i in range(base.size):
            fn(og.g, C.c_void_p(addr + 16 * i))
        return None
    out = np.empty(base.size, result)
    for i in range(base.size):
        out[i] = fn(og.g, C.c_void_p(addr + 16 * i))
    return out


def test_per_hash_operations_at_scale():
    with np.errstate(over="ignore"):
        rng = np.random.default_rng(21)
        h = rng.integers(0, 1 << 63, 150_000, dtype=np.int64).astype(np.uint64)
        h = np.concatenate([h, h[::3], h[:50_000]])            # repeats: counters climb, add-if-absent meets both cases
        og = rbo.Graph(BITS // 64, BITS // 64, BITS // 64, 2, 2, 2, 25, False, True, 6)
        g = BloomFilterDeBruijnGraph(BITS // 64, BITS // 64, BITS // 64, 2, 2, 2, 25, False, True, rngSeed=6)
        for name_o, name_g, part in (("rbo_graph_add", "add", h[:120_000]), ("rbo_graph_add_if_absent", "addIfAbsent", h[60_000:200_000]),
                                     ("rbo_graph_add_count_if_present", "addCountIfPresent", h[100_000:]),
                                     ("rbo_graph_add_read_pair", "addReadSingleKmerPair", h[:100_000])):
            _oracle_each(og, name_o, part); getattr(g, name_g)(part)
        same_state(og, g)
        q = np.concatenate([h[:100_000], rng.integers(0, 1 << 63, 100_000, dtype=np.int64).astype(np.uint64)])
        assert np.array_equal(g.contains(q), _oracle_each(og, "rbo_graph_contains", q, result=np.int32).astype(bool))
        assert np.array_equal(g.getCount(q), _oracle_each(og, "rbo_graph_get_count", q, result=np.float32))
        assert np.array_equal(g.lookupReadKmerPair(q), _oracle_each(og, "rbo_graph_lookup_read_pair", q, result=np.int32).astype(bool))
        g.destroy()


@pytest.mark.parametrize("G,mode,general_pairs", [(2, "replicated", False), (4, "split", False), (8, "split", False), (2, "replicated", True), (4, "split", True)])
def test_sharded_engine_at_scale(monkeypatch, G, mode, general_pairs):
    if general_pairs:       # the general paired-k-mer kernel in its index-collecting instantiation (k_pairs_insert<M, true>)
        monkeypatch.setenv("RB_PAIRS_GENERAL", "1")
    n = 200_000
    batch, seq, off = synthetic(n, seed=40 + G)
    og = rbo.Graph(BITS, BITS, BITS, 2, 2, 2, 25, False, True, 8)
    cl = LoopbackCluster(G, BITS, BITS, BITS, 2, 2, 2, 25, False, True, rngSeed=8, mode=mode)
    og.set_read_pair_distance(115); cl.setReadPairedKmerDistance(115)
    og.add_reads(seq, None, off, 3, rbo.STORE_READ_PAIRS)
    cl.addBatch(batch, 150, storeReadPairedKmers=True, first=0, n=n, reads_per_substep=70_000)
    same_state(og, cl)
    cl.destroy()


@pytest.mark.parametrize("k,err", [(25, 0.002), (35, 0.05)])
def test_index_keyed_grouping_at_scale(monkeypatch, k, err):
    """RB_GROUP_IDX=1: the first partition digit of the grouping comes from the k-mer's first filter index (csrc/rb_group.hip GrIdx; on
    by itself when a sub-batch is mostly new k-mers).  Any grouping key that is a function of the hash is valid — the filters must
    equal the oracle's on one GPU (many sub-batches, the 5 % error rate of long reads included: nearly every k-mer new) and on 4
    virtual ranks, whose records all lie in the rank's own index range."""
    monkeypatch.setenv("RB_GROUP_IDX", "1")
    monkeypatch.setenv("RB_SYNTH_KEEP_ERRORS", "1")          # substituted bases stay usable: at 5 % nearly every 35-mer is new
    n = 160_000
    batch, seq, off = synthetic(n, seed=90 + k, err=err)
    dist = 150 - k - 10
    og = rbo.Graph(BITS, BITS, BITS, 2, 2, 2, k, False, True, 6)
    og.set_read_pair_distance(dist)
    og.add_reads(seq, None, off, 0, rbo.STORE_READ_PAIRS)
    # min_base_qual 0 below + usable flags: the batch's substituted bases are masked on the device; the oracle gets the same bases
    g = BloomFilterDeBruijnGraph(BITS, BITS, BITS, 2, 2, 2, k, False, True, rngSeed=6, maxBatchKmers=3_000_000)
    g.setReadPairedKmerDistance(dist)
    st = g.addReads(seq, None, off, 0, storeReadPairedKmers=True)
    same_state(og, g)
    assert g.popcount(N.DBGBF) > (4_000_000 if k == 25 else 25_000_000)      # (k = 35 at 5 % unmasked errors: nearly every window a new k-mer)
    g.destroy()
    cl = LoopbackCluster(4, BITS, BITS, BITS, 2, 2, 2, k, False, True, rngSeed=6, mode="split")
    cl.setReadPairedKmerDistance(dist)
    b2 = ReadBatch.from_ascii(seq, None, off, 0, device=0)
    cl.addBatch(b2, 150, storeReadPairedKmers=True, first=0, n=n, reads_per_substep=50_000)
    same_state(og, cl)
    cl.destroy()


def test_queries_at_scale():
    """getKmers over 100 000 reads in one call, the 4 successors / predecessors of ~350 000 k-mers, and the window
    hashes of a whole 200 000-read batch (canonical and strand-specific), all against the oracle — the oracle is asked
    read by read for a spread sample (beginning, middle, end of the launch: the high thread indices are the point)."""
    n = 200_000
    batch, seq, off = synthetic(n, seed=77)
    small = BITS // 16
    og = rbo.Graph(small, small, small, 2, 2, 2, 25, False, False, 2)
    g = BloomFilterDeBruijnGraph(small, small, small, 2, 2, 2, 25, False, False, rngSeed=2)
    og.add_reads(seq, None, off, 3, 0); g.addBatch(batch, first=0, n=n)
    same_state(og, g, pairs=False)
    # getKmers: 100 000 reads at once
    first = 60_000
    reads = [seq[off[i]:off[i + 1]].tobytes() for i in range(first, first + 100_000)]
    ko, f, r, c = g.getKmers(reads)
    assert ko[-1] > 10_000_000
    sample = list(range(0, 300)) + list(range(50_000, 50_300)) + list(range(99_700, 100_000))
    for i in sample:
        ef, er, ec = og.get_kmers(reads[i])
        a, b = ko[i], ko[i + 1]
        assert b - a == len(ef) and (f[a:b] == ef).all() and (r[a:b] == er).all() and (c[a:b] == ec).all()
    # neighbours of every third k-mer of the last 10 000 reads' worth of output
    idx = np.arange(int(ko[90_000]), int(ko[-1]), 3)
    assert idx.size > 300_000
    ch = np.full(idx.size, ord("A"), np.uint8)
    for direction in (0, 1):
        f4, r4, c4 = g.getNeighbors(f[idx], r[idx], ch, direction)
        for j in list(range(0, 400)) + list(range(idx.size - 400, idx.size)):
            of, orr, oc = og.neighbors(f[idx[j]], r[idx[j]], ord("A"), direction)
            assert (f4[j] == of).all() and (r4[j] == orr).all() and (c4[j] == oc).all()
    # window hashes of the whole batch
    for mode in (0, 1, 2):
        h0, rd, ps = batch.nthash(25, mode, first=0, n=n, with_positions=True)
        assert h0.size > 20_000_000
        for i in list(range(0, 200)) + list(range(n - 200, n)):
            s = seq[off[i]:off[i + 1]].tobytes()
            sel = rd == i
            exp = []
            for a, b in rbo.segments(s, None, 25, 3):
                hv, _ = rbo.hash_region(s, 25, 1, mode, a, b)
                exp.append(hv[:, 0])
            exp = np.concatenate(exp) if exp else np.zeros(0, np.uint64)
            assert (h0[sel] == exp).all()
    g.destroy()


def test_walks_at_scale():
    """300 000 greedy maximum-coverage walks in one rb_graph_walk call per direction (one lane per walk, up to 80 steps):
    a spread sample against the step-by-step oracle, plus invariants of every walk (each step's count is at least
    min_cov; a walk that stopped at the bound has that many steps)."""
    n = 100_000
    batch, seq, off = synthetic(n, genome=400_000, seed=123)
    small = BITS // 16
    og = rbo.Graph(small, small, small, 2, 2, 2, 25, False, False, 2)
    g = BloomFilterDeBruijnGraph(small, small, small, 2, 2, 2, 25, False, False, rngSeed=2)
    og.add_reads(seq, None, off, 3, 0); g.addBatch(batch, first=0, n=n)
    rng = np.random.default_rng(3)
    reads = rng.integers(0, n, 300_000); pos = rng.integers(0, 100, 300_000)
    seeds = [seq[off[r] + p: off[r] + p + 25].tobytes() for r, p in zip(reads, pos)]
    for direction in (0, 1):
        bases, f, r, c, ln, reason = g.walkMaxCov(seeds, direction, 80, 2.0)
        ok = np.arange(80)[None, :] < ln[:, None]
        assert (c[ok] >= 2.0).all() and (ln[reason == 3] == 80).all() and (ln[reason == 4] == 0).all()
        assert (reason == 3).sum() > 1000 and (reason == 0).sum() > 1000
        for i in list(range(0, 150)) + list(range(len(seeds) - 150, len(seeds))):
            eb, ec, er = rbo.walk_max_cov(og, seeds[i], direction, 80, 2.0)
            assert (int(ln[i]), int(reason[i])) == (len(eb), er) and bytes(bases[i, :ln[i]]) == eb
            assert (c[i, :ln[i]] == np.array(ec, np.float32)).all()
        # the same seeds through the lookahead extension (GraphUtils.greedyExtendRight / Left)
        bases, c, ln, reason = g.greedyExtend(seeds, direction, 4, 40)
        assert ((reason == 3) == (ln == 40)).all() and (reason == 3).sum() > 1000
        for i in list(range(0, 100)) + list(range(len(seeds) - 100, len(seeds))):
            eb, ec = rbo.greedy_extend(og, seeds[i].upper(), direction, 4, 40) if b"N" not in seeds[i] else (b"", [])
            assert int(ln[i]) == len(eb) and bytes(bases[i, :ln[i]]) == eb and (c[i, :ln[i]] == np.array(ec, np.float32)).all()
    g.destroy()
