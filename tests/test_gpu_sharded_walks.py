"""Graph traversal on a SHARDED graph (rb_shard_trav_*): maximum-coverage walks, greedy extension with lookahead and naive
extension run on each rank's GPU in the kernels of the single-GPU calls, the counts coming from the owners of the filter ranges
through query exchanges.  Reference for every check: the single-GPU graph built from the same reads (itself compared with the
oracle's step-by-step restatements in test_gpu_parity.py / test_gpu_api_holes.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from rnabloom import synth
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch
from rnabloom.sharded import LoopbackCluster


def build(G, stranded, sizes=(150_001, 600_011, 10_007), err=0.02, seed=57, n=1500):
    """a graph with plenty of branches: small filters => false-positive neighbours (as test_greedy_extend_with_lookahead_matches_oracle)"""
    d = synth.generate_pairs(n, G=3000, err=err, n_rate=0.0, seed=seed)
    s, off = synth.flat(d["left"]); q, _ = synth.flat(d["lqual"])
    unit = bytes(np.random.default_rng(1).choice(np.frombuffer(b"ACGT", np.uint8), 30)) * 5      # a tandem repeat: walks loop
    s = np.concatenate([s, np.frombuffer(unit, np.uint8)]); q = np.concatenate([q, np.full(len(unit), ord("I"), np.uint8)])
    off = np.concatenate([off, [off[-1] + len(unit)]])
    g1 = BloomFilterDeBruijnGraph(*sizes, 2, 2, 2, 25, stranded, False, rngSeed=9)
    cl = LoopbackCluster(G, *sizes, 2, 2, 2, 25, stranded, False, rngSeed=9)
    g1.addReads(s, q, off, 3)
    cl.addBatch(ReadBatch.from_ascii(s, q, off, 3), 150, reads_per_substep=700)
    rng = np.random.default_rng(4 + G)
    seeds = [bytes(s[off[r] + p: off[r] + p + 25]) for r, p in zip(rng.integers(0, n, 200), rng.integers(0, 120, 200))]
    seeds[3] = seeds[3][:8] + b"N" + seeds[3][9:]
    seeds[5] = unit[40:65]
    cuts = [0] + sorted(rng.integers(0, len(seeds), G - 1).tolist()) + [len(seeds)] if G > 1 else [0, len(seeds)]
    if G > 2:
        cuts[1] = cuts[0]                                   # rank 0 has no seeds at all
    reads = [bytes(s[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    return g1, cl, seeds, cuts, reads


def split(x, cuts):
    return [x[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]


@pytest.mark.parametrize("G,stranded", [(1, False), (2, True), (4, False)])
def test_max_coverage_walks_with_targets(G, stranded):
    g1, cl, seeds, cuts, reads = build(G, stranded)
    seen = set()
    for direction in (0, 1):
        for min_cov, bound in ((1.0, 45), (3.0, 12)):
            free = g1.walkMaxCov(seeds, direction, bound, min_cov)
            # targets: the k-mer each free walk reaches after a few steps (so the targeted walk stops there), or an unrelated one
            tg = []
            for i, sd in enumerate(seeds):
                ln = int(free[4][i])
                if ln >= 4 and i % 3:
                    path = (sd + bytes(free[0][i, :ln])) if direction == 0 else (bytes(free[0][i, :ln][::-1]) + sd)
                    tg.append(path[4:29] if direction == 0 else path[len(path) - 29:len(path) - 4])
                else:
                    tg.append(seeds[(i + 7) % len(seeds)].replace(b"N", b"A"))
            for targets in (None, tg):
                eb, ef, er_, ec, el, er = g1.walkMaxCov(seeds, direction, bound, min_cov, targets)
                got = cl.traverse(0, split(seeds, cuts), direction, bound=bound, min_cov=min_cov, targets=split(targets, cuts) if targets else None)
                for rk in range(G):
                    bases, f, r, c, ln, reason, rounds = got[rk]
                    a, b = cuts[rk], cuts[rk + 1]
                    assert (ln == el[a:b]).all() and (reason == er[a:b]).all()
                    m = np.arange(bound)[None, :] < ln[:, None]
                    assert (bases[m] == eb[a:b][m]).all() and (c[m] == ec[a:b][m]).all() and (f[m] == ef[a:b][m]).all() and (r[m] == er_[a:b][m]).all()
                    if b > a:
                        assert rounds <= bound + 2                     # one exchange round per step
                seen |= set(er.tolist())
    assert seen >= {0, 1, 2, 3, 4}
    g1.destroy(); cl.destroy()


@pytest.mark.parametrize("G,stranded", [(1, False), (2, True), (4, False)])
def test_greedy_extension_with_lookahead(G, stranded):
    g1, cl, seeds, cuts, _ = build(G, stranded)
    branched = 0
    for direction in (0, 1):
        for lookahead, bound in ((3, 40), (5, 25), (0, 10), (1, 10), (6, 12)):
            eb, ec, el, er = g1.greedyExtend(seeds, direction, lookahead, bound)
            got = cl.greedyExtend(split(seeds, cuts), direction, lookahead, bound, answer_cap=1024)
            for rk in range(G):
                bases, c, ln, reason = got[rk]
                a, b = cuts[rk], cuts[rk + 1]
                assert (ln == el[a:b]).all() and (reason == er[a:b]).all(), (direction, lookahead, rk)
                m = np.arange(bound)[None, :] < ln[:, None]
                assert (bases[m] == eb[a:b][m]).all() and (c[m] == ec[a:b][m]).all()
            if lookahead == 3:
                plain = g1.walkMaxCov(seeds, direction, bound, 1.0, hashes=False)[0]
                branched += int((plain != eb).any(axis=1).sum())
    assert branched > 0
    # a cache too small for the lookahead search: those walks say so (reason 8) and stop where they were; the others are untouched
    eb, ec, el, er = g1.greedyExtend(seeds, 0, 6, 30)
    got = cl.greedyExtend(split(seeds, cuts), 0, 6, 30, answer_cap=12)
    wide = 0
    for rk in range(G):
        bases, c, ln, reason = got[rk]
        a, b = cuts[rk], cuts[rk + 1]
        ok = reason != 8
        wide += int((~ok).sum())
        assert (ln[ok] == el[a:b][ok]).all() and (reason[ok] == er[a:b][ok]).all()
        assert (ln[~ok] <= el[a:b][~ok]).all()
        for i in np.nonzero(~ok)[0]:
            assert (bases[i, :ln[i]] == eb[a + i, :ln[i]]).all()
    assert wide > 0
    g1.destroy(); cl.destroy()


@pytest.mark.parametrize("G,stranded", [(1, True), (2, False), (8, False)])
def test_naive_extension_three_forms(G, stranded):
    g1, cl, seeds, cuts, reads = build(G, stranded, sizes=(400_003, 1_500_007, 10_007), err=0.004)
    rng = np.random.default_rng(12)
    frags = [reads[int(x)] for x in rng.integers(0, len(reads), len(seeds))]          # terminators: the k-mers of some read
    for i in range(0, len(seeds), 4):                                                   # a seed inside its own terminator sequence
        frags[i] = reads[i % len(reads)]
        seeds[i] = frags[i][10:35] if b"N" not in frags[i][10:35] else seeds[i]
    seen = set()
    for direction in (0, 1):
        for mode, kw in ((0, dict(cap=64)), (0, dict(cap=3)), (1, dict(bound=20)), (2, dict(bound=20)), (1, dict(bound=0)), (1, dict(bound=30, minKmerCov=2.0))):
            eb, er = g1.naiveExtend(seeds, direction, mode, terminators=frags if mode == 0 else None, **kw)
            got = cl.naiveExtend(split(seeds, cuts), direction, mode, terminators=split(frags, cuts) if mode == 0 else None, **kw)
            for rk in range(G):
                a, b = cuts[rk], cuts[rk + 1]
                assert got[rk][0] == eb[a:b] and (got[rk][1] == er[a:b]).all(), (direction, mode, kw, rk)
            seen |= set(er.tolist())
    assert seen >= {0, 1, 3, 4, 5, 6}                    # (2, several neighbours, needs a false-positive branch without a back branch: not in every graph)
    g1.destroy(); cl.destroy()


@pytest.mark.parametrize("G", [1, 4])
def test_greedy_extension_through_a_gate_filter(G):
    """the `bf` variants (a neighbour must pass bf.lookup before its count is read): on a sharded graph the gate is another sharded
    graph's dbgbf, looked up by its owners in the same exchange round; reference: rb_graph_greedy_extend with a stand-alone filter
    holding the same k-mers (itself checked against the oracle in test_greedy_extend_with_bloom_filter_gate)"""
    from rnabloom.bloom import BloomFilter
    g1, cl, seeds, cuts, reads = build(G, False, err=0.01, seed=67)
    third = reads[:500]
    s = np.frombuffer(b"".join(third), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(r) for r in third])]).astype(np.int64)
    gb = ReadBatch.from_ascii(s, None, off, 3)
    bf = BloomFilter(400_009, 2, 25)
    bf.add(gb.nthash(25, 1))
    gate = LoopbackCluster(G, 400_009, 400_009, 0, 2, 2, 2, 25, False, False, rngSeed=1)
    gate.addBatch(gb, 150, reads_per_substep=200)
    from rnabloom import _native as N
    assert (gate.exportFilter(N.DBGBF) == bf.toBytes()).all()
    shorter = 0
    for direction in (0, 1):
        eb, ec, el, er = g1.greedyExtend(seeds, direction, 4, 30, bf=bf)
        free = g1.greedyExtend(seeds, direction, 4, 30)[2]
        got = cl.greedyExtend(split(seeds, cuts), direction, 4, 30, answer_cap=1024, bf=gate)
        for rk in range(G):
            bases, c, ln, reason = got[rk]
            a, b = cuts[rk], cuts[rk + 1]
            assert (ln == el[a:b]).all() and (reason == er[a:b]).all()
            m = np.arange(30)[None, :] < ln[:, None]
            assert (bases[m] == eb[a:b][m]).all() and (c[m] == ec[a:b][m]).all()
        shorter += int((el < free).sum())
    assert shorter > 0                      # the gate did stop walks
    g1.destroy(); cl.destroy(); gate.destroy(); bf.destroy()


def test_few_walks_on_a_rank_ask_for_more_than_the_request_buffer_holds():
    """rb_shard_trav_begin sizes the per-round request buffer at 8 entries per walk, and a branchy greedy step files 4
    requests for every neighbourhood its lookahead search opens (12-16 with two or three candidates).  With one or two walks on a rank the
    round's requests exceed the buffer: the kernel drops what does not fit, rb_shard_trav_advance sends what did (it used to fail the whole
    traversal, leaving the peers in the all-to-all), and the suspended walk asks for the rest when its step is replayed.  Results must
    equal the single-GPU calls — one or two seeds per rank, deep lookahead, the branchy graph."""
    g1, cl, seeds, _, _ = build(2, False)
    picked = 0
    for base in range(0, 60, 3):
        mine = [seeds[base:base + 1], seeds[base + 1:base + 3]]                 # rank 0: one walk, rank 1: two
        flat = mine[0] + mine[1]
        for direction, lookahead, bound in ((0, 6, 30), (1, 5, 25)):
            eb, ec, el, er = g1.greedyExtend(flat, direction, lookahead, bound)
            got = cl.greedyExtend(mine, direction, lookahead, bound, answer_cap=1024)
            at = 0
            for rk in range(2):
                bases, c, ln, reason = got[rk]
                n = len(mine[rk])
                assert (ln == el[at:at + n]).all() and (reason == er[at:at + n]).all(), (base, direction, rk)
                m = np.arange(bound)[None, :] < ln[:, None]
                assert (bases[m] == eb[at:at + n][m]).all() and (c[m] == ec[at:at + n][m]).all()
                at += n
            picked += int((el > 3).sum())
    assert picked > 20                                                          # the walks did get somewhere
    g1.destroy(); cl.destroy()
