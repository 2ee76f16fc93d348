"""BASELINE configs[0] and configs[4] inside the driver-run suite at (or near) their stated sizes.

configs[0] — "1M synthetic 150bp single-end reads, k=25": SURVEY.md s8(d) row 1: 1 M x 150 bp as `-sef`, transcriptome of 16 Mb
with UNIFORM expression, no substitutions, nk = 16.5 M (filters 39 MB / 313 MB / 39 MB; a single-end input still fills rpkbf,
R/RNABloom.java:7139-7140, with d = 150 - 25 - 10 = 115).  126 M k-mers + 11 M paired k-mers: the CPU oracle does that in
about 15-25 s, so here the whole config is compared with it BYTE FOR BYTE — all three filters, the statistics, and the counts
read back through the query path.

configs[4] — "5M synthetic ONT long reads (~2kb), k=35, strobemer/minimizer hashing path": 500 000 reads (a tenth; 1.1 G bases,
the full 5 M run is tools/longread_full.py, profiles/r04_longreads.txt) through the k = 35 insert (no pairs:
R/RNABloom.java:1313-1316) into filters sized for nk = 0.6 x bases — the all-new-k-mers regime — checked through the
size-independent properties of tests/test_gpu_fullsize.py (sub-batch invariance of every byte via popcount + device digest, no
false negatives, counts >= 1, occupancy, idempotent bit set) plus the oracle on a sample; strobemers (n = 3, k = 11,
w = [12, 61]) and minimizers (k = 13, w = 15) of a 100 000-read piece: counts per read, positions inside their windows, the
first 200 reads against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import rbo
from rnabloom import _native as N
from rnabloom import graph as G
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch


def test_config0_at_its_size_equals_the_oracle_byte_for_byte():
    n, k, nk = 1_000_000, 25, 16_500_000
    dist = 150 - k - 10
    bits = N.lib.rb_expected_size(nk, 0.01, 2)
    assert 300_000_000 < bits < 330_000_000
    # left reads of 1 M pairs drawn from a 16 Mb transcriptome, expression sigma 0 = uniform, no substitutions, N at 1e-4
    batch = ReadBatch.synthetic(n, 16_000_000, 150, 300, 30, 0.0, 1e-4, 0.0, seed=0x5EED, device=0)
    seq, off = batch.download(0, n)
    assert off.size == n + 1 and (np.diff(off) == 150).all()
    og = rbo.Graph(bits, bits, bits, 2, 2, 2, k, False, True, 7)
    og.set_read_pair_distance(dist)
    so = og.add_reads(seq, None, off, 3, rbo.STORE_READ_PAIRS)            # sequential: the reference's -t 1 order
    g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, k, False, True, device=0, rngSeed=7)
    g.setReadPairedKmerDistance(dist)
    st = g.addBatch(batch, storeReadPairedKmers=True, first=0, n=n)
    assert (st.kmers, st.pairs) == (so.kmers, so.pairs) and st.kmers > 125_000_000 and st.pairs > 10_000_000
    for which, want in ((N.DBGBF, og.dbgbf_bytes()), (N.CBF, og.cbf_bytes()), (N.RPKBF, og.rpkbf_bytes())):
        got = g.exportFilter(which)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, "filter %d differs from the oracle at %d bytes, first %s" % (which, bad.size, bad[:5])
    cbf = og.cbf_bytes()
    assert 2 <= int(cbf.max()) < 64 and int((cbf != 0).sum()) > 10_000_000          # mean coverage ~8x: counters in the deterministic range mostly
    # the same reads through the host-ASCII boundary (rb_graph_add_reads) into a second graph: the same bytes again
    g2 = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, k, False, True, device=0, rngSeed=7)
    g2.setReadPairedKmerDistance(dist)
    g2.addReads(seq, None, off, 3, storeReadPairedKmers=True)
    for which in (N.DBGBF, N.CBF, N.RPKBF):
        assert g2.fold(which) == g.fold(which) and g2.popcount(which) == g.popcount(which)
    g2.destroy()
    # queries at size: getKmers (hashes and counts) of the last 2 000 reads equal the oracle's; nothing inserted is missing
    reads = [seq[off[i]:off[i + 1]].tobytes() for i in range(n - 2000, n)]
    ko, f, r, c = g.getKmers(reads)
    want = [og.get_kmers(x) for x in reads]
    assert np.array_equal(f, np.concatenate([w[0] for w in want])) and np.array_equal(r, np.concatenate([w[1] for w in want]))
    assert np.array_equal(c, np.concatenate([w[2] for w in want]).astype(np.float32))
    h0 = batch.nthash(k, 1, first=n - 20_000, n=20_000)
    assert h0.size > 2_000_000 and bool(np.all(g.contains(h0))) and float(g.getCount(h0).min()) >= 1.0
    g.destroy()


def _long_reads(n_reads, seed, genome):
    """ONT-like reads: log-normal lengths around 2 kb, 5 % substitutions (as tools/longread_full.py)"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    lens = np.clip(rng.lognormal(np.log(2000), 0.5, n_reads), 200, 12000).astype(np.int64)
    starts = rng.integers(0, genome.size - 12000, n_reads)
    off = np.zeros(n_reads + 1, np.int64); np.cumsum(lens, out=off[1:])
    seq = np.empty(int(off[-1]), np.uint8)
    ol, sl, ll = off.tolist(), starts.tolist(), lens.tolist()
    for i in range(n_reads):
        seq[ol[i]:ol[i + 1]] = genome[sl[i]:sl[i] + ll[i]]
    pos = np.cumsum(rng.geometric(0.05, int(seq.size * 0.0525) + 1000)) - 1
    pos = pos[pos < seq.size]
    seq[pos] = acgt[rng.integers(0, 4, pos.size, dtype=np.uint8)]
    return seq, off


def test_config4_at_a_tenth_of_its_size_properties_and_oracle_sample(monkeypatch):
    K, PIECE, PIECES = 35, 100_000, 5
    genome = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(1).integers(0, 4, 100_000_000, dtype=np.uint8)]
    data = [_long_reads(PIECE, 1000 + p, genome) for p in range(PIECES)]
    bases = sum(int(o[-1]) for _, o in data)
    assert bases > 1_000_000_000
    bits = N.lib.rb_expected_size(int(bases * 0.6), 0.01, 2)
    batches = [ReadBatch.from_ascii(s, None, o, 3, device=0) for s, o in data]

    def build(max_batch):
        g = BloomFilterDeBruijnGraph(bits, bits, 0, 2, 2, 1, K, False, False, device=0, rngSeed=1, maxBatchKmers=max_batch)
        km = sum(g.addBatch(b).kmers for b in batches)
        return g, km
    ga, km = build(0)
    assert km > 0.95 * (bases - PIECE * PIECES * (K - 1))
    # occupancy: the set holds ~1.6x the k-mers the filters were sized for (5 % errors make nearly every 35-mer new)
    assert 0 < ga.getDbgbfFPR() < 0.03 and 0 < ga.getCbfFPR() < 0.03
    # no false negatives, every inserted k-mer counts
    for b in (batches[0], batches[-1]):
        h0 = b.nthash(K, 1, first=0, n=2000)
        assert h0.size > 1_000_000 and bool(np.all(ga.contains(h0))) and float(ga.getCount(h0).min()) >= 1.0
    pop = {w: ga.popcount(w) for w in (N.DBGBF, N.CBF)}
    dig = {w: ga.fold(w) for w in (N.DBGBF, N.CBF)}
    # sub-batch invariance: other cut points, no cold-start ramp — every byte of both filters (device digests)
    monkeypatch.setenv("RB_NO_RAMP", "1")
    gb, km_b = build(1 << 27)
    monkeypatch.delenv("RB_NO_RAMP")
    assert km_b == km
    for w in pop:
        assert gb.popcount(w) == pop[w] and gb.fold(w) == dig[w], "filter %d depends on the sub-batch size" % w
    gb.destroy()
    # the oracle on a sample big enough to collide: the first 4 000 reads (8 M k-mers) into small filters, both engines
    seq, off = data[0]
    m = 4000
    sbits = 50_000_017
    og = rbo.Graph(sbits, sbits, 64, 2, 2, 1, K, False, False, 3)
    og.add_reads(seq[: off[m]], None, off[: m + 1], 3, 0)
    gs = BloomFilterDeBruijnGraph(sbits, sbits, 0, 2, 2, 1, K, False, False, device=0, rngSeed=3)
    gs.addBatch(batches[0], first=0, n=m)
    assert np.array_equal(gs.exportFilter(N.DBGBF), og.dbgbf_bytes()) and np.array_equal(gs.exportFilter(N.CBF), og.cbf_bytes())
    gs.destroy()
    # idempotent bit set; counters only grow
    for b in batches[:2]:
        ga.addBatch(b)
    assert ga.popcount(N.DBGBF) == pop[N.DBGBF] and ga.fold(N.DBGBF) == dig[N.DBGBF] and ga.popcount(N.CBF) >= pop[N.CBF]
    ga.destroy()

    # ---- the hashing path of the config: strobemers and minimizers of one piece ----
    SK, SN, SWMIN, SWMAX, MK, MW = 11, 3, 12, 61, 13, 15
    so, sh, ss, se = G.strobemers((seq, off), SK, SN, SWMIN, SWMAX, device=0)
    nk = np.maximum(np.diff(off) - SK + 1, 0)
    cnt = np.where(nk > SWMAX * (SN - 1), nk - SWMAX * (SN - 2) - SWMIN, 0)
    assert np.array_equal(np.diff(so), cnt)
    rd = np.repeat(np.arange(len(cnt)), cnt)
    assert np.array_equal(ss[: so[-1]], np.arange(int(so[-1])) - so[rd])
    span = se[: so[-1]] - (SK - 1) - ss[: so[-1]]
    assert int(span.min()) >= SWMAX * (SN - 2) + SWMIN and int(span.max()) < SWMAX * (SN - 1)
    mo, mh, mp = G.minimizers((seq, off), MK, MW, 1, device=0)
    wn = np.maximum(np.maximum(np.diff(off) - MK + 1, 0) - MW + 1, 0)
    assert np.array_equal(np.diff(mo), wn)
    pw = np.arange(int(mo[-1])) - mo[np.repeat(np.arange(len(wn)), wn)]
    assert bool(np.all((mp[: mo[-1]] >= pw) & (mp[: mo[-1]] < pw + MW)))
    take = 200
    reads = [seq[off[i]:off[i + 1]].tobytes() for i in range(take)]
    o3 = [rbo.strobemers(r, SK, SN, SWMIN, SWMAX) for r in reads]
    for j, got in enumerate((sh, ss, se)):
        assert np.array_equal(got[: so[take]], np.concatenate([o[j] for o in o3]))
    o2 = [rbo.minimizers(r, MK, MW, 1) for r in reads]
    for j, got in enumerate((mh, mp)):
        assert np.array_equal(got[: mo[take]], np.concatenate([o[j] for o in o2]))
