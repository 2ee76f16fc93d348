"""BASELINE configs[3] ("200M 150bp paired-end (pooled multi-sample), k=25, 8xMI355X — Bloom > HBM of one GPU") through the
REAL insert path at the largest filter size one GPU holds: filters sized for nk = 7.5 G distinct k-mers (142 G bits /
counters each: every index needs 38 bits; 17.8 + 142.4 + 17.8 GB), a pooled library of four samples inserted in the
reference's file order — all forward files, then all reverse files (R/RNABloom.java:1290-1316, populateGraph2) — by

  * the single-GPU engine (38-bit indices in index_of, prefilter caches at their caps, u32 occurrence ids),
  * the single-GPU engine cut into other sub-batches (order-exactness: not a bit may move),
  * the sharded engine with 8 virtual ranks (the rank count the config names): routing by 38-bit index / span,
    owner-relative addressing, replicated cache, conflict components across ranks.

150 GB of counters cannot be exported, so the three filters are compared ON THE DEVICE: popcount + rb_filter_fold (a
64-bit digest whose shard values add up to the whole filter's).  The digest itself is pinned to the exported bytes at a
small size (test_fold_is_the_digest_of_the_exported_bytes), where the same engines equal the oracle bit for bit
(tests/test_gpu_parity.py, tests/test_gpu_sharded.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu

K, FPR = 25, 0.01
FILTERS = (0, 1, 2)          # DBGBF, CBF, RPKBF


def _pooled_insert(add, batches, pairs_per_sample):
    """populateGraph2's order for a pooled library: every sample's forward file, then every sample's reverse file
    (reverse-complemented), read-paired k-mers stored"""
    kmers = pairs = 0
    for b in batches:
        st = add(b, False, 0, pairs_per_sample)
        if st is not None: kmers += st.kmers; pairs += st.pairs
    for b in batches:
        st = add(b, True, pairs_per_sample, pairs_per_sample)
        if st is not None: kmers += st.kmers; pairs += st.pairs
    return kmers, pairs


def test_fold_is_the_digest_of_the_exported_bytes():
    from rnabloom import _native as N
    from rnabloom import sharded
    from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch, fold_bytes

    sizes = (3_000_017, 2_400_011, 1_000_003)
    batch = ReadBatch.synthetic(20_000, 400_000, 150, 300, 30, 0.002, 1e-3, 2.0, seed=3, device=0)
    g = BloomFilterDeBruijnGraph(*sizes, 2, 2, 2, K, False, True, device=0, rngSeed=9)
    g.setReadPairedKmerDistance(115)
    g.addBatch(batch, storeReadPairedKmers=True, first=0, n=20_000)
    g.addBatch(batch, reverseComplement=True, storeReadPairedKmers=True, first=20_000, n=20_000)
    want = {}
    for w in FILTERS:
        data = g.exportFilter(w)
        want[w] = fold_bytes(data)
        assert g.fold(w) == want[w] and want[w] != 0
        # the digest sees every byte and its place
        flipped = data.copy(); flipped[len(flipped) // 3] ^= 1
        assert fold_bytes(flipped) != want[w]
        assert fold_bytes(np.roll(data, 4)) != want[w]
    for G in (2, 8):
        cl = sharded.LoopbackCluster(G, *sizes, 2, 2, 2, K, False, True, device=0, rngSeed=9)
        cl.setReadPairedKmerDistance(115)
        cl.addBatch(batch, 150, storeReadPairedKmers=True, first=0, n=20_000)
        cl.addBatch(batch, 150, reverseComplement=True, storeReadPairedKmers=True, first=20_000, n=20_000)
        for w in FILTERS:
            assert cl.fold(w) == want[w], "the shards' digests do not add up to the filter's (filter %d, %d ranks)" % (w, G)
            assert cl.popcount(w) == g.popcount(w)
        cl.destroy()
    g.destroy()


@pytest.mark.parametrize("nk", [7_500_000_000])
def test_config3_shape_at_the_largest_size_one_gpu_holds(nk, monkeypatch):
    from rnabloom import _native as N
    from rnabloom import sharded
    from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch

    SAMPLES, PAIRS = 4, 2_000_000                 # 16 M reads = 2.0 G k-mers, 8 files
    GENOME = 256_000_000
    bits = N.lib.rb_expected_size(nk, FPR, 2)
    assert bits > (1 << 37)                       # indices of 38 bits
    batches = [ReadBatch.synthetic(PAIRS, GENOME, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0xC0F3, device=0,
                                   pair_offset=s * PAIRS, total_pairs=SAMPLES * PAIRS) for s in range(SAMPLES)]

    def graph(max_batch):
        g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, K, False, True, device=0, rngSeed=1, maxBatchKmers=max_batch)
        g.setReadPairedKmerDistance(150 - K - 10)
        return g

    # ---- single-GPU engine ----
    ga = graph(1 << 28)
    kmers, pairs = _pooled_insert(lambda b, rc, first, n: ga.addBatch(b, reverseComplement=rc, storeReadPairedKmers=True, first=first, n=n),
                                  batches, PAIRS)
    assert kmers > 1_900_000_000 and pairs > 140_000_000
    pop = {w: ga.popcount(w) for w in FILTERS}
    dig = {w: ga.fold(w) for w in FILTERS}
    assert pop[0] > 100_000_000 and pop[1] > 0.3 * pop[0] and pop[2] > 100_000_000      # a counter is touched from a k-mer's second sighting on
    assert 0 < ga.getDbgbfFPR() < FPR and 0 < ga.getCbfFPR() < FPR and 0 < ga.getRpkbfFPR() < FPR
    # no false negatives; every inserted k-mer counts at least once
    for b, first in ((batches[0], 0), (batches[2], PAIRS - 10_000), (batches[3], 2 * PAIRS - 20_000)):
        h0 = b.nthash(K, 1, first=first, n=20_000)
        assert h0.size > 2_000_000
        assert bool(np.all(ga.contains(h0)))
        assert int(ga.getCount(h0).min()) >= 1
    # idempotence of the bit sets on the first sample's forward file; counters only grow
    ga.addBatch(batches[0], storeReadPairedKmers=True, first=0, n=PAIRS)
    assert ga.popcount(0) == pop[0] and ga.fold(0) == dig[0]
    assert ga.popcount(2) == pop[2] and ga.fold(2) == dig[2]
    assert ga.popcount(1) >= pop[1] and ga.fold(1) != dig[1]
    ga.destroy()

    # ---- other sub-batch cuts, no cold-start ramp ----
    monkeypatch.setenv("RB_NO_RAMP", "1")
    gb = graph(1 << 26)
    assert _pooled_insert(lambda b, rc, first, n: gb.addBatch(b, reverseComplement=rc, storeReadPairedKmers=True, first=first, n=n),
                          batches, PAIRS) == (kmers, pairs)
    monkeypatch.delenv("RB_NO_RAMP")
    for w in FILTERS:
        assert gb.popcount(w) == pop[w] and gb.fold(w) == dig[w], "filter %d depends on the sub-batch size" % w
    gb.destroy()

    # ---- 8 virtual ranks of the sharded engine: each holds 1/8 of every filter ----
    for native in (False, True):
        cl = sharded.LoopbackCluster(8, bits, bits, bits, 2, 2, 2, K, False, True, device=0, rngSeed=1, maxBatchKmers=1 << 29, native=native)
        cl.setReadPairedKmerDistance(150 - K - 10)
        _pooled_insert(lambda b, rc, first, n: cl.addBatch(b, 150, reverseComplement=rc, storeReadPairedKmers=True, first=first, n=n),
                       batches, PAIRS)
        for w in FILTERS:
            assert cl.popcount(w) == pop[w], "sharded engine (native driver: %s): popcount of filter %d" % (native, w)
            assert cl.fold(w) == dig[w], "sharded engine (native driver: %s) differs in filter %d" % (native, w)
        # the whole index range is in use: every eighth of a filter holds its share of the entries (an index computed in 37
        # bits, or a span applied to the wrong filter, leaves the upper ranks empty)
        for w in FILTERS:
            local = [r.local_popcount(w) for r in cl.ranks]
            assert min(local) > 0.9 * pop[w] / 8 and max(local) < 1.1 * pop[w] / 8, (w, local)
        # queries on the sharded graph find every k-mer (38-bit indices through the query exchange)
        h0 = batches[1].nthash(K, 1, first=0, n=4_000)
        per_rank = [h0[r::8] for r in range(8)]
        assert all(bool(np.all(x)) for x in cl.contains(per_rank))
        assert all(float(x.min()) >= 1.0 for x in cl.getCount(per_rank))
        cl.destroy()
