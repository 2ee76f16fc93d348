"""BASELINE config 2 at full size (50 M read pairs x 2 x 150 bp, k = 25, nk = 450 M, FPR 0.01: 12.26 G k-mers, 0.94 G
paired k-mers) — the CPU oracle would need minutes per file for it, so the full-size run is checked through properties
that hold at any size and that a wrong engine breaks:

  * sub-batch invariance: the engine is order-exact, so where the input is cut into sub-batches (2^30 or 2^28 records,
    with or without the cold-start ramp) cannot change a single bit or counter;
  * engine invariance: the sharded engine (8 virtual ranks: routed probes, conflict components, replicated cache) must
    produce the same three filters as the single-GPU engine;
  * idempotence of the Bloom bit sets: adding the same reads again leaves dbgbf and rpkbf unchanged (the counting
    filter moves on, and only upwards);
  * no false negatives: every k-mer of sampled reads is contained, with graph count >= 1;
  * occupancy: the measured false-positive rates stay below the configured one (the filters are sized for nk = 450 M
    and the set holds ~400 M distinct k-mers).

The small-size parity tests (tests/test_gpu_parity.py, tests/test_gpu_sharded.py) pin the same engines to the oracle
bit for bit; this file carries that to the size the benchmark is quoted on."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu

PAIRS, GENOME, NK, K, FPR = 50_000_000, 64_000_000, 450_000_000, 25, 0.01


def _insert(g, batch, n_pairs):
    """the reference's stage 1 over the two files of a paired-end library (R/RNABloom.java:1080-1215): left reads
    forward, right reads reverse-complemented, read-paired k-mers stored"""
    s1 = g.addBatch(batch, reverseComplement=False, storeReadPairedKmers=True, first=0, n=n_pairs)
    s2 = g.addBatch(batch, reverseComplement=True, storeReadPairedKmers=True, first=n_pairs, n=n_pairs)
    return s1.kmers + s2.kmers, s1.pairs + s2.pairs


def test_config2_full_size_properties(monkeypatch):
    from rnabloom import _native as N
    from rnabloom import sharded
    from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch

    bits = N.lib.rb_expected_size(NK, FPR, 2)
    batch = ReadBatch.synthetic(PAIRS, GENOME, 150, 300, 30, 0.001, 1e-4, 2.0, seed=0x5EED, device=0)
    assert batch.n_reads == 2 * PAIRS

    def graph(max_batch):
        g = BloomFilterDeBruijnGraph(bits, bits, bits, 2, 2, 2, K, False, True, device=0, rngSeed=1, maxBatchKmers=max_batch)
        g.setReadPairedKmerDistance(150 - K - 10)
        return g

    ga = graph(0)                                   # default: 2^30 records per sub-batch
    kmers, pairs = _insert(ga, batch, PAIRS)
    assert kmers > 12_000_000_000 and pairs > 900_000_000
    ref = {w: ga.exportFilter(w) for w in (N.DBGBF, N.CBF, N.RPKBF)}
    pop = {w: ga.popcount(w) for w in (N.DBGBF, N.CBF, N.RPKBF)}

    # occupancy
    assert 0 < ga.getDbgbfFPR() < FPR and 0 < ga.getCbfFPR() < FPR and 0 < ga.getRpkbfFPR() < FPR

    # no false negatives on sampled reads (first, middle, last 20 000 reads of the set)
    for first in (0, PAIRS - 10_000, 2 * PAIRS - 20_000):
        h0 = batch.nthash(K, 1, first=first, n=20_000)
        assert h0.size > 2_000_000
        assert bool(np.all(ga.contains(h0)))
        assert int(ga.getCount(h0).min()) >= 1

    # sub-batch invariance: other cut points, no cold-start ramp
    monkeypatch.setenv("RB_NO_RAMP", "1")
    gb = graph(1 << 28)
    assert _insert(gb, batch, PAIRS) == (kmers, pairs)
    monkeypatch.delenv("RB_NO_RAMP")
    for w in ref:
        assert gb.popcount(w) == pop[w]
        assert np.array_equal(gb.exportFilter(w), ref[w]), "filter %d depends on the sub-batch size" % w
    gb.destroy()

    # engine invariance: 8 virtual ranks of the sharded engine (the rank count of BASELINE configs 3 and 4)
    cl = sharded.LoopbackCluster(8, bits, bits, bits, 2, 2, 2, K, False, True, device=0, rngSeed=1)
    cl.setReadPairedKmerDistance(150 - K - 10)
    cl.addBatch(batch, 150, reverseComplement=False, storeReadPairedKmers=True, first=0, n=PAIRS)
    cl.addBatch(batch, 150, reverseComplement=True, storeReadPairedKmers=True, first=PAIRS, n=PAIRS)
    for w in ref:
        assert cl.popcount(w) == pop[w]
        assert np.array_equal(cl.exportFilter(w), ref[w]), "sharded engine differs in filter %d" % w
    cl.destroy()

    # idempotence of the bit sets; counters only grow
    _insert(ga, batch, PAIRS)
    assert ga.popcount(N.DBGBF) == pop[N.DBGBF] and ga.popcount(N.RPKBF) == pop[N.RPKBF]
    assert np.array_equal(ga.exportFilter(N.DBGBF), ref[N.DBGBF])
    assert np.array_equal(ga.exportFilter(N.RPKBF), ref[N.RPKBF])
    cbf2 = ga.exportFilter(N.CBF)
    assert bool(np.all(cbf2 >= ref[N.CBF])) and int((cbf2 > ref[N.CBF]).sum()) > 100_000_000
    ga.destroy()
