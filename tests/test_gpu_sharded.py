"""Sharded (multi-GPU) engine on ONE GPU: G virtual ranks in one process (LoopbackCluster) must
reproduce the sequential oracle bit for bit, for every G."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import rbo
from rnabloom import _native as N
from rnabloom import synth
from rnabloom.graph import ReadBatch
from rnabloom.sharded import LoopbackCluster


def deal(seq2d, qual2d, G, rps):
    """global order = sub-batch by sub-batch, rank by rank, `rps` reads per rank per sub-batch"""
    n = seq2d.shape[0]
    owner = (np.arange(n) // rps) % G
    order = []
    blocks = -(-n // rps)
    for t in range(0, blocks, G):
        for r in range(G):
            b = t + r
            order.extend(range(b * rps, min(n, (b + 1) * rps)))
    per_rank = [np.nonzero(owner == r)[0] for r in range(G)]
    return per_rank, np.asarray(order)


@pytest.mark.parametrize("G", [1, 2, 4, 8])
@pytest.mark.parametrize("sizes", [(400_003, 3_000_017, 90_001), (70_001, 250_007, 9_001)])
def test_loopback_matches_oracle(G, sizes):
    d = synth.generate_pairs(2400, G=25000, err=0.003, n_rate=1e-3, seed=17 + G)
    rps = 160
    for name, rc in (("left", False),):
        pass
    og = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 9)
    cl = LoopbackCluster(G, *sizes, 2, 2, 2, 25, False, True, rngSeed=9)
    og.set_read_pair_distance(115); cl.setReadPairedKmerDistance(115)
    for name, rc in (("left", False), ("right", True)):
        seq2d, qual2d = d[name], d[name[0] + "qual"]
        per_rank, order = deal(seq2d, qual2d, G, rps)
        # oracle consumes the reads in the global order the sharded engine defines
        s, off = synth.flat(seq2d[order]); q, _ = synth.flat(qual2d[order])
        og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
        batches = []
        for r in range(G):
            rs, roff = synth.flat(seq2d[per_rank[r]]); rq, _ = synth.flat(qual2d[per_rank[r]])
            batches.append(ReadBatch.from_ascii(rs, rq, roff, 3))
        cl.addBatches(batches, 150, reverseComplement=rc, storeReadPairedKmers=True, reads_per_substep=rps)
        assert (cl.exportFilter(N.DBGBF) == og.dbgbf_bytes()).all(), "dbgbf differs"
        assert (cl.exportFilter(N.RPKBF) == og.rpkbf_bytes()).all(), "rpkbf differs"
        cg, co = cl.exportFilter(N.CBF), og.cbf_bytes()
        bad = np.nonzero(cg != co)[0]
        assert bad.size == 0, "cbf differs at %d bytes: %s gpu %s oracle %s" % (bad.size, bad[:6], cg[bad[:6]], co[bad[:6]])
    pc = og.popcounts()
    assert (cl.popcount(N.DBGBF), cl.popcount(N.CBF), cl.popcount(N.RPKBF)) == pc
    assert sum(r.stats["conflict_ops"] for r in cl.ranks) > 0
    cl.destroy()


def test_loopback_high_multiplicity():
    d = synth.generate_pairs(5000, G=3000, err=0.001, n_rate=1e-3, seed=3, uniform_expr=True)
    sizes = (100_003, 150_001, 20_011)
    og = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 1)
    cl = LoopbackCluster(4, *sizes, 2, 2, 2, 25, False, True, rngSeed=1)
    per_rank, order = deal(d["left"], d["lqual"], 4, 500)
    s, off = synth.flat(d["left"][order]); q, _ = synth.flat(d["lqual"][order])
    og.add_reads(s, q, off, 3, 0)
    batches = []
    for r in range(4):
        rs, roff = synth.flat(d["left"][per_rank[r]]); rq, _ = synth.flat(d["lqual"][per_rank[r]])
        batches.append(ReadBatch.from_ascii(rs, rq, roff, 3))
    cl.addBatches(batches, 150, reads_per_substep=500)
    assert (cl.exportFilter(N.DBGBF) == og.dbgbf_bytes()).all()
    assert (cl.exportFilter(N.CBF) == og.cbf_bytes()).all()
    assert og.cbf_bytes().max() > 24
    cl.destroy()
