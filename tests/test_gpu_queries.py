"""The lookup leg on reads that are resident in HBM: rb_graph_batch_counts (the count profile of getKmers without the hashes)
against the CPU oracle's getKmers and against rb_graph_kmers, for uniform and ragged batches, canonical and stranded graphs,
both output layouts, host and device outputs, 2 and 3 hash functions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import rbo
from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch
from test_gpu_parity import graph_pair, make_reads, ragged_reads


def expected_rows(og, reads, k):
    rows = []
    for s in reads:
        _, _, ec = og.get_kmers(s)
        assert len(ec) == max(0, len(s) - k + 1)
        rows.append(np.asarray(ec, np.float32))
    return rows


@pytest.mark.parametrize("stranded", [False, True])
@pytest.mark.parametrize("hashes,piece", [(2, None), (3, None), (2, "1000"), (2, "1")])
def test_batch_counts_match_oracle_and_get_kmers(monkeypatch, stranded, hashes, piece):
    if piece:                                   # the host copy in many small pieces through the two device buffers
        monkeypatch.setenv("RB_QUERY_PIECE", piece)
    (ls, lq, off), _ = make_reads(1500, 12000, 0.002, 1e-3, seed=33)
    og, gg = graph_pair(300_007, 2_000_003, 10_007, stranded=stranded, pairs=False, dbg_h=hashes, cbf_h=hashes)
    og.add_reads(ls, lq, off, 3, 0); gg.addReads(ls, lq, off, 3)
    # uniform reads (the packed layout) ...
    reads = [bytes(ls[off[i]:off[i + 1]]) for i in range(300)]
    b = ReadBatch.from_reads(reads, None)
    rows = expected_rows(og, reads, 25)
    got = gg.batchCounts(b)
    stride = len(reads[0]) - 24
    assert got.size == len(reads) * stride
    assert (got.reshape(len(reads), stride) == np.stack(rows)).all()
    ko, _, _, c = gg.getKmers(reads)
    assert (got == c).all()
    # ... a sub-range of the batch, host and device outputs
    sub = gg.batchCounts(b, 17, 100)
    assert (sub == got[17 * stride:117 * stride]).all()
    dev = gg.batchCounts(b, 17, 100, to_host=False)
    assert (dev.cpu().numpy() == sub).all()
    # ... and ragged ones with unusable bases, reads shorter than k, empty reads: packed rows by koffsets, padded rows without
    rag, _ = ragged_reads(5, 300)
    rag += [reads[0][:40] + b"N" + reads[1][:80], b"", reads[2]]
    rb = ReadBatch.from_reads(rag, None)
    rrows = expected_rows(og, rag, 25)
    ko, _, _, c = gg.getKmers(rag)
    packed = gg.batchCounts(rb, koffsets=ko)
    assert packed.size == ko[-1] and (packed == c).all()
    assert (packed == np.concatenate(rrows)).all()
    padded = gg.batchCounts(rb)
    stride = max(len(s) for s in rag) - 24
    padded = padded.reshape(len(rag), stride)
    for i, r in enumerate(rrows):
        assert (padded[i, :len(r)] == r).all() and (padded[i, len(r):] == 0).all()
    part = gg.batchCounts(rb, 100, 150, koffsets=ko[100:251] - ko[100])
    assert (part == c[ko[100]:ko[250]]).all()


def test_batch_counts_argument_errors():
    _, gg = graph_pair(100_003, 100_003, 10_007, pairs=False)
    b = ReadBatch.from_reads([b"ACGT" * 20] * 4, None)
    with pytest.raises(RuntimeError, match="read range outside the batch"):
        gg.batchCounts(b, 2, 5)
    assert gg.batchCounts(b, 0, 0).size == 0
    with pytest.raises(RuntimeError, match="koffsets"):
        gg.batchCounts(b, 0, 2, koffsets=np.array([5, 61, 117]))
