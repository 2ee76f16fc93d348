"""Read batches packed in HOST memory (include/rb_capi.h rb_batch_download_packed / rb_packed_stream_* / rb_graph_add_packed: the input format
SURVEY.md §8(d) quotes the metric on): a packed round trip reproduces the batch word for word, an insert streamed from host memory in
chunks leaves the filters the oracle leaves, and a chunk whose lengths do not describe its words is refused."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import rbo
from rnabloom import _native as N
from rnabloom import synth
from rnabloom.graph import BloomFilterDeBruijnGraph, PackedHost, PackedStream, ReadBatch


def ragged_reads(n, seed):
    """reads of 0 .. 400 bases with N bases and low qualities sprinkled in"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 400, n)
    lens[rng.integers(0, n, n // 50)] = 0
    lens[:7] = (0, 1, 31, 32, 33, 64, 65)
    reads, quals = [], []
    for L in lens.tolist():
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].copy()
        q = np.full(L, ord("I"), np.uint8)
        if L:
            s[rng.integers(0, L, max(1, L // 60))] = ord("N")
            q[rng.integers(0, L, max(1, L // 40))] = ord("#")
        reads.append(s.tobytes()); quals.append(q.tobytes())
    return reads, quals


@pytest.mark.parametrize("pinned", [True, False])
def test_packed_round_trip_reproduces_the_batch(pinned):
    reads, quals = ragged_reads(20_000, 3)
    b = ReadBatch.from_reads(reads, quals, 3)
    seq0, off0 = b.download()
    ph = b.downloadPacked(pinned=pinned)
    assert ph.n_reads == len(reads) and ph.n_words == int(((np.diff(off0) + 31) // 32).sum())
    assert (ph.len == np.diff(off0)).all()
    ps = PackedStream(8_000, int(ph.n_words))
    got_seq, got_len = [], []
    cuts = [0, 8_000, 8_001, 15_000, 20_000]                       # a chunk of one read among them; two buffers take turns
    ps.begin(ph, cuts[0], cuts[1] - cuts[0])
    for i in range(len(cuts) - 1):
        c = ps.finish()
        if i + 2 < len(cuts):
            ps.begin(ph, cuts[i + 1], cuts[i + 2] - cuts[i + 1])
        info = c.info()
        assert info["n_reads"] == cuts[i + 1] - cuts[i] and info["n_bases"] == int(off0[cuts[i + 1]] - off0[cuts[i]])
        s, o = c.download()
        got_seq.append(s); got_len.append(np.diff(o))
        # the device-only columns were rebuilt on the GPU: hashing every window of the uploaded chunk equals hashing the same reads of the original
        assert (c.nthash(25, 1) == b.nthash(25, 1, cuts[i], cuts[i + 1] - cuts[i])).all()
    assert (np.concatenate(got_len) == np.diff(off0)).all() and (np.concatenate(got_seq) == seq0).all()
    ps.close(); ph.close()


def test_an_insert_streamed_from_packed_host_memory_leaves_the_oracles_filters():
    d = synth.generate_pairs(6000, G=60000, err=0.002, n_rate=1e-3, seed=4)
    k, dist = 25, 150 - 25 - 10
    sizes = (2_400_011, 19_200_013, 2_400_011)
    og = rbo.Graph(*sizes, 2, 2, 2, k, False, True, 7)
    gg = BloomFilterDeBruijnGraph(*sizes, 2, 2, 2, k, False, True, rngSeed=7)
    og.set_read_pair_distance(dist); gg.setReadPairedKmerDistance(dist)
    kmers = 0
    for name, rc in (("left", False), ("right", True)):
        seq, off = synth.flat(d[name])
        qual, _ = synth.flat(d[name[0] + "qual"])
        ost = og.add_reads(seq, qual, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
        b = ReadBatch.from_ascii(seq, qual, off, 3)
        ph = b.downloadPacked()
        st = gg.addPacked(ph, reverseComplement=rc, storeReadPairedKmers=True, pieceReads=1700)       # four pieces, the last one short
        assert st.kmers == ost.kmers and st.pairs == ost.pairs and st.reads == len(d[name])
        kmers += st.kmers
        ph.close(); b.close()
    assert kmers > 0
    assert (gg.exportFilter(N.DBGBF) == og.dbgbf_bytes()).all()
    assert (gg.exportFilter(N.CBF) == og.cbf_bytes()).all()
    assert (gg.exportFilter(N.RPKBF) == og.rpkbf_bytes()).all()


def test_ragged_reads_streamed_in_small_pieces_through_several_sub_batches():
    """reads of 0 .. 400 bases (empty ones, one-word ones), pieces of 900 reads, sub-batches of at most 20 000 records: every sub-batch waits for the
    pieces its words lie in and nothing else — the filters are those of the same reads inserted from a resident batch"""
    reads, quals = ragged_reads(12_000, 11)
    b = ReadBatch.from_reads(reads, quals, 3)
    ph = b.downloadPacked()
    digests = []
    for how in ("resident", "packed", "packed-one-piece"):
        g = BloomFilterDeBruijnGraph(3_000_017, 3_000_017, 0, 2, 2, 1, 25, False, False, rngSeed=3, maxBatchKmers=20_000)
        for rep in range(2):                                          # a second pass: every k-mer re-sighted
            if how == "packed" and rep == 1:
                g.prefetchPacked(ph, pieceReads=900)                  # started ahead: the add below picks the upload up where it is
            st = g.addBatch(b) if how == "resident" else g.addPacked(ph, pieceReads=900 if how == "packed" else 50_000)
            assert st.reads == 12_000
        digests.append((st.kmers, g.popcount(N.DBGBF), g.fold(N.DBGBF), g.fold(N.CBF)))
        g.destroy()
    assert digests[0] == digests[1] == digests[2], digests
    ph.close(); b.close()


def test_a_chunk_whose_lengths_do_not_describe_its_words_is_refused_and_the_stream_lives_on():
    reads, quals = ragged_reads(3000, 5)
    b = ReadBatch.from_reads(reads, quals, 3)
    ph = b.downloadPacked(pinned=False)
    ps = PackedStream(3000, int(ph.n_words))
    N.check(N.lib.rb_packed_stream_begin(ps.h, ph.codes.ctypes.data, ph.valid.ctypes.data, ph.len.ctypes.data, 3000, ph.n_words - 1))
    out = C.c_void_p()
    assert N.lib.rb_packed_stream_finish(ps.h, C.byref(out)) != 0 and b"add up to" in N.lib.rb_last_error()
    assert N.lib.rb_packed_stream_finish(ps.h, C.byref(out)) != 0          # nothing in flight any more
    ps.begin(ph)
    with pytest.raises(N.NativeError):
        ps.begin(ph)                                                        # one chunk in flight at a time
    c = ps.finish()
    assert (c.download()[0] == b.download()[0]).all()
    g = BloomFilterDeBruijnGraph(100_003, 100_003, 100_003, 2, 2, 2, 25, False, False)
    with pytest.raises(N.NativeError):
        N.check(N.lib.rb_graph_add_packed(g.h, ph.codes.ctypes.data, ph.valid.ctypes.data, ph.len.ctypes.data, 3000, ph.n_words - 5, 0, 0, None))
    ps.close()


@pytest.mark.parametrize("piece", ["401", "5000", "1000000000"])
def test_host_ascii_reads_streamed_in_pieces_leave_the_oracles_filters(monkeypatch, piece):
    """rb_graph_add_reads over more than one piece (csrc/rb_packed.hip add_reads_streamed; RB_ASCII_PIECE sets the piece): lengths and word offsets
    computed on the GPU from the caller's offsets, bases + qualities through two staging buffers and the encode kernel piece by piece, one insert over
    the batch that is still arriving — ragged reads (empty ones, one-word ones, a piece of one read), N bases, low qualities, sub-batches of at most
    20 000 records, two passes; with a piece larger than the input the chunk-by-chunk path runs: all three leave what the oracle leaves"""
    monkeypatch.setenv("RB_ASCII_PIECE", piece)
    reads, quals = ragged_reads(9_000, 21)
    seq = np.frombuffer(b"".join(reads), np.uint8); qual = np.frombuffer(b"".join(quals), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    sizes = (2_000_003, 2_000_003, 300_007)
    og = rbo.Graph(*sizes, 2, 2, 2, 25, False, True, 5)
    gg = BloomFilterDeBruijnGraph(*sizes, 2, 2, 2, 25, False, True, rngSeed=5, maxBatchKmers=20_000)
    og.set_read_pair_distance(60); gg.setReadPairedKmerDistance(60)
    for rep in range(2):
        ost = og.add_reads(seq, qual, off, 3, rbo.STORE_READ_PAIRS)
        st = gg.addReads(seq, qual, off, 3, storeReadPairedKmers=True)
        assert (st.kmers, st.pairs, st.reads) == (ost.kmers, ost.pairs, len(reads))
    assert (gg.exportFilter(N.DBGBF) == og.dbgbf_bytes()).all()
    assert (gg.exportFilter(N.CBF) == og.cbf_bytes()).all()
    assert (gg.exportFilter(N.RPKBF) == og.rpkbf_bytes()).all()
    # without qualities, and a bad offsets array: refused, the handle lives on
    st2 = gg.addReads(seq, None, off, 0)
    ost2 = og.add_reads(seq, None, off, 0, 0)
    assert st2.kmers == ost2.kmers and (gg.exportFilter(N.CBF) == og.cbf_bytes()).all()
    bad = off.copy(); bad[4000] = bad[3999] - 5
    with pytest.raises(N.NativeError):
        gg.addReads(seq, qual, bad, 3)
    assert gg.addReads(seq, qual, off, 3).reads == len(reads)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_streamed_ascii_ingest_with_random_pieces_and_sub_batches(monkeypatch, seed):
    """the same comparison over random shapes: piece size, sub-batch bound, read-length mix, stranded or not, with and without qualities and
    read pairs — the piece boundaries fall anywhere relative to the sub-batches' and the reads' words"""
    rng = np.random.default_rng(1000 + seed)
    monkeypatch.setenv("RB_ASCII_PIECE", str(int(rng.integers(64, 60_000))))
    n = int(rng.integers(500, 7000))
    reads, quals = ragged_reads(n, 100 + seed)
    if seed % 2:                                                   # a block of equal-length reads in the middle: the read-per-lane walkers
        k0 = n // 3
        for i in range(k0, 2 * k0):
            reads[i] = (reads[i] * 3 + b"ACGTACGTAC" * 20)[:150]; quals[i] = b"I" * 150
    seq = np.frombuffer(b"".join(reads), np.uint8); qual = np.frombuffer(b"".join(quals), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    stranded, pairs, with_q = bool(seed & 2), bool(seed % 3), seed != 4
    k = [25, 31, 36][seed % 3]
    sizes = (1_000_003, 1_500_007, 200_003)
    og = rbo.Graph(*sizes, 2, 2, 2, k, stranded, pairs, seed)
    gg = BloomFilterDeBruijnGraph(*sizes, 2, 2, 2, k, stranded, pairs, rngSeed=seed, maxBatchKmers=int(rng.integers(3_000, 200_000)))
    if pairs:
        og.set_read_pair_distance(40); gg.setReadPairedKmerDistance(40)
    for rc in (False, True):
        fl = (rbo.STORE_READ_PAIRS if pairs else 0) | (rbo.REVCOMP if rc else 0)
        ost = og.add_reads(seq, qual if with_q else None, off, 3 if with_q else 0, fl)
        st = gg.addReads(seq, qual if with_q else None, off, 3 if with_q else 0, reverseComplement=rc, storeReadPairedKmers=pairs)
        assert (st.kmers, st.pairs) == (ost.kmers, ost.pairs)
    assert (gg.exportFilter(N.DBGBF) == og.dbgbf_bytes()).all() and (gg.exportFilter(N.CBF) == og.cbf_bytes()).all()
    if pairs:
        assert (gg.exportFilter(N.RPKBF) == og.rpkbf_bytes()).all()
