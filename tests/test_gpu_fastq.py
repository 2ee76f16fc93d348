"""FASTQ text parsed on the GPU (rb_batch_create_fastq / rb_graph_add_fastq) against the host-side splitter, which
tests/test_io_formats.py checks against a line reader written like FastqReader (R/io/FastqReader.java:140-186)."""
import os

import numpy as np
import pytest

from test_io_formats import make_fastq

pytestmark = pytest.mark.gpu


def _same_batch(a, b):
    assert a.n_reads == b.n_reads and a.info()["n_bases"] == b.info()["n_bases"]
    sa, oa = a.download(); sb, ob = b.download()
    assert (oa == ob).all() and (sa == sb).all()
    if a.n_reads:
        assert (a.nthash(7, 1) == b.nthash(7, 1)).all()


@pytest.mark.parametrize("eol", [b"\n", b"\r\n", b"\r"])
def test_gpu_fastq_records_match_the_host_splitter(eol):
    from rnabloom import io as RIO, _native as N
    from rnabloom.graph import ReadBatch
    for n, tail in ((0, True), (1, True), (1, False), (7, True), (250, False), (3000, True)):
        text = make_fastq(n, 5 + n, eol, tail)
        for cut in (0, 1, 17, 40):                                # a truncated last record is dropped
            t = text[:len(text) - cut] if cut and len(text) > cut else text
            for use_qual in (True, False):
                try:
                    seq, qual, off = RIO.splitFastq(t, 4, with_qual=use_qual)
                except N.NativeError as e:                         # cut inside the last quality line
                    assert "bases and" in str(e) and use_qual
                    with pytest.raises(N.NativeError, match="bases and"):
                        RIO.batchFromFastq(t, with_qual=True)
                    continue
                want = ReadBatch.from_ascii(seq, qual, off, 3)
                got, used = RIO.batchFromFastq(t, minBaseQual=3, with_qual=use_qual)
                _same_batch(got, want)
                assert used <= len(t)
                got.close(); want.close()


def test_gpu_fastq_pieces_resume_at_record_boundaries():
    from rnabloom import io as RIO
    for eol in (b"\n", b"\r\n", b"\r"):
        text = make_fastq(400, 77, eol, True)
        whole, used = RIO.batchFromFastq(text)
        assert used == len(text)
        sw, ow = whole.download()
        for cut in (len(text) // 3, len(text) // 2 + 1, len(text) - 2, 8192, 8193, 16384 + 5):
            cut = min(cut, len(text) - 1)
            first, u1 = RIO.batchFromFastq(text[:cut], final=False)
            assert 0 < u1 <= cut
            rest, u2 = RIO.batchFromFastq(text[u1:], final=True)
            assert u1 + u2 == len(text) and first.n_reads + rest.n_reads == whole.n_reads
            s1, o1 = first.download(); s2, o2 = rest.download()
            assert (np.concatenate([s1, s2]) == sw).all() and (np.concatenate([o1, o2[1:] + o1[-1]]) == ow).all()
            first.close(); rest.close()
        whole.close()


def test_gpu_fastq_errors_are_the_reference_messages():
    from rnabloom import io as RIO, _native as N
    good = make_fastq(50, 3)
    lines = good.split(b"\n")
    bad1 = b"\n".join([b"read0"] + lines[1:])
    bad3 = b"\n".join(lines[:6] + [b"-"] + lines[7:])
    with pytest.raises(N.NativeError, match="Line 1 of FASTQ record is expected to start with '@'"):
        RIO.batchFromFastq(bad1)
    with pytest.raises(N.NativeError, match=r"Line 3 of FASTQ record is expected to start with '\+'"):
        RIO.batchFromFastq(bad3)
    short = b"\n".join(lines[:3] + [lines[3][:-1]] + lines[4:])
    with pytest.raises(N.NativeError, match="record 0 has different numbers of bases and qualities"):
        RIO.batchFromFastq(short)
    b, used = RIO.batchFromFastq(short, with_qual=False)          # FASTA-like use: qualities ignored
    assert b.n_reads == 50
    b.close()


def test_add_fastq_equals_add_reads_of_the_split_text(monkeypatch):
    from rnabloom import io as RIO, _native as N
    from rnabloom.graph import BloomFilterDeBruijnGraph
    rng = np.random.default_rng(9)
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), 30000)
    recs = []
    for i in range(6000):
        p = int(rng.integers(0, genome.size - 150)); L = int(rng.integers(30, 151))
        sq = genome[p:p + L].copy()
        sq[rng.random(L) < 0.01] = ord("N")
        ql = np.full(L, ord("I"), np.uint8); ql[rng.random(L) < 0.02] = ord("#")
        recs.append(b"@r%d\n" % i + sq.tobytes() + b"\n+\n" + ql.tobytes() + b"\n")
    text = b"".join(recs)
    sz = N.lib.rb_expected_size(60000, 0.01, 2)
    def fresh():
        g = BloomFilterDeBruijnGraph(sz, sz, sz, 2, 2, 2, 25, False, True, rngSeed=4)
        g.setReadPairedKmerDistance(40)
        return g
    want = fresh()
    seq, qual, off = RIO.splitFastq(text)
    st0 = want.addReads(seq, qual, off, 3, storeReadPairedKmers=True)
    for piece in (None, "20000", "700"):                          # one piece, many, pieces of a few records
        if piece: monkeypatch.setenv("RB_FASTQ_PIECE", piece)
        g = fresh()
        st, n = g.addFastq(text, 3, storeReadPairedKmers=True)
        assert n == 6000 and (st.kmers, st.pairs, st.reads) == (st0.kmers, st0.pairs, st0.reads)
        for which in (N.DBGBF, N.CBF, N.RPKBF):
            assert (g.exportFilter(which) == want.exportFilter(which)).all()
        g.destroy()
    # ... and the same text from a FILE, streamed (rb_graph_add_fastq_file: read / inflate the next piece while this one is inserted):
    # plain, one gzip member, several members followed by bytes that are no gzip header (GZIPInputStream reads every member and takes
    # such bytes for the end of the stream), pieces of every size
    import gzip, tempfile
    monkeypatch.delenv("RB_FASTQ_PIECE", raising=False)
    with tempfile.TemporaryDirectory() as d:
        plain = os.path.join(d, "r.fq"); open(plain, "wb").write(text)
        one = os.path.join(d, "r.fq.gz"); open(one, "wb").write(gzip.compress(text, 1))
        cut = [0, 1000, len(text) // 3, len(text) // 3 + 1, len(text)]
        many = os.path.join(d, "m.fq.gz")
        open(many, "wb").write(b"".join(gzip.compress(text[a:b], 1) for a, b in zip(cut, cut[1:])) + b"\0" * 9 + b"trailing bytes that are no gzip header end the stream")
        for path in (plain, one, many):
            for piece in (None, "150000", "900"):
                if piece: monkeypatch.setenv("RB_FASTQ_PIECE", piece)
                else: monkeypatch.delenv("RB_FASTQ_PIECE", raising=False)
                g = fresh()
                st, n = g.addFastqFile(path, 3, storeReadPairedKmers=True)
                assert n == 6000 and (st.kmers, st.pairs, st.reads) == (st0.kmers, st0.pairs, st0.reads), (path, piece)
                for which in (N.DBGBF, N.CBF, N.RPKBF):
                    assert (g.exportFilter(which) == want.exportFilter(which)).all(), (path, piece, which)
                g.destroy()
        monkeypatch.delenv("RB_FASTQ_PIECE", raising=False)
        bad = os.path.join(d, "bad.gz"); open(bad, "wb").write(gzip.compress(text, 1)[:-200])
        g = fresh()
        with pytest.raises(Exception, match="gzip"):
            g.addFastqFile(bad, 3)
        with pytest.raises(Exception, match="cannot open"):
            g.addFastqFile(os.path.join(d, "missing.fq"), 3)
        g.destroy()
        # FASTA file, streamed
        fa = b"".join(b">s%d\n" % i + r.split(b"\n")[1] + b"\n" for i, r in enumerate(recs))
        fpath = os.path.join(d, "t.fa.gz"); open(fpath, "wb").write(gzip.compress(fa, 1))
        ga, gb = fresh(), fresh()
        sa, na = ga.addFasta(fa, storeReadPairedKmers=True)
        monkeypatch.setenv("RB_FASTQ_PIECE", "5000")
        sb, nb = gb.addFastaFile(fpath, storeReadPairedKmers=True)
        monkeypatch.delenv("RB_FASTQ_PIECE", raising=False)
        assert na == nb == 6000 and (sa.kmers, sa.pairs) == (sb.kmers, sb.pairs)
        for which in (N.DBGBF, N.CBF, N.RPKBF):
            assert (ga.exportFilter(which) == gb.exportFilter(which)).all()
        ga.destroy(); gb.destroy()
    want.destroy()


# ---- FASTA on the GPU (rb_batch_create_fasta / rb_graph_add_fasta) against FastaReader.next restated over a line reader ----
def _fasta_ref(text):
    """BufferedReader.lines() + FastaReader.next() (R/io/FastaReader.java:70-104) -> (sequences, error)"""
    s = text.decode("latin1")
    lines, cur, i = [], [], 0
    while i < len(s):
        c = s[i]
        if c == "\n" or c == "\r":
            lines.append("".join(cur)); cur = []
            if c == "\r" and i + 1 < len(s) and s[i + 1] == "\n": i += 1
        else:
            cur.append(c)
        i += 1
    if cur: lines.append("".join(cur))
    trim = lambda x: x.strip("".join(chr(c) for c in range(0x21)))          # String.trim(): characters <= ' '
    out, pos, header = [], 0, None
    while True:
        if pos >= len(lines): break                                            # !itr.hasNext() -> null
        if header is None:
            header = trim(lines[pos]); pos += 1
        if header == "": break                                                 # null: the iteration ends
        if header[0] != ">": return out, "Incorrect FASTA header format"
        seq = []
        while pos < len(lines):
            line = trim(lines[pos]); pos += 1
            if line == "": header = None; break
            if line[0] == ">": header = line; break
            seq.append(line)
        else:
            header = None if header is None else header
            out.append("".join(seq))
            # the loop ran out of lines: a later call sees !hasNext() and returns null whatever header is pending
            break
        out.append("".join(seq))
    return out, None


def _norm(seq):
    t = {ord(a): ord(b) for a, b in zip("acgtuU", "ACGTTT")}
    return "".join(ch if ch in "ACGT" else "N" for ch in seq.translate(t))


def _make_fasta(n, seed, eol=b"\n", wrap=60, blanks=True):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        L = int(rng.integers(0, 400))
        sq = bytes(rng.choice(np.frombuffer(b"ACGTNacgtu", np.uint8), L).tolist())
        out.append(b">r%d some comment" % i + (b"  " if i % 5 == 0 else b"") + eol)
        w = wrap if i % 3 else max(1, L)
        for a in range(0, L, w):
            out.append((b" \t" if i % 7 == 0 else b"") + sq[a:a + w] + (b" " if i % 4 == 0 else b"") + eol)
        if blanks and i % 11 == 3: out.append(eol)                            # an empty line closes the record; a header follows
    return b"".join(out)


@pytest.mark.parametrize("eol", [b"\n", b"\r\n", b"\r"])
def test_gpu_fasta_records_match_the_reader(eol):
    from rnabloom import io as RIO
    cases = [_make_fasta(n, 3 + n, eol) for n in (0, 1, 2, 40, 700)]
    cases.append(_make_fasta(50, 8, eol) + eol + eol + _make_fasta(5, 9, eol))      # two empty lines: the iteration ends there
    cases.append(_make_fasta(30, 10, eol) + b">last_header_without_sequence")       # never handed out
    cases.append(_make_fasta(30, 11, eol) + b">hdr" + eol + b"ACGT")               # open last line
    cases.append(eol + _make_fasta(3, 12, eol))                                     # empty first line: nothing
    cases.append(b">a" + eol + eol + b">b" + eol + b"AC" + eol)                     # empty record, then one more
    for text in cases:
        want, err = _fasta_ref(text)
        assert err is None
        b, used, ended = RIO.batchFromFasta(text)
        assert b.n_reads == len(want), (b.n_reads, len(want))
        s, off = b.download()
        got = [bytes(s[off[i]:off[i + 1]]).decode() for i in range(b.n_reads)]
        assert got == [_norm(x) for x in want]
        assert used == len(text)
        b.close()


def test_gpu_fasta_errors_pieces_and_insert(monkeypatch):
    from rnabloom import io as RIO, _native as N
    from rnabloom.graph import BloomFilterDeBruijnGraph
    with pytest.raises(N.NativeError, match="Incorrect FASTA header format"):
        RIO.batchFromFasta(b"ACGT\n>r\nACGT\n")
    with pytest.raises(N.NativeError, match="Incorrect FASTA header format"):
        RIO.batchFromFasta(b">r\nACGT\n\nACGT\n>s\nAC\n")                       # a sequence line where a header must stand
    b, _, ended = RIO.batchFromFasta(b">r\nACGT\n\n\nACGT\n")                     # ... but not behind the end of the iteration
    assert b.n_reads == 1 and ended
    b.close()
    # pieces: the last record of a piece that is not final stays unread
    text = _make_fasta(300, 21, b"\n", blanks=False)
    whole, _, _ = RIO.batchFromFasta(text)
    sw, ow = whole.download()
    for cut in (len(text) // 3, len(text) // 2 + 7, len(text) - 3):
        first, u1, _ = RIO.batchFromFasta(text[:cut], final=False)
        assert 0 < u1 <= cut and text[u1:u1 + 1] == b">"
        rest, u2, _ = RIO.batchFromFasta(text[u1:], final=True)
        s1, o1 = first.download(); s2, o2 = rest.download()
        assert first.n_reads + rest.n_reads == whole.n_reads
        assert (np.concatenate([s1, s2]) == sw).all() and (np.concatenate([o1, o2[1:] + o1[-1]]) == ow).all()
        first.close(); rest.close()
    whole.close()
    # FastaToGraphWorker: the same filters as addReads of the reader's sequences, in one piece and in many
    rng = np.random.default_rng(2)
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), 20000)
    recs, seqs = [], []
    for i in range(900):
        p = int(rng.integers(0, genome.size - 2500)); L = int(rng.integers(200, 2500))
        sq = genome[p:p + L].copy(); sq[rng.random(L) < 0.02] = ord("N")
        seqs.append(sq.tobytes())
        recs.append(b">read%d\n" % i + b"\n".join(sq.tobytes()[a:a + 70] for a in range(0, L, 70)) + b"\n")
    text = b"".join(recs)
    sz = N.lib.rb_expected_size(40000, 0.01, 2)
    want = BloomFilterDeBruijnGraph(sz, sz, sz, 2, 2, 2, 31, False, False, rngSeed=4)
    off = np.zeros(len(seqs) + 1, np.int64); np.cumsum([len(x) for x in seqs], out=off[1:])
    st0 = want.addReads(np.frombuffer(b"".join(seqs), np.uint8), None, off, 3)
    for piece in (None, "100000", "9000"):
        if piece: monkeypatch.setenv("RB_FASTQ_PIECE", piece)
        g = BloomFilterDeBruijnGraph(sz, sz, sz, 2, 2, 2, 31, False, False, rngSeed=4)
        st, n = g.addFasta(text)
        assert n == 900 and st.kmers == st0.kmers
        for which in (N.DBGBF, N.CBF):
            assert (g.exportFilter(which) == want.exportFilter(which)).all()
        g.destroy()
    want.destroy()


def test_gpu_fastq_text_just_below_4_gib():
    """text sizes in (0xFFFFE000, 0xFFFFFF00): the tile count used to be computed in 32 bits and wrapped to ~0 there
    (rb_io.hip fastq_batch_create: too small a device buffer, an underflowed memset length).  One 4 GiB - 4 KiB text of
    identical records, a record cut at the end."""
    from rnabloom import io as RIO
    rec = b"@r\n" + b"ACGTTGCAAGGCTTAC" * 16 + b"\n+\n" + b"I" * 256 + b"\n"          # 520 bytes, 256 bases
    n = 0xFFFFF000
    reps = n // len(rec) + 1
    text = np.tile(np.frombuffer(rec, np.uint8), reps)[:n]
    complete = n // len(rec)
    b, used = RIO.batchFromFastq(text, final=False)
    assert used == complete * len(rec)
    info = b.info()
    assert info["n_reads"] == complete and info["n_bases"] == complete * 256
    seq, off = b.download(complete - 3, 3)
    assert bytes(seq) == (b"ACGTTGCAAGGCTTAC" * 16) * 3 and list(off) == [0, 256, 512, 768]
    b.close()
    with pytest.raises(Exception, match="at most 4 GiB"):
        RIO.batchFromFastq(np.zeros(0xFFFFFF00, np.uint8))
