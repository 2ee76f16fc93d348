"""FASTQ record splitting and .nbits encoding (host side of SURVEY §8 f2 / f3) against line-by-line Python restatements
of the reference readers.  CPU only: these two entry points never touch the device."""
import numpy as np
import pytest

from rnabloom import _native as N
from rnabloom import io as RIO


def ref_split(text):
    """BufferedReader.lines() + FastqReader.nextWithoutName (R/io/FastqReader.java:140-186)"""
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
    recs = []
    for r in range(len(lines) // 4):
        l1, sq, l3, ql = lines[4 * r:4 * r + 4]
        if not l1.startswith("@"): raise ValueError("Line 1")
        if not l3.startswith("+"): raise ValueError("Line 3")
        recs.append((sq, ql))
    return recs


def make_fastq(n, seed, eol=b"\n", tail=True):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        L = int(rng.integers(1, 160))
        sq = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), L).tolist())
        ql = bytes(rng.choice(np.frombuffer(b"@+#I5!~", np.uint8), L).tolist())      # qualities that look like headers
        out.append(b"@read%d/1 comment" % i + eol + sq + eol + b"+" + (b"read%d" % i if i % 3 == 0 else b"") + eol + ql + (eol if (tail or i < n - 1) else b""))
    return b"".join(out)


@pytest.mark.parametrize("eol", [b"\n", b"\r\n", b"\r"])
@pytest.mark.parametrize("threads", [1, 3, 8, 64])
def test_fastq_split_matches_line_reader(eol, threads):
    for n, tail in ((0, True), (1, True), (1, False), (7, True), (250, False), (250, True)):
        text = make_fastq(n, 11 + n, eol, tail)
        for cut in (0, 1, 17):                                   # a truncated last record is dropped
            t = text[:len(text) - cut] if cut else text
            want = ref_split(t)
            if any(len(a) != len(b) for a, b in want):           # cut inside the last quality line: the reader hands out a short
                with pytest.raises(N.NativeError, match="bases and"):   # string there; the splitter refuses it
                    RIO.splitFastq(t, threads)
                s2, _, o2 = RIO.splitFastq(t, threads, with_qual=False)
                assert [bytes(s2[o2[i]:o2[i + 1]]).decode("latin1") for i in range(o2.size - 1)] == [a for a, _ in want]
                continue
            seq, qual, off = RIO.splitFastq(t, threads)
            assert off.size - 1 == len(want)
            for i, (sq, ql) in enumerate(want):
                assert bytes(seq[off[i]:off[i + 1]]) == sq.encode("latin1") and bytes(qual[off[i]:off[i + 1]]) == ql.encode("latin1")
            s2, q2, o2 = RIO.splitFastq(t, threads, with_qual=False)
            assert q2 is None and (o2 == off).all() and (s2 == seq).all()


def test_fastq_split_errors_like_the_reference():
    good = make_fastq(5, 3)
    with pytest.raises(N.NativeError, match="Line 1 of FASTQ record is expected to start with '@'"):
        RIO.splitFastq(good.replace(b"@read2/1", b"read2/1"), 4)
    with pytest.raises(N.NativeError, match="Line 3 of FASTQ record is expected to start with '\\+'"):
        RIO.splitFastq(b"@a\nACGT\n-\nIIII\n", 2)
    with pytest.raises(N.NativeError, match="bases and"):
        RIO.splitFastq(b"@a\nACGT\n+\nIII\n", 2)
    seq, qual, off = RIO.splitFastq(b"@a\nACGT\n+\nIII\n", 2, with_qual=False)     # FastqReader.next() never looks at line 4
    assert bytes(seq) == b"ACGT"


def test_fastq_gz_and_plain_files(tmp_path):
    import gzip
    text = make_fastq(300, 5)
    (tmp_path / "a.fq").write_bytes(text)
    with gzip.open(tmp_path / "a.fq.gz", "wb") as f: f.write(text)
    a = RIO.readFastq(tmp_path / "a.fq", 4); b = RIO.readFastq(tmp_path / "a.fq.gz", 4)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all() and a[2].size == 301


def ref_nbits(seqs):
    """NucleotideBitsWriter.write + SeqBitsUtils.seqToBits (R/util/SeqBitsUtils.java:236-263, getByte :159-161)"""
    idx = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("U"): 3, ord("a"): 0, ord("c"): 1, ord("g"): 2, ord("t"): 3, ord("u"): 3}
    out = bytearray()
    for s in seqs:
        out += len(s).to_bytes(4, "big")
        for q in range(0, len(s), 4):
            v = [idx[c] for c in s[q:q + 4]] + [0, 0, 0]
            out.append((v[0] * 64 + v[1] * 16 + v[2] * 4 + v[3] - 128) & 0xFF)
    return bytes(out)


def test_nbits_encode_matches_seq_to_bits(tmp_path):
    rng = np.random.default_rng(8)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTUacgt", np.uint8), int(L)).tolist()) for L in list(range(0, 40)) + [150, 1001, 4096]]
    seqs += [b"CACGAGACCTCTCTACATCTCGTATGCCGTCTTCTGCTTGAAAAAAAAAAGGCAGCT", b"TCGTATGGATGAGACAGCTTGAACACACAA"]      # NucleotideBitsReader.main
    seq = np.frombuffer(b"".join(seqs), np.uint8); off = np.zeros(len(seqs) + 1, np.int64); np.cumsum([len(s) for s in seqs], out=off[1:])
    n = RIO.writeNbits(tmp_path / "x.nbits", seq, off)
    data = (tmp_path / "x.nbits").read_bytes()
    assert n == len(data) and data == ref_nbits(seqs)
    with pytest.raises(N.NativeError, match="non-ACGTU"):
        RIO.writeNbits(tmp_path / "y.nbits", np.frombuffer(b"ACGNA", np.uint8), np.array([0, 5], np.int64))


# ---- .gz input (FileUtils.getTextFileReader, R/util/FileUtils.java:50-57): every member of the file, BGZF in parallel ----
def _bgzf(data, block=60000):
    """the byte string bgzip writes: members of at most 64 KiB with a 'B','C' extra field holding the member's size, + the empty EOF member"""
    import struct, zlib
    out = []
    for a in list(range(0, len(data), block)) + [None]:
        chunk = b"" if a is None else data[a:a + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        out.append(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) + body
                   + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


@pytest.mark.parametrize("threads", [1, 5, 0])
def test_gunzip_single_member_concatenated_members_and_bgzf(threads):
    import gzip
    text = make_fastq(3000, 9)
    assert len(text) > 400_000
    assert RIO.gunzip(gzip.compress(text), threads).tobytes() == text
    parts = [text[:100], text[100:250_000], b"", text[250_000:]]
    assert RIO.gunzip(b"".join(gzip.compress(p) for p in parts), threads).tobytes() == text          # GZIPInputStream reads every member
    assert RIO.gunzip(gzip.compress(text) + b"\0" * 7, threads).tobytes() == text                   # zero padding after the last member
    # what follows a member and is no gzip header ends the stream silently (GZIPInputStream.readTrailer swallows the header error);
    # a member BEHIND such bytes is therefore never reached
    assert RIO.gunzip(gzip.compress(text) + b"not a gzip header", threads).tobytes() == text
    assert RIO.gunzip(gzip.compress(text[:5000]) + b"\0\0\0" + gzip.compress(text[5000:]), threads).tobytes() == text[:5000]
    bg = _bgzf(text)
    assert gzip.decompress(bg) == text                                                                 # the fixture is a valid gzip file
    assert RIO.gunzip(bg, threads).tobytes() == text
    assert RIO.gunzip(gzip.compress(b""), threads).size == 0 and RIO.gunzip(b"", threads).size == 0


def test_gunzip_refuses_damaged_input():
    import gzip
    z = gzip.compress(make_fastq(500, 2))
    for bad, what in ((z[:len(z) // 2], "unexpected end"), (b"@r1\nACGT\n+\nIIII\n", "not in gzip format"), (z[:40] + bytes([z[40] ^ 0x55]) + z[41:], "gzip")):
        with pytest.raises(N.NativeError, match=what):
            RIO.gunzip(bad)
    bg = bytearray(_bgzf(make_fastq(800, 3), 20000))
    bg[len(bg) // 2] ^= 0xFF
    with pytest.raises(N.NativeError):
        RIO.gunzip(bytes(bg), 4)


def test_read_fastq_from_gz_file(tmp_path):
    import gzip
    text = make_fastq(400, 5, b"\r\n")
    (tmp_path / "a.fq").write_bytes(text)
    (tmp_path / "a.fq.gz").write_bytes(gzip.compress(text))
    (tmp_path / "b.fq.GZ").write_bytes(_bgzf(text, 5000))
    want = RIO.splitFastq(text)
    for name in ("a.fq", "a.fq.gz", "b.fq.GZ"):
        got = RIO.readFastq(tmp_path / name)
        assert all((a == b).all() for a, b in zip(got, want))
