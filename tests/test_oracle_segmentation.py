"""The oracle's read segmentation against the reference's OWN regular expressions (reference-held material
that pins something below the hash layer without a JVM).  tests/golden/seq_patterns.json holds the PHRED33 alphabet and the two
pattern templates of rnabloom.util.SeqUtils (src/rnabloom/util/SeqUtils.java:1426-1438), extracted by
tests/golden/gen_seq_patterns.py; Python's `re` runs them as they stand (`\\Q...\\E` -> re.escape), driven by the nested
`while (mQual.find()) { mSeq.region(...); while (mSeq.find())` loop of FastqToGraphWorker (src/rnabloom/RNABloom.java:572-577),
and `oracle/rb_oracle.c::rbo_segments` — what the HIP path is compared with — must cut every read in the same places."""
import json
import os
import re

import numpy as np
import pytest

from oracle import rbo

HERE = os.path.dirname(os.path.abspath(__file__))
PAT = json.load(open(os.path.join(HERE, "golden", "seq_patterns.json")))


def java_patterns(min_qual, min_len):
    """the two java.util.regex patterns, rebuilt from the reference's literal parts for Python's re (bytes patterns: Java's
    CASE_INSENSITIVE without UNICODE_CASE folds ASCII letters only, and so do bytes patterns in Python)"""
    qp = PAT["qual_pattern_parts"]
    assert qp[0] == "[\\Q" and qp[2] == "\\E]{" and qp[4] == ",}" and PAT["seq_pattern_flags"] == ["CASE_INSENSITIVE"]
    alphabet = PAT["phred33"][min_qual:]                                  # PHRED33.substring(minQual)
    qual = "[" + re.escape(alphabet) + "]{" + str(min_len) + ",}"        # \Q...\E = the characters literally
    sp = PAT["seq_pattern_parts"]
    seq = sp[0] + str(min_len) + sp[2]
    return re.compile(qual.encode("latin-1")), re.compile(seq.encode("latin-1"), re.IGNORECASE)


def reference_segments(seq, qual, k, min_qual, pats):
    mq, ms = pats
    out = []
    if qual is None:                                                      # FastaToGraphWorker: the sequence pattern alone (:677-697)
        return [[m.start(), m.end()] for m in ms.finditer(seq)]
    for q in mq.finditer(qual):
        for m in ms.finditer(seq, q.start(), q.end()):                    # Matcher.region(start, end) + find()
            out.append([m.start(), m.end()])
    return out


def test_pattern_fixture_is_the_reference_alphabet():
    assert PAT["phred33"] == "".join(chr(c) for c in range(33, 127))     # PHRED+33: '!' (0) ... '~' (93)


@pytest.mark.parametrize("k,min_q", [(25, 3), (5, 3), (11, 0), (7, 20), (3, 40), (1, 3), (31, 93)])
def test_oracle_segments_equal_the_reference_regexes_on_random_reads(k, min_q):
    rng = np.random.default_rng(k * 1000 + min_q)
    pats = java_patterns(min_q, k)
    seq_alpha = np.frombuffer(b"ACGTUacgtuNnRYKMSWXBDHV.-*", np.uint8)
    n_reads = 10_000 if k >= 5 else 3_000
    bad = 0
    for i in range(n_reads):
        n = int(rng.integers(0, 200)) if i % 50 else int(rng.integers(0, 4 * k + 2))
        # bases: mostly ACGT with runs broken at random places (so that runs around length k are common)
        p = np.full(seq_alpha.size, 0.02 / (seq_alpha.size - 4)); p[:4] = 0.245
        seq = seq_alpha[rng.choice(seq_alpha.size, n, p=p / p.sum())].tobytes()
        # qualities: every byte value occurs (below '!', above '~', high bit set), mostly good ones
        qv = rng.integers(0, 256, n, dtype=np.int64)
        good = rng.random(n) < 0.95
        qual = np.where(good, rng.integers(33 + min_q, 127, n), qv).astype(np.uint8).tobytes()
        want = reference_segments(seq, qual, k, min_q, pats)
        got = rbo.segments(seq, qual, k, min_q).tolist()
        bad += got != want
        assert got == want, (seq, qual, got, want)
        if i % 7 == 0:
            assert rbo.segments(seq, None, k, min_q).tolist() == reference_segments(seq, None, k, min_q, pats)
    assert bad == 0


def test_oracle_segments_on_every_single_quality_byte():
    """one read per byte value: that byte in the middle of an otherwise perfect read splits it iff the reference's class rejects it"""
    k, min_q = 4, 3
    pats = java_patterns(min_q, k)
    seq = b"ACGTACGTACGT"
    for b in range(256):
        qual = bytearray(b"I" * len(seq)); qual[5] = b
        assert rbo.segments(seq, bytes(qual), k, min_q).tolist() == reference_segments(seq, bytes(qual), k, min_q, pats), b
