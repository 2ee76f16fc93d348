#!/usr/bin/env python3
"""Generate tests/golden/stage1_rich/: four tiny stage-1 jobs off the happy path, each with the graph files the reference
would leave behind, written in the reference's on-disk format by the CPU oracle.

    python tests/golden/gen_stage1_rich.py          (needs only this repo; nothing is read from /root/reference)

tests/golden/stage1_small pins the plain case (uniform 100 bp pairs).  One JVM session over THIS fixture
(tools/replay_with_jar.sh) pins what it leaves out:
  pe/        300 bp pairs, half of the reads ragged (18..299 bases, some shorter than k: skipped, R/RNABloom.java:567-570),
             lower-case stretches, U for T, N, qualities '#' (PHRED 2: cuts a segment at -q 3) and '$' (PHRED 3: does not),
             -left / -right -revcomp-right: segmentation (R/util/SeqUtils.java:1432-1438), the reverse-complement iterators
             (R/RNABloom.java:540-545), the read-length quartiles behind the pair distance (:1010-1098)
  stranded/  the same two files with -stranded: HashFunction instead of CanonicalHashFunction (R/graph/BFDBG.java:90-96)
  sef/       L.fq alone as -sef: one forward file, paired k-mers of single reads (:7106-7112)
  long/      six long reads (0.8-1.6 kb, 4 % substitutions, mostly unmarked) as -long at k = 35: no pair filter at all
             (initializeGraph(..., useReadPairedKmers = false) :7126-7128, populateGraph2 passes false :1313-1316),
             readPairedKmersDistance stays -1 in the graph file
Every counter stays below 16, so MiniFloat.increment never draws a random number (R/util/MiniFloat.java:31-38): Java, oracle
and HIP path must agree on every byte.  File formats as in gen_stage1_small.py."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rna-bloom_amd"), os.path.dirname(os.path.abspath(__file__))]
from oracle import rbo                      # noqa: E402
from rnabloom import synth                  # noqa: E402  (numpy read generator; no device code)
from rnabloom import io as RIO              # noqa: E402  (host FASTQ splitter)
from gen_stage1_small import expected_size, java_float     # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "stage1_rich")
FPR, NUM_HASH, MIN_PAIRS, MIN_Q, SAMPLE = 0.01, 2, 10, 3, 1000
PAIRS, L = 32, 300


def quartiles(lengths):                     # rnabloom.util.Common.getQuartiles(int[]) :134-164 -> (q1, median, q3)
    a = sorted(int(x) for x in lengths)
    n = len(a); half = n // 2; q1i = n // 4; q3i = half + q1i
    med = (a[half - 1] + a[half]) // 2 if n % 2 == 0 else a[half]
    if n % 4 == 0: return (a[q1i - 1] + a[q1i]) // 2, med, (a[q3i - 1] + a[q3i]) // 2
    return a[q1i], med, a[q3i]


def weighted_q1(paths, k):                  # RNABloom.getReadLengthQuartiles :1034-1098 (Math.rint = round half to even)
    sizes, q1s = [], []
    for p in paths:
        _, _, off = RIO.readFastq(p, 2)
        lens = [int(x) for x in np.diff(off) if x >= k][:SAMPLE]                    # getReadLengths :917-956
        q1s.append(quartiles(lens)[0]); sizes.append(os.path.getsize(p))
    tot = float(sum(sizes))
    return int(np.rint(sum(s / tot * q for s, q in zip(sizes, q1s))))


def dress(reads, quals, rng):
    """ragged lengths, lower case, U, '$' qualities on top of synth's substitutions ('#') and N"""
    out = []
    for i in range(reads.shape[0]):
        n = L if rng.random() < 0.7 else int(rng.integers(18, L))
        n = {5: 19, 11: 25, 17: 24}.get(i, n)                                      # shorter than k (skipped), exactly k, one short of it
        s, q = reads[i, :n].copy(), quals[i, :n].copy()
        for _ in range(int(rng.integers(0, 3))):                                   # lower-case stretches
            a = int(rng.integers(0, max(1, n - 6))); b = min(n, a + int(rng.integers(1, 7)))
            s[a:b] = np.where(s[a:b] == ord("N"), s[a:b], s[a:b] | 0x20)
        t = np.flatnonzero(s == ord("T"))
        if t.size: s[rng.choice(t, min(2, t.size), replace=False)] = ord("U")
        q[rng.random(n) < 0.01] = ord("$")
        out.append((s.tobytes(), q.tobytes()))
    return out


def write_fastq(path, recs, tag):
    with open(path, "wb") as f:
        for i, (s, q) in enumerate(recs):
            f.write(b"@" + tag + b"%d\n" % i + s + b"\n+\n" + q + b"\n")


def long_reads(rng):
    genome, tstart, tlen = synth.make_transcriptome(9000, 777)
    recs = []
    for i in range(6):
        t = int(rng.integers(0, len(tstart)))
        n = int(min(tlen[t], rng.integers(800, 1600)))
        a = int(tstart[t] + rng.integers(0, tlen[t] - n + 1))
        s = genome[a:a + n].copy(); q = np.full(n, ord("5"), np.uint8)
        e = np.flatnonzero(rng.random(n) < 0.04)
        s[e] = np.frombuffer(b"CGTA", np.uint8)[np.searchsorted(np.frombuffer(b"ACGT", np.uint8), s[e])]
        q[e[rng.random(e.size) < 0.3]] = ord("#")                                   # a third of the errors are marked
        s[rng.random(n) < 0.002] = ord("N")
        recs.append((s.tobytes(), q.tobytes()))
    return recs


def save(og, out, k, stranded, dist, size, with_pairs):
    os.makedirs(out, exist_ok=True)
    cbf = og.cbf_bytes()
    assert cbf.max() < 16, "a counter reached the probabilistic regime (%d)" % cbf.max()
    fprs = og.fprs()
    assert max(fprs[:2]) <= 0.02, "FPR above 2 x -fpr: the reference would resize and repopulate (R/RNABloom.java:7142-7180)"
    g = os.path.join(out, "rnabloom.graph")
    with open(g, "w") as w:
        w.write("dbgbfCbfMaxNumHash:%d\nstranded:%s\nk:%d\nreadPairedKmersDistance:%d\nfragmentPairedKmersDistance:%d\n"
                % (NUM_HASH, "true" if stranded else "false", k, dist, -1))
    files = [(".dbgbf", og.dbgbf_bytes(), fprs[0]), (".cbf", cbf, fprs[1])] + ([(".rpkbf", og.rpkbf_bytes(), fprs[2])] if with_pairs else [])
    for ext, raw, fpr in files:
        raw.tofile(g + ext)
        with open(g + ext + ".desc", "w") as w:
            w.write("size:%d\nnumhash:%d\nfpr:%s\n" % (size, NUM_HASH, java_float(fpr)))
    return {"max_counter": int(cbf.max()), "popcounts": [int(x) for x in og.popcounts()], "fprs": [java_float(x) for x in fprs]}


def run(files, k, nk, stranded, with_pairs, out):
    """files: [(path, reverse_complement)], in populateGraph2's order (forward files, then reverse files, then long)"""
    size = expected_size(nk, FPR, NUM_HASH)
    dist = max(1, weighted_q1([p for p, _ in files], k) - k - MIN_PAIRS) if with_pairs else -1     # setReadLengthBasedParams :1010-1024
    og = rbo.Graph(size, size, size if with_pairs else 64, NUM_HASH, NUM_HASH, NUM_HASH, k, stranded, with_pairs, 0)
    if with_pairs: og.set_read_pair_distance(dist)
    stats = []
    for p, rc in files:
        seq, qual, off = RIO.readFastq(p, 2)
        st = og.add_reads(seq, qual, off, MIN_Q, (rbo.STORE_READ_PAIRS if with_pairs else 0) | (rbo.REVCOMP if rc else 0))
        stats.append({"file": os.path.basename(p), "reverse_complement": bool(rc), "reads": int(off.size - 1), "kmers": int(st.kmers), "pairs": int(st.pairs)})
    m = save(og, out, k, stranded, dist, size, with_pairs)
    m.update({"k": k, "nk": nk, "stranded": stranded, "read_pairs": with_pairs, "filter_size": size, "read_pair_distance": dist, "files": stats})
    return m


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20260929)
    d = synth.generate_pairs(PAIRS, G=4000, L=L, err=0.004, n_rate=2e-3, seed=20260929, frag_mean=420.0, frag_sd=40.0, uniform_expr=True)
    Lq, Rq, Gq = os.path.join(OUT, "L.fq"), os.path.join(OUT, "R.fq"), os.path.join(OUT, "long.fq")
    write_fastq(Lq, dress(d["left"], d["lqual"], rng), b"pair/1_")
    write_fastq(Rq, dress(d["right"], d["rqual"], rng), b"pair/2_")
    write_fastq(Gq, long_reads(rng), b"long")
    common = "-t 1 -fpr 0.01 -stage 1 -savebf"
    man = {"fpr": FPR, "num_hash": NUM_HASH, "min_base_qual": MIN_Q, "runs": {
        "pe": dict(run([(Lq, False), (Rq, True)], 25, 7000, False, True, os.path.join(OUT, "pe")),
                   args="-left L.fq -right R.fq -revcomp-right -k 25 -nk 7000 " + common),
        "stranded": dict(run([(Lq, False), (Rq, True)], 25, 7000, True, True, os.path.join(OUT, "stranded")),
                         args="-left L.fq -right R.fq -revcomp-right -stranded -k 25 -nk 7000 " + common),
        "sef": dict(run([(Lq, False)], 25, 4000, False, True, os.path.join(OUT, "sef")),
                    args="-sef L.fq -k 25 -nk 4000 " + common),
        "long": dict(run([(Gq, False)], 35, 9000, False, False, os.path.join(OUT, "long")),
                     args="-long long.fq -k 35 -nk 9000 " + common),
    }}
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as w:
        json.dump(man, w, indent=1)
    for name, m in man["runs"].items():
        print(name, {k_: m[k_] for k_ in ("filter_size", "read_pair_distance", "max_counter", "popcounts", "fprs")}, m["files"])


if __name__ == "__main__":
    main()
