#!/usr/bin/env python3
"""Generate tests/golden/stage1_small/: a tiny paired-end stage-1 job and the graph files the reference would leave
behind for it, written in the reference's own on-disk format by the CPU oracle.

    python tests/golden/gen_stage1_small.py          (needs only this repo; nothing is read from /root/reference)

Why it exists: the reference ships no golden vectors and this image has no JVM, so the oracle's filter / counter /
segmentation / pair semantics are pinned by reading alone.  This fixture is the smallest complete job whose outputs
can be compared byte for byte with a real RNA-Bloom run the day a JDK is at hand (tools/replay_with_jar.sh):
  * reads: 60 pairs x 100 bp from a 2.5 kb transcriptome, right reads as sequenced (-revcomp-right), substitution
    errors carry quality '#' (PHRED 2 < -q 3, so they cut segments, R/RNABloom.java:572-577), a few N bases;
  * every counter stays below 16, so MiniFloat.increment never draws a random number (R/util/MiniFloat.java:31-38) and
    the counting filter is deterministic — Java, oracle and HIP path must agree on every byte;
  * -nk 4000 -fpr 0.01 gives the filter sizes of BloomFilter.getExpectedSize (R/bloom/BloomFilter.java:196-199,
    R/RNABloom.java:6985-7011); the read-pair distance is max(1, q1 - k - 10) = 65 (R/RNABloom.java:1010-1024).
Files follow R/graph/BloomFilterDeBruijnGraph.java:297-329 (graph desc), R/bloom/BloomFilter.java:113-124 and
R/bloom/CountingBloomFilter.java:106-118 (<file>.desc: size / numhash / fpr, <file>: raw bytes)."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rna-bloom_amd")]
from oracle import rbo                      # noqa: E402
from rnabloom import synth                  # noqa: E402  (numpy read generator; no device code)

OUT = os.path.join(ROOT, "tests", "golden", "stage1_small")
K, NK, FPR, NUM_HASH, PAIRS, L, MIN_PAIRS, MIN_Q = 25, 4000, 0.01, 2, 60, 100, 10, 3


def expected_size(n, fpr, h):               # BloomFilter.getExpectedSize with a Java float fpr
    f = float(np.float32(fpr))
    r = -h / math.log(1.0 - math.exp(math.log(f) / h))
    return int(math.ceil(n * r))


def java_float(x):                          # Float.toString
    x = np.float32(x)
    if x == 0: return "0.0"
    e = int(np.format_float_scientific(x, unique=True).split("e")[1])
    if -3 <= e < 7:
        p = np.format_float_positional(x, unique=True, trim="0")
        return p + "0" if p.endswith(".") else p
    m = np.format_float_scientific(x, unique=True, trim="0").split("e")[0]
    return "%sE%d" % (m + "0" if m.endswith(".") else m, e)


def write_fastq(path, reads, quals, mate):
    with open(path, "wb") as f:
        for i in range(reads.shape[0]):
            f.write(b"@pair%d/%d\n" % (i, mate) + reads[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    d = synth.generate_pairs(PAIRS, G=2500, L=L, err=0.004, n_rate=2e-3, sigma=1.0, seed=20260928, frag_mean=180.0, frag_sd=20.0)
    write_fastq(os.path.join(OUT, "L.fq"), d["left"], d["lqual"], 1)
    write_fastq(os.path.join(OUT, "R.fq"), d["right"], d["rqual"], 2)
    size = expected_size(NK, FPR, NUM_HASH)
    dist = max(1, L - K - MIN_PAIRS)
    og = rbo.Graph(size, size, size, NUM_HASH, NUM_HASH, NUM_HASH, K, False, True, 0)
    og.set_read_pair_distance(dist)
    stats = []
    for name, q, rc in (("left", "lqual", False), ("right", "rqual", True)):          # populateGraph2: forward files, then reverse
        s, off = synth.flat(d[name]); ql, _ = synth.flat(d[q])
        st = og.add_reads(s, ql, off, MIN_Q, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
        stats.append({"kmers": int(st.kmers), "pairs": int(st.pairs)})
    cbf = og.cbf_bytes()
    assert cbf.max() < 16, "a counter reached the probabilistic regime (%d): shrink the coverage" % cbf.max()
    pop = og.popcounts()
    fprs = og.fprs()
    g = os.path.join(OUT, "rnabloom.graph")
    with open(g, "w") as w:                                                           # saveDesc :297-305
        w.write("dbgbfCbfMaxNumHash:%d\nstranded:false\nk:%d\nreadPairedKmersDistance:%d\nfragmentPairedKmersDistance:%d\n" % (NUM_HASH, K, dist, -1))
    for ext, raw, fpr in ((".dbgbf", og.dbgbf_bytes(), fprs[0]), (".cbf", cbf, fprs[1]), (".rpkbf", og.rpkbf_bytes(), fprs[2])):
        raw.tofile(g + ext)
        with open(g + ext + ".desc", "w") as w:
            w.write("size:%d\nnumhash:%d\nfpr:%s\n" % (size, NUM_HASH, java_float(fpr)))
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as w:
        json.dump({"k": K, "nk": NK, "fpr": FPR, "num_hash": NUM_HASH, "filter_size": size, "read_pair_distance": dist,
                   "min_base_qual": MIN_Q, "pairs": PAIRS, "read_len": L, "files": stats, "max_counter": int(cbf.max()),
                   "popcounts": [int(x) for x in pop], "fprs": [java_float(x) for x in fprs],
                   "reference_command": "java -jar RNA-Bloom.jar -left L.fq -right R.fq -revcomp-right -k 25 -t 1 -fpr 0.01 -nk 4000 "
                                        "-stage 1 -savebf -outdir O   # then cmp O/rnabloom.graph{,.dbgbf,.cbf,.rpkbf}{,.desc}"}, w, indent=1)
    print("wrote", OUT, "filter size", size, "max counter", int(cbf.max()), "popcounts", pop, "fprs", fprs, stats)


if __name__ == "__main__":
    main()
