#!/usr/bin/env python3
"""Generate tests/golden/nthash_kat.json from the LITERAL constant tables of the reference.

Run in the build container only (needs /root/reference):
    python tests/golden/gen_golden.py

The reference ships no tests and no JVM exists here, so outputs of the reference itself cannot be
recorded.  What this script records instead are known-answer vectors computed by a tiny pure-Python
evaluation of the table-lookup formulas *over the reference's own literal tables* (vecA/C/G/T,
msTab, seedTab, multiSeed, multiShift as parsed from R/bloom/hash/NTHash.java:30-168) — i.e. the
vectors pin the oracle (and the HIP path) to the reference's constants, independent of the
oracle's `rotl(seed, j)` construction.  The fixture holds inputs + expected outputs only.
"""
import json
import os
import random
import re

REF = "/root/reference/src/rnabloom/bloom/hash/NTHash.java"
M64 = (1 << 64) - 1


def parse_tables(src):
    consts = {m.group(1): int(m.group(2), 16)
              for m in re.finditer(r"long\s+(seed[ACGTN]|multiSeed)\s*=\s*0x([0-9a-fA-F]+)L", src)}
    consts["multiShift"] = int(re.search(r"multiShift\s*=\s*(\d+)", src).group(1))
    consts["cpOff"] = int(re.search(r"cpOff\s*=\s*0x([0-9a-fA-F]+)", src).group(1), 16)
    vec = {}
    for m in re.finditer(r"long\[\]\s+(vec[ACGTN])\s*=\s*\{(.*?)\};", src, re.S):
        toks = [t.strip() for t in m.group(2).replace("\n", " ").split(",") if t.strip()]
        vals = []
        for t in toks:
            vals.append(consts[t] if t.startswith("seed") else int(t.rstrip("L"), 16))
        assert len(vals) == 64, (m.group(1), len(vals))
        vec[m.group(1)] = vals

    def rows(name):
        body = re.search(r"%s\s*=\s*\{(.*?)\};" % name, src, re.S).group(1)
        body = re.sub(r"//.*", "", body)
        return [t.strip() for t in body.replace("\n", " ").split(",") if t.strip()]

    ms = rows(r"long\[\]\[\]\s+msTab")
    st = rows(r"long\[\]\s+seedTab")
    assert len(ms) == 256 and len(st) == 256
    msTab = [vec[t] for t in ms]
    seedTab = [consts[t] for t in st]
    return consts, msTab, seedTab


def main():
    src = open(REF).read()
    consts, msTab, seedTab = parse_tables(src)
    cp = consts["cpOff"]

    def ntp64(s, k):            # NTHash.java:332-337
        h = 0
        for i in range(k):
            h ^= msTab[s[i]][(k - 1 - i) % 64]
        return h

    def ntp64rc(s, k):          # :367-373
        h = 0
        for i in range(k):
            h ^= msTab[s[i] & cp][i % 64]
        return h

    def signed(x):
        return x - (1 << 64) if x >> 63 else x

    def ntm64(b, k, m):         # :518-527
        out = [b]
        for i in range(1, m):
            t = (b * ((i ^ ((k * consts["multiSeed"]) & M64)) & M64)) & M64
            t ^= t >> consts["multiShift"]
            out.append(t)
        return out

    rng = random.Random(20260928)
    alphabet = b"ACGTUacgtu"
    kats = []
    fixed = [b"ACGTACGTACGTACGT", b"ACGUACGUACGUACGU"]          # NTHash.main :746-754
    for s in fixed:
        kats.append((s, len(s)))
    for k in (1, 2, 11, 13, 25, 31, 32, 35, 63, 64, 65, 97, 128):
        for _ in range(4):
            s = bytes(rng.choice(alphabet) for _ in range(k))
            kats.append((s, k))
    # strings with non-nucleotide characters hash those positions as seed 0
    kats.append((b"ACGTNACGTNACGTNACGTNACGTN", 25))
    kats.append((b"NNNNNNNNNNNNN", 13))

    vectors = []
    for s, k in kats:
        f, r = ntp64(s, k), ntp64rc(s, k)
        c = r if signed(r) < signed(f) else f
        vectors.append({"seq": s.decode(), "k": k, "fwd": "%016x" % f, "rev": "%016x" % r,
                        "canonical": "%016x" % c,
                        "multi4_fwd": ["%016x" % v for v in ntm64(f, k, 4)]})

    # rolling KATs: one long sequence, every window hashed from scratch with the literal tables
    long_seq = bytes(rng.choice(b"ACGT") for _ in range(300))
    rolls = []
    for k in (25, 35, 64, 70):
        fs = ["%016x" % ntp64(long_seq[i:i + k], k) for i in range(len(long_seq) - k + 1)]
        rs = ["%016x" % ntp64rc(long_seq[i:i + k], k) for i in range(len(long_seq) - k + 1)]
        rolls.append({"k": k, "fwd": fs, "rev": rs})

    out = {
        "_generator": "tests/golden/gen_golden.py over the literal tables of R/bloom/hash/NTHash.java",
        "seeds": {n: "%016x" % consts[n] for n in ("seedA", "seedC", "seedG", "seedT", "seedN")},
        "multiSeed": "%016x" % consts["multiSeed"], "multiShift": consts["multiShift"], "cpOff": cp,
        "seedTab_nonzero": {str(i): "%016x" % v for i, v in enumerate(seedTab) if v},
        "msTab_checksum": "%016x" % (sum((i * 64 + j + 1) * msTab[i][j] for i in range(256) for j in range(64)) & M64),
        "msTab_rows": {str(i): ["%016x" % v for v in msTab[i]] for i in (1, 3, 4, 5, 7, 65, 67, 71, 84, 85)},
        "kat": vectors,
        "roll_seq": long_seq.decode(),
        "roll": rolls,
    }
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nthash_kat.json")
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=0, separators=(",", ":"))
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
