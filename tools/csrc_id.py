#!/usr/bin/env python3
"""Identity of the kernel sources: sha256 over rna-bloom_amd/csrc/* and include/rb_capi.h (names + bytes, sorted), 16 hex digits.
The Makefile compiles it into librb_hip.so (rb_build_id()), tools/final_profile.sh writes it next to the PMC summaries
(profiles/rNN_pmc_meta.json) and bench.py reports counter bytes only when the two agree — counters of one tree are never
quoted beside the times of another."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_id(root=ROOT):
    h = hashlib.sha256()
    d = os.path.join(root, "rna-bloom_amd", "csrc")
    files = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".hpp"))]
    files.append(os.path.join(root, "include", "rb_capi.h"))
    for p in files:
        h.update(os.path.relpath(p, root).encode() + b"\0")
        h.update(open(p, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sys.stdout.write(csrc_id())
