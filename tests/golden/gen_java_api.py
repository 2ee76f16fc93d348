#!/usr/bin/env python3
"""Writes tests/golden/java_api.json: the public constructors and methods (name + parameter types) of the reference classes
the drop-in boundary keeps: rnabloom.graph.BloomFilterDeBruijnGraph and rnabloom.bloom.{BloomFilter,
CountingBloomFilter, PairedKeysBloomFilter} and rnabloom.graph.{Kmer, CanonicalKmer} (round 6).  A fixture is data: names and types, no method bodies.
    python tests/golden/gen_java_api.py [/root/reference]"""
import json, os, re, sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
FILES = {"BloomFilterDeBruijnGraph": "src/rnabloom/graph/BloomFilterDeBruijnGraph.java", "BloomFilter": "src/rnabloom/bloom/BloomFilter.java",
         "CountingBloomFilter": "src/rnabloom/bloom/CountingBloomFilter.java", "PairedKeysBloomFilter": "src/rnabloom/bloom/PairedKeysBloomFilter.java",
         "Kmer": "src/rnabloom/graph/Kmer.java", "CanonicalKmer": "src/rnabloom/graph/CanonicalKmer.java"}


def public_methods(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = []
    for m in re.finditer(r"\bpublic\s+((?:static\s+|final\s+|synchronized\s+)*)([\w<>\[\], ]+?\s+)?(\w+)\s*\(([^)]*)\)\s*(?:throws [\w, .]+)?\s*\{", text):
        params = [re.sub(r"\bfinal\s+", "", p).strip().rsplit(None, 1)[0].replace(" ", "") for p in m.group(4).split(",") if p.strip()]
        out.append({"name": m.group(3), "params": params, "static": "static" in m.group(1)})
    return out


api = {cls: public_methods(open(os.path.join(ref, path)).read()) for cls, path in FILES.items()}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "java_api.json"), "w") as fh:
    json.dump(api, fh, indent=1, sort_keys=True)
print({k: len(v) for k, v in api.items()})
