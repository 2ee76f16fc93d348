"""Extracts the two segmentation patterns of the reference — DATA, not code: the PHRED33 alphabet literal and the two pattern
templates of rnabloom.util.SeqUtils (src/rnabloom/util/SeqUtils.java:1426-1438: getPhred33Pattern, getNucleotideCharsPattern) —
into tests/golden/seq_patterns.json.  Run in the build container (needs /root/reference); the JSON is what travels.

    python tests/golden/gen_seq_patterns.py
"""
import json
import os
import re

SRC = "/root/reference/src/rnabloom/util/SeqUtils.java"
text = open(SRC).read()


def java_string(lit):
    """value of a Java string literal body (the escapes that occur here: \\" and \\\\)"""
    out, i = [], 0
    while i < len(lit):
        if lit[i] == "\\":
            out.append(lit[i + 1]); i += 2
        else:
            out.append(lit[i]); i += 1
    return "".join(out)


m = re.search(r'String PHRED33 = "((?:[^"\\]|\\.)*)";', text)
phred33 = java_string(m.group(1))
assert len(phred33) == 94 and phred33[0] == "!" and phred33[-1] == "~"
# getPhred33Pattern: "[\\Q" + PHRED33.substring(minQual) + "\\E]{" + minLength + ",}"
q = re.search(r'getPhred33Pattern\(int minQual, int minLength\)\s*\{\s*return Pattern\.compile\("((?:[^"\\]|\\.)*)" \+ PHRED33\.substring\(minQual\) \+ "((?:[^"\\]|\\.)*)" \+ Integer\.toString\(minLength\) \+ "((?:[^"\\]|\\.)*)"\);', text)
# getNucleotideCharsPattern: "[ACGTU]{" + minLength + ",}", CASE_INSENSITIVE
s = re.search(r'getNucleotideCharsPattern\(int minLength\)\s*\{\s*return Pattern\.compile\("((?:[^"\\]|\\.)*)" \+ Integer\.toString\(minLength\) \+ "((?:[^"\\]|\\.)*)", Pattern\.(\w+)\);', text)
out = {
    "source": "src/rnabloom/util/SeqUtils.java:1426-1438 (RNA-Bloom v2.0.1)",
    "phred33": phred33,
    "qual_pattern_parts": [java_string(q.group(1)), "<PHRED33.substring(minQual)>", java_string(q.group(2)), "<minLength>", java_string(q.group(3))],
    "seq_pattern_parts": [java_string(s.group(1)), "<minLength>", java_string(s.group(2))],
    "seq_pattern_flags": [s.group(3)],
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "seq_patterns.json")
json.dump(out, open(path, "w"), indent=1)
print(out)
