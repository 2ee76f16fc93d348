#!/usr/bin/env python3
"""b.csv - a.csv of two profiles/summarize.py kernel summaries (kernel,calls,total_ms,avg_us,percent): what the second run did beyond the first.
    python tools/diff_kernel_stats.py a.csv b.csv"""
import sys


def load(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        k, calls, tot, _, _ = line.rsplit(",", 4)
        out[k] = (int(calls), float(tot))
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k, (cb, tb) in b.items():
    ca, ta = a.get(k, (0, 0.0))
    if cb - ca > 0 or tb - ta > 0.01:
        rows.append((tb - ta, k, cb - ca))
tot = sum(r[0] for r in rows)
print("kernel,calls,total_ms,avg_us,percent")
for t, k, c in sorted(rows, reverse=True):
    print("%s,%d,%.3f,%.1f,%.2f" % (k, c, t, t * 1e3 / max(c, 1), 100.0 * t / max(tot, 1e-9)))
