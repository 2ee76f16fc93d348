#!/usr/bin/env python3
"""Idle time of the device inside the timed steps of bench.py, from a rocprofv3 --kernel-trace CSV: the union of all kernels' busy intervals against the wall
time of the window, the largest gaps and which kernels border them.   tools/trace_gaps.py <kernel_trace.csv> [skip_fraction=0.4]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "")) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
# the LAST step: from its clearAllBf (the last cluster of k_zero16 launches: clusters are more than 50 ms apart) to the end of the trace
z = [e[0] for e in ev if e[2].startswith("k_zero16")]
if z:
    lo = z[-1]
    for a, b in zip(reversed(z[:-1]), reversed(z[1:])):
        if b - a > 50_000_000: break
        lo = a
else:
    lo = t0 + int((t1 - t0) * skip)
ev = [e for e in ev if e[0] >= lo]
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
last_name = ev[0][2]
for s, e, name, q in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, name))
        cur_s, cur_e = s, e
    if e > cur_e:
        cur_e, last_name = e, name
busy += cur_e - cur_s
wall = cur_e - ev[0][0]
print("window %.1f ms, device busy %.1f ms, idle %.1f ms in %d gaps" % (wall / 1e6, busy / 1e6, (wall - busy) / 1e6, len(gaps)))
import collections
by = collections.Counter()
for g, a, b in gaps:
    by[(a.split("(")[0][:40], b.split("(")[0][:40])] += g
for (a, b), g in by.most_common(25):
    print("  %8.2f ms idle between %-40s -> %s" % (g / 1e6, a, b))
