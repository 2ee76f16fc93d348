#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel_stats / counter_collection) into a short tracked summary.
usage: summarize.py stats <kernel_stats.csv> | pmc <counter_collection.csv> <COUNTER> | db <results.db>"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"wrapped_(\w+?)_config", name)
    if "rocprim" in name and m:
        tail = "keys" if "empty_type" in name else "pairs"
        kind = re.search(r"detail::(\w+)\(|detail::(\w+)<", name.split("target_arch")[-1])
        return "rocprim::%s(%s)" % (m.group(1), tail)
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")


def stats(path):
    rows = list(csv.DictReader(open(path)))
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        a = agg[short(r["Name"])]
        a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    print("kernel,calls,total_ms,avg_us,percent")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%s,%d,%.3f,%.1f,%.2f" % (k, c, t / 1e6, t / c / 1e3, 100 * t / tot))


def pmc(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    print("kernel,dispatches,%s_total,%s_per_dispatch" % (counter, counter))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%s,%d,%.1f,%.1f" % (k, c, t, t / c))


def db(path):
    """rocprofv3's default sqlite output (no --output-format csv): aggregate the kernel dispatches"""
    import sqlite3
    c = sqlite3.connect(path).cursor()
    agg = defaultdict(lambda: [0, 0.0])
    for name, s, e in c.execute("select name,start,end from kernels"):
        a = agg[short(name)]
        a[0] += 1; a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    print("kernel,calls,total_ms,avg_us,percent")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%s,%d,%.3f,%.1f,%.2f" % (k, n, t / 1e6, t / n / 1e3, 100 * t / tot))


if __name__ == "__main__":
    if sys.argv[1] == "db":
        db(sys.argv[2])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
