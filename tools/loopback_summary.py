#!/usr/bin/env python3
"""profiles/<tag>_loopback.txt out of what tools/loopback_profile.sh <tag> kernels left under gpurun_out/loop_<tag>/:

    python tools/loopback_summary.py r06 [r05] > profiles/r06_loopback.txt

the per-rank table of both drivers, the kernel time of the 8-rank pass by term (this round beside the previous round's committed
profiles/<prev>_loopback8_py_kernel_stats.csv), the RB_SHARD_ORDER_ALL=1 runs if there are any, and the runs' own JSON lines.  Also copies the two
kernel-stat files to profiles/<tag>_loopback8_{py,native}_kernel_stats.csv.  Runs on the CPU; reads nothing but those files."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (label, name prefixes) — first match wins; kernel names as rocprofv3 prints them, namespaces stripped
TERMS = [
    ("window walk (k_filter_reads*)", ("k_filter_reads",)),
    ("emit (k_hash_windows*)", ("k_hash_windows",)),
    ("grouping (part_count / part_scatter / group_buckets / group_big / seg_tiles / bucket_bounds)",
     ("k_part_", "k_group_", "k_seg_tiles", "k_bucket_bounds", "k_tile_")),
    ("pair walker + copy merge (k_pairs_reads, k_or_*)", ("k_pairs_", "k_or_")),
    ("resolve + writes (k_resolve_shard, k_emit_writes, k_own_writes, k_apply_tagged, k_cbf_heavy, select)",
     ("k_resolve_shard", "k_emit_writes", "k_own_writes", "k_apply_", "k_cbf_heavy", "k_select", "k_heavy")),
    ("probes / claims (k_shard_probe, k_own_*, k_local_*)", ("k_shard_probe", "k_own_", "k_local_")),
    ("ordered set at the owners (k_order_*)", ("k_order_",)),
    ("routing passes (k_route*, k_pick_*)", ("k_route", "k_pick_")),
    ("scans (k_scan_*)", ("k_scan",)),
    ("cache updates (k_cache_apply)", ("k_cache_",)),
    ("conflict path (k_edge_*, k_conf_*, k_run_*, k_replay_*, k_tab_emit)", ("k_edge_", "k_conf_", "k_run_", "k_replay", "k_tab_", "k_label_", "k_cs_")),
    ("fills + copies (fillBuffer, copyBuffer, k_zero16)", ("__amd_rocclr_fillBuffer", "__amd_rocclr_copyBuffer", "k_zero")),
    ("torch concatenations of the Python driver (not in the native driver)", ("at::native::CatArray",)),
    ("read synthesis (outside the pass)", ("k_synth", "at::native::", "void at::native")),
]


def bare(name):
    for p in ("rb::", "void rb::", "void "):
        if name.startswith(p):
            name = name[len(p):]
    return name


def by_term(path, steps):
    out = {label: 0.0 for label, _ in TERMS}
    out["other kernels"] = 0.0
    if not os.path.exists(path):
        return None
    for line in open(path).read().splitlines()[1:]:                  # kernel,calls,total_ms,avg_us,percent — template arguments carry commas: split from the right
        name, _, total_ms, _, _ = line.rsplit(",", 4)
        n, ms = bare(name.strip('"')), float(total_ms) / steps
        for label, pre in TERMS:
            if any(n.startswith(p) for p in pre):
                out[label] += ms
                break
        else:
            out["other kernels"] += ms
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    prev = sys.argv[2] if len(sys.argv) > 2 else "r%02d" % (int(tag[1:]) - 1)
    d = os.path.join(ROOT, "gpurun_out", "loop_" + tag)
    runs = {}
    for drv in ("native", "python"):
        for g in (1, 2, 4, 8):
            p = os.path.join(d, "%s_%d.json" % (drv, g))
            if os.path.exists(p) and os.path.getsize(p):
                runs[drv, g] = json.loads(open(p).read().strip().splitlines()[-1])
    row = lambda drv: "".join("%-10s" % ("%.1f" % runs[drv, g]["per_rank_ms_if_concurrent"] if (drv, g) in runs else "-") for g in (1, 2, 4, 8))
    print("# profiles/%s_loopback.txt — tools/loopback_profile.sh %s kernels, written by tools/loopback_summary.py: tools/loopback_bench.py, config 2 (50 M pairs, 12.26 G k-mers per "
          "step), G virtual ranks on ONE MI355X" % (tag, tag))
    print("# (wall = the sum of the ranks' work: a real G-GPU run does the ranks' work side by side, its step is about wall / G + link time).  UNMEASURED ON HARDWARE: no run with more")
    print("# than one physical GPU exists; RCCL has run at world 1 only.")
    print("# per rank if concurrent (ms):            G = 1     G = 2     G = 4     G = 8      (review's marks: Sigma <= 410 at 8 ranks = <= 52 per rank, <= 190 at 2)")
    print("#   exchange driver below the C ABI       " + row("native"))
    print("#   Python driver (torch.distributed)     " + row("python"))
    for g in (8, 2):
        for drv in ("native", "python"):
            p = os.path.join(d, "%s_%d_order_all.json" % (drv, g))
            if os.path.exists(p) and os.path.getsize(p) and (drv, g) in runs:
                j = json.loads(open(p).read().strip().splitlines()[-1])
                print("#   RB_SHARD_ORDER_ALL=1 (every run that shares a counter replayed, the engine before the ordered set O*), %s driver, %d ranks: wall %.1f ms, conflict_ops %d"
                      "  —  with O*: wall %.1f ms, conflict_ops %d" % (drv, g, j["wall_ms_per_step"], j["stats"]["conflict_ops"], runs[drv, g]["wall_ms_per_step"],
                                                                       runs[drv, g]["stats"]["conflict_ops"]))
    print("#")
    for drv, short in (("python", "py"), ("native", "native")):
        src = os.path.join(d, "kernels8_%s.csv" % drv)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(ROOT, "profiles", "%s_loopback8_%s_kernel_stats.csv" % (tag, short)))
    # one profiled pass each (--warmup 0, one step): the sums are per pass
    cols = [("%s python" % prev, by_term(os.path.join(ROOT, "profiles", "%s_loopback8_py_kernel_stats.csv" % prev), 1)),
            ("%s python" % tag, by_term(os.path.join(d, "kernels8_python.csv"), 1)),
]                                                         # the native driver's ranks are host threads whose kernels overlap on the one GPU: their durations
                                                                  # stretch each other and do not add up to work; its kernel file is committed for the call counts
    cols = [(n, c) for n, c in cols if c]
    if cols:
        print("# kernel time summed over the 8 virtual ranks, ms per pass, Python driver (its ranks run one after another: the sums add; the native driver's ranks are host threads whose "
              "kernels overlap on the one GPU and stretch each other: durations there are not work)")
        print("%-100s" % "term" + "".join("%12s" % n for n, _ in cols))
        labels = [l for l, _ in TERMS] + ["other kernels"]
        for l in labels:
            print("%-100s" % l + "".join("%12.1f" % c[l] for _, c in cols))
        skip = ("torch concatenations of the Python driver (not in the native driver)", "read synthesis (outside the pass)")
        print("%-100s" % "engine total (without synthesis and torch concatenations)" + "".join("%12.1f" % sum(v for l, v in c.items() if l not in skip) for _, c in cols))
        print("#")
    for (drv, g), j in sorted(runs.items()):
        print("# %s_%d" % (drv, g))
        print(json.dumps(j))


if __name__ == "__main__":
    main()
