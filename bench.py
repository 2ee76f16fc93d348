#!/usr/bin/env python3
"""bench.py — k-mers/s hashed + inserted into the Bloom dBG (k=25, 150 bp paired-end reads).

One "step" = one full stage-1 pass (R/RNABloom.java:7123-7188 populateGraph2) over the synthetic
read set, starting from cleared filters: forward (left) file, then the reverse-complemented (right)
file, every k-mer through graph.add and every read-paired k-mer through rpkbf.add.  Reads are
generated on the device before the timed region and stay resident in HBM (packed 2-bit format).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P] [--genome G] [--nk NK]

N>1: one rank per GPU.  Launched as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the
script is a rank (RANK / WORLD_SIZE in the environment); launched plainly as `python bench.py --gpus N` it starts
those N ranks itself (spawn_ranks: the same torch.distributed.run command on 127.0.0.1 with a free port) and hands
their output through, so both forms print the same one line.  Reads are data-parallel, every filter
is sharded by index range and a k-mer belongs to the rank that holds its first counter; records / probes /
replies / counter writes travel by RCCL send/recv groups inside the library (csrc/rb_comm.hip, csrc/rb_shard.hip;
RB_SHARD_DRIVER=torch: all_to_all through rnabloom/sharded.py).  The job (total read pairs) is
fixed, so scaling is "strong".
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0           # MI355X spec HBM3E bandwidth (MI355X_MICROARCH.md)
SECTOR = 64                     # bytes moved per random probe (SURVEY.md §8(d) sector model; matches FETCH_SIZE)


def algorithmic_bytes(stage, n_kmers, n_pairs, n_runs, words, n_sorted=None, group_bits=32, h=2, sharded=False, k=25):
    """ALGORITHMIC bytes one step moves in each pipeline stage.
    n_kmers = k-mer occurrences, n_sorted = occurrences that survive the no-op prefilter,
    n_pairs = paired k-mers, n_runs = distinct runs, words = 32-base words."""
    passes = -(-group_bits // 8)
    n_all = n_kmers
    n_kmers = n_sorted if n_sorted is not None else n_kmers
    model = {
        # prefilter: packed reads in (16 B/word), one 64 B cache sector per window, count + mask out (8 B/word),
        # single GPU: + the rolling state saved for the emit pass (16 B/word)
        # minimizer-bucketed cache (k <= 31): a 128 B bucket is fetched when the window's minimizer changes,
        # on average every (k - m + 2) / 2 = 5.5 windows (k = 25, m = 16); sharded engine: one 64 B line per window
        # (round 3: the minimizer is that of the k-mer's middle kp = min(k, 21 or 20) bases: a fetch every (kp - m + 2) / 2 windows)
        "filter_windows": words * (24 if sharded else 40) + (int(n_all * 128 * 2 / (mpf_kp(k) - min(16, k) + 2)) if not sharded else n_all * SECTOR),
        # one-pass prefilter + emit: packed reads in (16 B/word), one 64 B cache sector per window, survivors out (12 B)
        "filter_emit": words * 16 + n_all * SECTOR + n_kmers * 12,
        # grouping stage (csrc/rb_group.hip), per launch of each kernel: the histogram pass reads the 8-byte keys; a partition
        # pass moves every 12-byte record in and out (two passes per sub-batch: the figure is per kernel launch set);
        # the bucket kernel reads the records and writes occurrence (4 B) + strength (1 B) per record and 16 B per run
        "group_part_count": n_kmers * 8 * 2,
        "group_part_scatter": n_kmers * 24 * 2,
        "group_buckets": n_kmers * (12 + 5) + n_runs * 16,
        # packed reads in (8 B codes + 4 B validity + 4 B owner per word), (h0, occurrence) out; single GPU: + the 16 B of rolling state the
        # prefilter saved, + the word's output offset and keep mask (4 B each: round 5 — the emit pass reads only the 55 % of words that keep
        # a window, but a gapped read touches every 128-byte line all the same: profiles/r05_gapped_fetch.txt)
        "hash_windows": words * (16 if sharded else 40) + n_kmers * 12,
        "count_windows": words * 12,
        "strengths": n_kmers * 5,
        "distinct_runs": n_kmers * 8 + n_runs * 16,
        # per run: h Bloom-bit sector reads + h counter claims (an atomic in L2: the sector is fetched once; its write-back is
        # the next stage's store to the same byte) + 36 B of records — still an upper bound: runs of one occurrence whose k-mer
        # is new claim nothing (k_late_claim); the counters show 0.7 of this (round 2 counted the claim twice: 2x the counters)
        "probe_claim": n_runs * (h * SECTOR + h * SECTOR + 36),
        # per run: h counter byte stores (sector write), one strength sector, 40 B of records
        "resolve_apply": n_runs * (h * SECTOR + SECTOR + 40),
        # paired k-mers behind the seen-pair cache (round 5, DESIGN.md s3 step 7 / s5: 16 W + P (128 f + miss (h 64) + new (h 64 + 64))): the reads are
        # re-walked (16 B per word); a 128-byte cache bucket is fetched once per anchor change (f = PAIR_FETCHES_PER_PAIR of a pair); a pair the
        # cache does not know (PAIR_UNKNOWN) tests its h bits (one sector each); a pair that is new to the filter (PAIR_NEW) writes them back and
        # stores its cache entry (one sector).  f / unknown / new are the rates the walker's own statistics print on this workload (RB_DEBUG=1)
        "pairs_insert": words * 16 + int(n_pairs * (128 * PAIR_FETCHES_PER_PAIR + PAIR_UNKNOWN * h * SECTOR + PAIR_NEW * (h * SECTOR + SECTOR))),
    }
    return model.get(stage)


# the pair walker's cache statistics on config 2 (profiles/r05_pairs_seen.txt, HISTORY "seen-pair cache"): bucket fetches per pair, the share of
# pairs the cache does not vouch for (10.9 % of the first file's, 15.4 % of the second's), the share that sets a new bit (72 M of 943 M pairs)
PAIR_FETCHES_PER_PAIR, PAIR_UNKNOWN, PAIR_NEW = 0.37, 0.131, 0.076


def random_request_bytes(stage, n_pairs, h=2):
    """The part of a stage's model that is single-sector requests at random places.  FETCH_SIZE tallies those at their true 64 bytes while it
    tallies every 128-byte line of a streamed or bucket read as 64 (profiles/r05_pmc_calibration.txt), so for a stage that mixes the two the
    corrected counter is 2 x FETCH_SIZE minus these bytes (doubling would count them twice)."""
    if stage == "pairs_insert":
        return n_pairs * PAIR_UNKNOWN * h * SECTOR
    return 0.0


# Stages whose model is an UPPER BOUND by construction (the contract test lets their counters / model ratio fall below 0.6), with the reason:
UPPER_BOUND_MODELS = {
    "probe_claim": "charges every run h Bloom-bit sectors + h counter claims; runs of one occurrence whose k-mer is new claim nothing (k_late_claim) "
                   "and with index-keyed grouping probe 0 of consecutive runs shares sectors - the counters show 0.55-0.6 of it",
}


# dominant-stage kernels in the committed rocprofv3 PMC summaries (profiles/r03_pmc_*.csv: separate
# --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this same command, values in KB per dispatch).
# sort_occurrences is rocPRIM's onesweep: 1 histogram + 4 scatter dispatches per launch of the stage, under one kernel
# name that also covers the (small) sorts of the conflict path, so its bytes are the name's total over the
# number of sub-batches (= dispatches of k_probe) — an upper bound, the conflict sorts add < 8 %.
PMC_KERNELS = {"filter_windows": "rb::k_filter_reads", "hash_windows": "rb::k_hash_windows_resume",
               "probe_claim": "k_probe", "resolve_apply": "k_resolve_apply", "pairs_insert": "k_pairs_reads",
               "group_part_count": "rb::k_part_count", "group_part_scatter": "rb::k_part_scatter", "group_buckets": "rb::k_group_buckets"}
PMC_TAG = next((t for t in ("r06", "r05", "r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", t + "_pmc_fetch_size.csv"))), "r06")
PMC_FILES = (PMC_TAG + "_pmc_fetch_size.csv", PMC_TAG + "_pmc_write_size.csv")


def pmc_meta():
    """Which tree the committed PMC summaries profiled: profiles/<tag>_pmc_meta.json, written by tools/final_profile.sh on the box that ran
    the passes ({"csrc_id": tools/csrc_id.py of that tree = rb_build_id() of its library, "git_head": ...}).  Summaries without the file
    (rounds 1-4) have no known tree."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_TAG + "_pmc_meta.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}
# Correction of FETCH_SIZE (MI355X_MICROARCH.md, HBM / rocprofv3 section: the counter tallies 128-byte requests of wide
# coalesced streaming reads as 64 bytes; "other access widths are uncalibrated: calibrate on a known byte count in your
# own access pattern").  Calibration committed in profiles/r03_pmc_calibration.txt: (1) k_part_count reads exactly 8 bytes per
# record in the 8-bytes-per-lane streaming pattern all grouping kernels use, and its FETCH_SIZE comes out at half of that;
# (2) tools/microbench/gather_bench calib: 2^28 random requests of 8 B, of one whole 64-byte line and of one whole 128-byte
# bucket (8 x 16 B by one lane, the prefilter cache's access) are ALL tallied as 64 bytes per request.  So x 2 for the
# streaming kernels and for filter_windows (its traffic is 128-byte bucket fetches), x 1 for kernels whose requests are
# single words at random places.
FETCH_FACTOR = {"filter_windows": 2.0, "group_part_count": 2.0, "group_part_scatter": 2.0, "group_buckets": 2.0, "hash_windows": 2.0,
                "pairs_insert": 2.0}     # (pairs_insert: streamed reads + 128-byte cache buckets x 2, minus its random bit tests — random_request_bytes)


def mpf_kp(k):
    """csrc/rb_device.hpp mpf_kp: width of the k-mer's middle piece whose minimizer picks the prefilter cache's bucket"""
    return k if k <= 21 else 21 - ((k & 1) ^ 1)


def pmc_traffic(stage, n_pairs_per_step=0):
    """HBM bytes per launch of the stage's kernel from the committed PMC summaries (None if absent): FETCH_SIZE (corrected
    as the guide prescribes, see FETCH_FACTOR) + WRITE_SIZE, both in KB per dispatch in the summaries."""
    kern = PMC_KERNELS.get(stage)
    if not kern:
        return None
    tot = 0.0
    dispatches = 0.0
    for name in PMC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            return None
        rows = [line.rsplit(",", 3) for line in open(path).read().splitlines()[1:]]
        hit = [r for r in rows if len(r) == 4 and r[0].startswith(kern)]
        if not hit:
            return None
        dispatches = sum(float(r[1]) for r in hit)
        per_dispatch = sum(float(r[2]) for r in hit) / dispatches * 1024.0
        tot += per_dispatch * (FETCH_FACTOR.get(stage, 1.0) if "fetch" in name else 1.0)
    if tot and dispatches:
        tot -= random_request_bytes(stage, n_pairs_per_step) * PMC_RUN_STEPS / dispatches
    return tot or None


PMC_RUN_STEPS = int(pmc_meta().get("run_steps", 2))        # steps of kernel work inside a PMC pass (tools/final_profile.sh writes it: `bench.py --steps 1 --warmup 1` = 2, + 2 when the host-resident leg ran too)


def pmc_step_bytes(stage, n_pairs_per_step=0):
    """HBM bytes per STEP of the stage's kernel(s) by the counters (corrected FETCH_SIZE + WRITE_SIZE), None if absent"""
    kern = PMC_KERNELS.get(stage)
    if not kern:
        return None
    tot = 0.0
    for name in PMC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            return None
        rows = [line.rsplit(",", 3) for line in open(path).read().splitlines()[1:]]
        hit = [r for r in rows if len(r) == 4 and r[0].startswith(kern)]
        if not hit:
            return None
        tot += sum(float(r[2]) for r in hit) * 1024.0 * (FETCH_FACTOR.get(stage, 1.0) if "fetch" in name else 1.0)
    return tot / PMC_RUN_STEPS - random_request_bytes(stage, n_pairs_per_step)


def pmc_path_bytes(n_pairs_per_step=0):
    """HBM bytes per STEP of EVERY kernel of the step by the committed counters (FETCH_SIZE corrected per kernel as above, x 1 for the
    kernels without a calibration, + WRITE_SIZE), and the part of it that is the prefilter cache's bucket fetches (filter_windows' FETCH_SIZE
    beyond the 16 B per word of packed reads it streams): (total, cache_fetches) or None"""
    factor_of = {PMC_KERNELS[st]: f for st, f in FETCH_FACTOR.items() if st in PMC_KERNELS}
    total = cache = 0.0
    for name in PMC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            return None
        for line in open(path).read().splitlines()[1:]:
            r = line.rsplit(",", 3)
            if len(r) != 4 or r[0].startswith("k_synth") or r[0].startswith("k_alloc_probe"):      # (outside the timed steps)
                continue
            f = 1.0
            if "fetch" in name:
                f = next((v for kk, v in factor_of.items() if r[0].startswith(kk)), 1.0)
            b = float(r[2]) * 1024.0 * f
            total += b
            if "fetch" in name and r[0].startswith(PMC_KERNELS["filter_windows"]):
                cache += b
    return total / PMC_RUN_STEPS - sum(random_request_bytes(st, n_pairs_per_step) for st in FETCH_FACTOR), cache / PMC_RUN_STEPS


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=50_000_000, help="read pairs (BASELINE config 2: 50M)")
    ap.add_argument("--genome", type=int, default=64_000_000, help="synthetic transcriptome bases")
    ap.add_argument("--nk", type=int, default=450_000_000, help="expected distinct k-mers (-nk) sizing the filters")
    ap.add_argument("--k", type=int, default=25)
    ap.add_argument("--fpr", type=float, default=0.01)
    ap.add_argument("--err", type=float, default=0.001)
    ap.add_argument("--batch-kmers", type=int, default=0)
    ap.add_argument("--cpu-sample-pairs", type=int, default=6_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="use the sharded engine + RCCL collectives even on 1 GPU")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-resident leg (packed reads in pinned host memory, uploaded inside the timed region)")
    ap.add_argument("--host-piece-reads", type=int, default=0, help="reads per uploaded piece of the host-resident leg (0: 2^20, doubling up to 2^23)")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher around it: become the launcher.  The N ranks are started exactly as the
    driver's own command starts them (torch.distributed.run, one node, 127.0.0.1, a free port); rank 0's JSON line is the
    last line of their common stdout, which is this process's stdout."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        # a launcher's world size is the number of ranks that exist; --gpus must say the same or the line would lie about n_gpus
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N ranks with --gpus N, or plain `python bench.py --gpus N`)" % (a.gpus, world))
    backend = os.environ.get("RB_BENCH_BACKEND", "nccl")     # "gloo": several ranks on ONE GPU (functional check of this script)
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())
    elif local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d, this node shows %d (RB_BENCH_BACKEND=gloo runs several ranks on one GPU as a functional check)"
                         % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    sharded_mode = world > 1 or a.force_sharded
    if sharded_mode:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from rnabloom import _native as N
    from rnabloom.graph import BloomFilterDeBruijnGraph, ReadBatch

    k = a.k
    pairs_total = a.pairs
    dbg_bits = N.lib.rb_expected_size(a.nk, a.fpr, 2)
    cbf_bytes = N.lib.rb_expected_size(a.nk, a.fpr, 2)
    pk_bits = N.lib.rb_expected_size(a.nk, a.fpr, 2)
    dist_pk = max(1, 150 - k - 10)              # R/RNABloom.java:1022 (minNumKmerPairs 10)

    # sharded engine: every rank holds the whole packed read set (replicated input: reads are 1/38 of the
    # bytes of their (hash, occurrence) records) and keeps the k-mers it owns
    batch = ReadBatch.synthetic(pairs_total, a.genome, 150, 300, 30, a.err, 1e-4, 2.0, seed=0x5EED, device=local)
    if not sharded_mode:
        g = BloomFilterDeBruijnGraph(dbg_bits, cbf_bytes, pk_bits, 2, 2, 2, k, False, True, device=local, rngSeed=1,
                                     maxBatchKmers=a.batch_kmers)
        g.setReadPairedKmerDistance(dist_pk)

        def step():
            g.clearAllBf()
            s1 = g.addBatch(batch, storeReadPairedKmers=True, first=0, n=pairs_total)
            s2 = g.addBatch(batch, reverseComplement=True, storeReadPairedKmers=True, first=pairs_total, n=pairs_total)
            return s1, s2
    else:
        # filters sharded by index range over the ranks, a k-mer owned where its first counter lives; reads are
        # data-parallel; RCCL moves records, probes, replies and writes
        from types import SimpleNamespace
        from rnabloom import sharded
        sr = sharded.ShardRank((dbg_bits, cbf_bytes, pk_bits, 2, 2, 2, k, 0, 1, local, 0, 1, a.batch_kmers), rank, world, local)
        sr.set_read_pair_distance(dist_pk)
        pos_bits, rps = sharded.plan(150, k, world, a.batch_kmers or sharded.default_batch_kmers(world, sr.mode))
        g = SimpleNamespace(profileEnable=lambda on: check_(sr, on), profileGet=lambda reset=True: prof_(sr, reset))
        # The exchange driver: by default phases AND exchanges run inside the library (csrc/rb_comm.hip: ncclSend / ncclRecv groups on the
        # library's stream, nothing interpreted between the phases — rb_shard_add_range).  Before the first step every rank runs the
        # communicator's self-test (a small all-to-all and all-gather of known bytes through that transport); if it fails on ANY rank, all
        # ranks fall back to the torch.distributed driver of rnabloom/sharded.py (RB_SHARD_DRIVER=torch selects that one directly).
        # What has run where: the native driver with 1-8 virtual ranks (threads + device copies) and over RCCL at world 1; the torch
        # driver between 2-4 real processes over gloo; neither has met a second physical GPU.
        want_native = os.environ.get("RB_SHARD_DRIVER", "native").lower() != "torch" and backend == "nccl"
        comm, native = None, False
        if want_native:
            ok = 1
            try:
                comm = sharded.NativeComm.rccl(dist, local)
                N.check(N.lib.rb_shard_comm_selftest(comm.h, rank, local, 0))
            except Exception as e:      # noqa: BLE001
                ok = 0
                print("[bench] rank %d: native exchange driver unavailable (%s)" % (rank, e), file=sys.stderr, flush=True)
            flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native = bool(flag.item())
            if not native and rank == 0:
                print("[bench] falling back to the torch.distributed exchange driver", file=sys.stderr, flush=True)

        def step():
            sr.clear()
            out = []
            for first, fl in ((0, N.ADD_STORE_READ_PAIRS), (pairs_total, N.ADD_STORE_READ_PAIRS | N.ADD_REVCOMP)):
                before = dict(sr.stats)
                if native:
                    sharded.add_range_native(sr, comm, batch, first, pairs_total, fl, rps, pos_bits)
                else:
                    sharded.run_distributed(sr.add_range(batch, first, pairs_total, fl, rps, pos_bits))
                out.append(SimpleNamespace(**{kk: sr.stats[kk] - before[kk] for kk in before}))
            return out

    def barrier():
        torch.cuda.synchronize()
        if sharded_mode:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    if sharded_mode and os.environ.get("RB_SHARD_TRACE"):
        sharded.TRACE = {}
        torch.cuda.synchronize()
        sharded._t_last[0] = time.perf_counter()
    g.profileEnable(True)   # HIP events on the library's own stream, around every stage launch
    g.profileGet(reset=True)
    barrier()
    t0 = time.perf_counter()
    kmers = pairs_ins = distinct = conflict = n_sorted = 0
    for _ in range(a.steps):
        s1, s2 = step()
        kmers += s1.kmers + s2.kmers
        pairs_ins += s1.pairs + s2.pairs
        distinct += s1.distinct + s2.distinct
        conflict += s1.conflict_ops + s2.conflict_ops
        n_sorted += getattr(s1, "sorted_kmers", s1.kmers) + getattr(s2, "sorted_kmers", s2.kmers)
    barrier()
    dt = time.perf_counter() - t0
    if sharded_mode:
        rdev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt], device=rdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([kmers], device=rdev, dtype=torch.int64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        kmers_all = int(c.item())
    else:
        kmers_all = kmers
    prof = g.profileGet(reset=True)

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = kmers_all / dt
        # dominant kernel class from the live HIP-event timings
        # (among the stages that have a byte model: on a toy workload a bookkeeping stage of the conflict path can take longest)
        modelled = {n: v for n, v in prof.items() if algorithmic_bytes(n, 1, 1, 1, 1, 1) is not None}
        dom = max(modelled.items(), key=lambda kv: kv[1][0]) if modelled else ("none", (0.0, 0))
        dom_name, (dom_ms, dom_launches) = dom
        roof = None
        words = 2 * pairs_total * 5 * a.steps          # 32-base words this rank walks
        if sharded_mode and sr.mode == "split":
            words //= world
        per_stage, per_stage_gb = {}, {}
        default_cfg = (a.pairs, a.genome, a.nk, a.k, a.batch_kmers, sharded_mode) == (50_000_000, 64_000_000, 450_000_000, 25, 0, False)
        # counter bytes are quoted only beside the code they were counted on: the PMC summaries' csrc id must be this library's
        meta = pmc_meta()
        build_id = (N.lib.rb_build_id() or b"").decode()
        pmc_ok = default_cfg and bool(build_id) and meta.get("csrc_id") == build_id
        for name, (ms, launches) in prof.items():
            ab = algorithmic_bytes(name, kmers, pairs_ins, distinct, words, n_sorted, sharded=sharded_mode, k=k)
            if ab and ms > 0:
                per_stage[name] = round(ab / (ms * 1e-3) / 1e9, 1)
                cb = pmc_step_bytes(name, pairs_ins // a.steps) if pmc_ok else None
                per_stage_gb[name] = {"model": round(ab / a.steps / 1e9, 1), "counters": round(cb / 1e9, 1) if cb else None}
        if dom_launches:
            ab = algorithmic_bytes(dom_name, kmers, pairs_ins, distinct, words, n_sorted, sharded=sharded_mode, k=k)
            if ab:
                achieved = ab / (dom_ms * 1e-3) / 1e9      # = bytes per launch / average launch duration
                traffic = pmc_traffic(dom_name, pairs_ins // a.steps) if pmc_ok else None      # the PMC passes profiled exactly this command on exactly this code
                roof = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "traffic_source": "profiles/%s_pmc_{fetch,write}_size.csv (separate rocprofv3 --pmc passes of this command), bytes per kernel launch, FETCH_SIZE x %.0f (calibration: profiles/%s_pmc_calibration.txt)" % (PMC_TAG, FETCH_FACTOR.get(dom_name, 1.0), PMC_TAG) if traffic else None,
                        "traffic_commit": meta.get("git_head"), "traffic_csrc_id": meta.get("csrc_id"), "build_csrc_id": build_id,
                        "traffic_withheld": None if (pmc_ok or not default_cfg) else "the committed PMC summaries (profiles/%s_pmc_*) were counted on kernel sources %s, this library is %s: traffic, path_frac and the counters column are left out rather than mixed with this run's times" % (PMC_TAG, meta.get("csrc_id", "of an unrecorded tree"), build_id),
                        "note": "dominant stage by HIP-event time on its own stream; achieved = model bytes / measured time, traffic = counters",
                        "algorithmic_bytes_per_launch": int(ab / dom_launches),
                        "avg_launch_ms": round(dom_ms / dom_launches, 3), "launches": dom_launches,
                        "all_stages_model_GBps": per_stage,
                        "all_stages_GB_per_step": per_stage_gb,     # model bytes beside the counters' (committed PMC passes)
                        "upper_bound_models": UPPER_BOUND_MODELS}
                # the PATH's roofline, not only the dominant kernel's: every HBM byte the counters saw in a step (all kernels) over the
                # step's wall time; and how much of that is the prefilter cache deciding to DROP occurrences (bytes this design added)
                pb = pmc_path_bytes(pairs_ins // a.steps) if pmc_ok else None
                if pb:
                    streamed = 16.0 * words / a.steps                 # the packed reads the prefilter walks anyway
                    roof["path_bytes_per_step"] = int(pb[0])
                    roof["path_frac"] = round(pb[0] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    roof["prefilter_cache_share"] = round(max(0.0, pb[1] - streamed) / pb[0], 4)
                    roof["path_note"] = "path_frac = counter bytes of ALL kernels per step (profiles/%s_pmc_*.csv) / this run's ms_per_step / peak" % PMC_TAG
        out = {
            "metric": "k-mers/sec hashed+inserted into Bloom dBG (k=25, 50M 150bp reads)",
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%dM synthetic 150bp paired-end reads, k=%d, Bloom FPR %.2f, nk=%d (configs[1])"
                       % (pairs_total // 1_000_000, k, a.fpr, a.nk),
                       "pairs": pairs_total, "genome_bases": a.genome, "dbgbf_bits": dbg_bits, "cbf_bytes": cbf_bytes,
                       "rpkbf_bits": pk_bits, "kmers_per_step": kmers_all // a.steps,
                       "read_pairs_per_step": pairs_ins // a.steps, "distinct_per_step": distinct // a.steps, "sorted_kmers_per_step": n_sorted // a.steps,
                       "conflict_ops_per_step": conflict // a.steps,
                       "mean_kmer_coverage": round(kmers_all / a.steps / max(1, a.genome), 1),
                       "prefilter_survival": round(n_sorted / max(1, kmers), 4),
                       "note": "throughput depends on the coverage: the no-op prefilter drops occurrences that provably cannot change a counter (here all but the survival fraction); at low coverage or k > 64 the same engine sorts every occurrence; substituted bases carry quality '#' and are masked (SURVEY s8(d)) - with every error passing the threshold (RB_SYNTH_KEEP_ERRORS=1) the step takes 1.3x as long (DESIGN.md s5)",
                       "parallelism": ("single GPU" if not sharded_mode else "filters index-sharded x%d, k-mers owned by first counter (%s mode), %s all-to-all"
                                       % (world, sr.mode, ("RCCL send/recv below the C ABI" if native else "RCCL") if backend == "nccl" else backend))},
            "stages_ms_per_step": {n: round(v[0] / a.steps, 2) for n, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
            "roofline": roof,
        }
        if sharded_mode and sharded.TRACE is not None:
            out["shard_phase_ms_per_step"] = {kk: round(v / a.steps, 1) for kk, v in sorted(sharded.TRACE.items(), key=lambda kv: -kv[1])}
        if not sharded_mode and not a.no_host_leg:
            out["host_resident"] = host_resident_leg(a, g, batch, pairs_total, kmers_all // a.steps)
        if not a.no_cpu_baseline and world == 1:       # the CPU baseline is timed on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(a, batch, dbg_bits, cbf_bytes, pk_bits, dist_pk)
    if sharded_mode:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio; flush it first so the JSON is the LAST line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def host_resident_leg(a, g, batch, pairs_total, kmers_per_step):
    """The metric as SURVEY.md s8(d) words it: input resident in HOST memory in the build's batch format.  Both files of the read set sit
    packed in pinned host memory (12 B per 32 bases + 4 B per read: rb_batch_download_packed); a step starts from cleared filters and every
    byte is uploaded INSIDE the timed region: one rb_graph_add_packed call per file (include/rb_capi.h, csrc/rb_packed.hip) sends the lengths,
    then codes / valid in pieces on a copy stream while the insert pipeline already works on the pieces that have arrived; the second file's
    upload is started with the first one's (rb_graph_prefetch_packed) and travels while the first file is inserted.  Same number of
    timed steps as the HBM-resident figure; `filters_equal_resident` compares the folds of all three filters with the resident leg's (which
    ran last on the same handle)."""
    import torch
    from rnabloom import _native as N
    from rnabloom.graph import PackedStream
    fold = lambda: tuple(_fold(g, w) for w in (N.DBGBF, N.CBF, N.RPKBF))
    ref = fold()
    files = [(batch.downloadPacked(0, pairs_total), False), (batch.downloadPacked(pairs_total, pairs_total), True)]
    nbytes = sum(ph.nbytes() for ph, _ in files)

    def step():
        km = 0
        for ph, _ in files:                   # both uploads are started (in file order, one copy stream): the second file travels while the first is inserted
            g.prefetchPacked(ph, pieceReads=a.host_piece_reads)
        g.clearAllBf()                        # (behind the prefetch calls: the clear waits for its memsets, 3.7 ms the first file's lengths use to get going)
        for ph, rc in files:
            km += g.addPacked(ph, reverseComplement=rc, storeReadPairedKmers=True, pieceReads=a.host_piece_reads).kmers
        return km

    step()                                     # warm-up (the handle's ingest buffers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    km = 0
    marks = [t0]
    for _ in range(a.steps):
        km += step()
        marks.append(time.perf_counter())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if os.environ.get("RB_HOST_TIMING"):
        print("[bench] host-resident steps (ms): " + " ".join("%.1f" % ((marks[i + 1] - marks[i]) * 1e3) for i in range(a.steps)), file=sys.stderr, flush=True)
    equal = fold() == ref
    # the link alone: both files through a packed stream, chunk after chunk, nothing inserted (after the timed steps: it has device buffers and a
    # stream of its own)
    chunk = min(pairs_total, 12_500_000)
    ps = PackedStream(chunk, max(ph.words_before(min(r0 + chunk, ph.n_reads)) - ph.words_before(r0) for ph, _ in files for r0 in range(0, ph.n_reads, chunk)), device=g.device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ph, _ in files:
        for r0 in range(0, ph.n_reads, chunk):
            ps.begin(ph, r0, min(chunk, ph.n_reads - r0)); ps.finish()
    link_s = time.perf_counter() - t0
    ps.close()
    for ph, _ in files:
        ph.close()
    return {"value": km / dt, "unit": "k-mers/s", "ms_per_step": dt / a.steps * 1e3, "steps": a.steps, "kmers_per_step": km // a.steps,
            "host_bytes_per_step": nbytes, "h2d_GBps": round(nbytes / link_s / 1e9, 1), "upload_alone_ms": round(link_s * 1e3, 1),
            "piece_reads": a.host_piece_reads or "2^22 words doubling to 2^25", "filters_equal_resident": bool(equal) and km // a.steps == kmers_per_step,
            "note": "packed reads in pinned host memory (rb_batch_download_packed format), every byte uploaded inside the timed region on a copy "
                    "stream beside the insert (rb_graph_add_packed, one call per file); `value` of the line itself times HBM-resident input"}


def _fold(g, which):
    import ctypes as C
    from rnabloom import _native as N
    v = C.c_uint64()
    N.check(N.lib.rb_filter_fold(g.h, which, C.byref(v)))
    return v.value


def check_(sr, on):
    from rnabloom import _native as N
    N.check(N.lib.rb_graph_profile_enable(sr.h, int(on)))


def prof_(sr, reset):
    import ctypes as C
    from rnabloom import _native as N
    p = N.Profile()
    N.check(N.lib.rb_graph_profile_get(sr.h, C.byref(p), int(reset)))
    return {p.name[i].decode(): (p.ms[i], p.launches[i]) for i in range(p.n)}


def cpu_baseline(a, batch, dbg_bits, cbf_bytes, pk_bits, dist_pk):
    """The oracle (C restatement of the reference's FastqToGraphWorker loop: T threads pulling reads under one lock,
    non-atomic byte RMW) on bounded samples of the SAME reads and filter sizes.  Three timings, every one on
    full-size filters (no cache-resident sweep): T = all useful threads and T = 8 with the reads in memory, and T = best
    from FASTQ text through the reader lock (what `-stage 1` would time)."""
    import numpy as np
    from oracle import rbo
    ncpu = os.cpu_count() or 1
    n = min(a.cpu_sample_pairs, batch.n_reads // 2)
    seq, off = batch.download(0, n)
    og = rbo.Graph(dbg_bits, cbf_bytes, pk_bits, 2, 2, 2, a.k, False, True, 1)
    og.set_read_pair_distance(dist_pk)

    def timed(nreads, threads):
        og.clear()
        t0 = time.perf_counter()
        st = og.add_reads(seq[: off[nreads]], None, off[: nreads + 1], 3, rbo.STORE_READ_PAIRS, threads=threads)
        dt = time.perf_counter() - t0
        return st.kmers / dt, st.kmers, dt

    def median3(fn):
        runs = sorted(fn() for _ in range(3))
        return runs[1], runs

    # the reference's workers serialise on one reader lock, so more threads is not always faster: 1/8 of the sample per candidate
    cands = sorted({t for t in (16, 32, 64, 128, ncpu) if t <= ncpu} | {min(8, ncpu)})
    sweep = {t: timed(max(1, n // 8), t)[0] for t in cands}
    best_t = max(sweep, key=sweep.get)
    # every reported figure is the median of three runs over the SAME reads: the first quarter of the sample
    m = max(1, n // 4)
    (best_rate, best_kmers, best_dt), best_runs = median3(lambda: timed(m, best_t))
    (t8_rate, _, t8_dt), t8_runs = median3(lambda: timed(m, min(8, ncpu)))
    # with parsing: FASTQ text of the same reads (qualities 'I': the synthetic batch carries usable flags, not PHRED)
    L = int(off[1] - off[0])
    rec = np.empty((m, 2 * L + 16), np.uint8)
    rec[:, :10] = np.frombuffer(b"@r" + b"0" * 8, np.uint8); rec[:, 10] = 10
    rec[:, 11:11 + L] = seq[: m * L].reshape(m, L); rec[:, 11 + L] = 10
    rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
    rec[:, 14 + L:14 + 2 * L] = ord("I"); rec[:, 14 + 2 * L] = 10
    text = np.ascontiguousarray(rec[:, :15 + 2 * L]).reshape(-1)

    def parsed():
        og.clear()
        t0 = time.perf_counter()
        stp = og.add_fastq(text, L, 3, rbo.STORE_READ_PAIRS, threads=best_t)
        dt = time.perf_counter() - t0
        return stp.kmers / dt, stp.kmers, dt
    (parse_rate, _, parse_dt), parse_runs = median3(parsed)
    return {"value": best_rate, "unit": "k-mers/s", "cores": best_t, "kind": "port", "host_cpus": ncpu,
            "t8_value": t8_rate, "with_fastq_parsing_value": parse_rate,
            "runs_Mkmers_per_s": {"value": [round(r[0] / 1e6, 2) for r in best_runs], "t8_value": [round(r[0] / 1e6, 2) for r in t8_runs],
                                  "with_fastq_parsing_value": [round(r[0] / 1e6, 2) for r in parse_runs]},
            "thread_sweep_Mkmers_per_s": {t: round(r / 1e6, 2) for t, r in sweep.items()},
            "sample": "first %d left reads of the same synthetic set (%d k-mers + read pairs), the same reads for every figure, each the median of three "
                      "runs: at the best of the probed thread counts (%.1f s a run), at T = 8 (%.1f s), from FASTQ text under the reader lock (%.1f s); the "
                      "thread sweep uses %d reads per candidate, one run each; all on the full-size filters; C restatement of the reference's "
                      "FastqToGraphWorker loop (no JVM in the image)" % (m, best_kmers, best_dt, t8_dt, parse_dt, max(1, n // 8))}


if __name__ == "__main__":
    main()
