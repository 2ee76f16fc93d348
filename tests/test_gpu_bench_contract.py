"""bench.py's one-line contract on a small workload: the single-GPU line (with roofline and cpu_baseline objects) and the line
of a 2-rank run launched the way the driver launches it (torch.distributed.run; the two ranks share the test box's GPU, so the
exchange goes over gloo — RCCL refuses two ranks on one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--pairs", "300000", "--genome", "500000", "--nk", "3000000", "--steps", "1", "--warmup", "1"]


def last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_single_gpu_line_and_two_rank_line():
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "bench.py", "--cpu-sample-pairs", "40000", "--host-piece-reads", "70000"] + SMALL,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = last_json(r.stdout)
    assert r.stdout.strip().splitlines()[-1].startswith("{"), "the JSON line must be the last line"
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(j[key], typ), key
    assert j["n_gpus"] == 1 and j["steps"] == 1 and j["warmup"] == 1 and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert j["unit"] == "k-mers/s" and j["value"] > 0 and "workload" in j["config"] and "model" not in j["config"]
    roof = j["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and roof["achieved"] > 0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and "traffic" in roof
    cpu = j["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["unit"] == "k-mers/s" and cpu["cores"] >= 1 and cpu["sample"]
    # the metric as SURVEY s8(d) words it — input packed in (pinned) host memory, uploaded inside the timed region: same k-mers, same filters
    hr = j["host_resident"]
    assert hr["unit"] == "k-mers/s" and hr["value"] > 0 and hr["ms_per_step"] > 0 and hr["h2d_GBps"] > 0 and hr["steps"] == 1
    assert hr["kmers_per_step"] == j["config"]["kmers_per_step"] and hr["filters_equal_resident"] is True
    assert hr["piece_reads"] == 70000 and hr["host_bytes_per_step"] == 600000 * (5 * 12 + 4)
    # two ranks, launched like the driver does it
    env["RB_BENCH_BACKEND"] = "gloo"
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--no-cpu-baseline"] + SMALL,
                        cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    j2 = last_json(r2.stdout)
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong" and j2["value"] > 0
    assert j2["config"]["kmers_per_step"] == j["config"]["kmers_per_step"], "both engines insert the same k-mers"
    assert "cpu_baseline" not in j2
    # the plain form: `python bench.py --gpus 2` with no launcher around it starts its two ranks itself and prints the same line
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r3 = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--no-cpu-baseline"] + SMALL, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r3.returncode == 0, r3.stderr[-3000:]
    j3 = last_json(r3.stdout)
    assert r3.stdout.strip().splitlines()[-1].startswith("{"), "the JSON line must be the last line"
    assert j3["n_gpus"] == 2 and j3["scaling"] == "strong" and j3["config"]["kmers_per_step"] == j["config"]["kmers_per_step"]


def test_force_sharded_line_goes_through_the_native_rccl_driver_by_default():
    """bench.py --force-sharded on one GPU: the sharded engine over RCCL at world 1 — since round 4 through the exchange driver below the
    C ABI (rb_shard_add_range, ncclSend / ncclRecv groups) after the communicator's self-test; RB_SHARD_DRIVER=torch is the other driver.
    Both lines insert the k-mers the single-GPU line inserts."""
    base = None
    for drv in ("native", "torch"):
        env = dict(os.environ)
        env["RB_SHARD_DRIVER"] = drv
        env["MASTER_PORT"] = str(free_port())
        r = subprocess.run([sys.executable, "bench.py", "--force-sharded", "--no-cpu-baseline"] + SMALL, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        j = last_json(r.stdout)
        assert j["n_gpus"] == 1 and j["value"] > 0
        assert ("below the C ABI" in j["config"]["parallelism"]) == (drv == "native"), j["config"]["parallelism"]
        assert "falling back" not in r.stderr
        base = base or j["config"]["kmers_per_step"]
        assert j["config"]["kmers_per_step"] == base


def test_communicator_self_test_on_a_loopback_hub():
    """rb_shard_comm_selftest with 4 virtual ranks (threads + device copies): an all-to-all and an all-gather of known bytes, messages of
    a few KB and of 3 MB"""
    import ctypes as C
    import threading
    sys.path[:0] = [ROOT, os.path.join(ROOT, "rna-bloom_amd")]
    from rnabloom import _native as N
    from rnabloom import sharded
    for big in (0, 3_000_000):
        comm = sharded.NativeComm.loopback(4)
        rc = [None] * 4

        def work(i):
            rc[i] = N.lib.rb_shard_comm_selftest(comm.h, i, 0, big)
        th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        for t in th: t.start()
        for t in th: t.join()
        assert rc == [0, 0, 0, 0], (rc, N.lib.rb_last_error())
        comm.destroy()


def test_a_rank_that_fails_fast_and_comes_back_does_not_unpoison_the_hub_for_its_peers():
    """loopback hub, two ranks: rank 0 fails its first collective call at once (bad argument) and re-enters before rank 1 has arrived.  The
    hub must stay poisoned until rank 1 has been through BOTH calls (a hub reset after rank 0's second entry — two entries, world two —
    left rank 1 waiting at a barrier for ever); then the ranks are in step again and the third call succeeds on both."""
    import threading
    sys.path[:0] = [ROOT, os.path.join(ROOT, "rna-bloom_amd")]
    from rnabloom import _native as N
    from rnabloom import sharded
    comm = sharded.NativeComm.loopback(2)
    st = lambda rank, big: N.lib.rb_shard_comm_selftest(comm.h, rank, 0, big)
    assert st(0, -1) != 0                      # call 1 of rank 0 fails
    assert st(0, 0) != 0                       # call 2 of rank 0: the hub is poisoned, fails at once instead of waiting for rank 1
    rc = []
    t = threading.Thread(target=lambda: rc.extend([st(1, 0), st(1, 0)]))     # calls 1 and 2 of rank 1: both must fail, neither may hang
    t.start(); t.join(60)
    assert not t.is_alive(), "rank 1 hangs at a barrier of a hub that was reset too early"
    assert rc[0] != 0 and rc[1] != 0
    res = [None, None]
    th = [threading.Thread(target=lambda i=i: res.__setitem__(i, st(i, 0))) for i in range(2)]      # call 3: in step again
    for x in th: x.start()
    for x in th: x.join(60)
    assert not any(x.is_alive() for x in th) and res == [0, 0], (res, N.lib.rb_last_error())
    comm.destroy()


def test_a_rank_that_fails_after_its_last_barrier_does_not_strand_a_peer_that_waits_in_the_next_call():
    """loopback hub, two ranks: rank 0 fails call 1 AFTER the last barrier of that call (injected: big_bytes = -2), when rank 1 has left call 1
    and already waits in the first barrier of call 2.  Rank 0's exit from call 1 must not clean the hub (everybody's `left` is >= the failed
    call, but rank 1 is INSIDE a call): rank 1 would find `failed` false and its arrival erased, and wait for ever.  Rank 1's call 2 fails,
    rank 0's call 2 fails at once, call 3 succeeds on both."""
    import threading
    sys.path[:0] = [ROOT, os.path.join(ROOT, "rna-bloom_amd")]
    from rnabloom import _native as N
    from rnabloom import sharded
    comm = sharded.NativeComm.loopback(2)
    st = lambda rank, big: N.lib.rb_shard_comm_selftest(comm.h, rank, 0, big)
    r0, r1 = [], []
    t0 = threading.Thread(target=lambda: r0.append(st(0, -2)))
    t1 = threading.Thread(target=lambda: r1.extend([st(1, 0), st(1, 0)]))
    t0.start(); t1.start()
    t0.join(60); t1.join(60)
    assert not t0.is_alive() and not t1.is_alive(), "a rank hangs at a barrier of a hub that was cleaned while it waited"
    assert r0[0] != 0 and r1[0] == 0 and r1[1] != 0, (r0, r1)
    assert st(0, 0) != 0                       # call 2 of rank 0: poisoned, fails at once
    res = [None, None]
    th = [threading.Thread(target=lambda i=i: res.__setitem__(i, st(i, 0))) for i in range(2)]      # call 3: in step again
    for x in th: x.start()
    for x in th: x.join(60)
    assert not any(x.is_alive() for x in th) and res == [0, 0], (res, N.lib.rb_last_error())
    comm.destroy()
