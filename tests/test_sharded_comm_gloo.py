"""world_size-2 gloo test (CPU) of the multi-rank exchange layer: run_distributed (torch.distributed
all_to_all / all_gather, what runs over RCCL on the GPUs) must deliver exactly what run_loopback
delivers — and run_loopback + the HIP phase functions are what tests/test_gpu_sharded.py proves
bit-exact against the oracle."""
import os
import pickle
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def toy(rank, G, log):
    """A rank coroutine with the same yield protocol as ShardRank.substep, on CPU tensors."""
    rng = np.random.default_rng(100 + rank)
    n_all = yield ("ints", [5 + rank, rank * 7])
    log.append(("ints", n_all))
    for rnd in range(4):
        counts = [[int(c) for c in rng.integers(0, 50, G)] for _ in range(3)]    # three tensors per exchange
        if rnd == 2:
            counts[0] = [0] * G                                # one empty tensor
            counts[2] = [0] * G
        if rnd == 3:
            for c in counts:
                c[rank] = 0
        send = [torch.from_numpy(rng.integers(0, 256, sum(c), dtype=np.uint8)) for c in counts]
        recv, rc = yield ("a2a", send, counts)
        log.append(("a2a", [t.clone().numpy().tobytes() for t in recv], [list(c) for c in rc]))
        reply = [(t.to(torch.int16) * 3 % 251).to(torch.uint8) for t in recv]    # replies travel the reverse way,
        back, bc = yield ("a2a", reply, rc, counts)                              # receive counts already known
        log.append(("back", [t.clone().numpy().tobytes() for t in back], [list(c) for c in bc]))
        assert [list(c) for c in bc] == counts
    g = torch.from_numpy(rng.integers(0, 256, 16 * rank, dtype=np.uint8))   # rank 0 contributes nothing
    allg, sizes = yield ("gather", g)
    log.append(("gather", allg.clone().numpy().tobytes(), list(sizes)))
    allg, sizes = yield ("gather", torch.zeros(0, dtype=torch.uint8))
    log.append(("gather0", allg.clone().numpy().tobytes(), list(sizes)))
    # several tensors in one gather (conflict edges + cache updates travel together)
    ga = torch.from_numpy(rng.integers(0, 256, 5 + 9 * rank, dtype=np.uint8))
    gb = torch.from_numpy(rng.integers(0, 256, 24 * (1 - rank % 2), dtype=np.uint8))
    (a_all, a_sz), (b_all, b_sz) = yield ("gather", [ga, gb])
    log.append(("gather2", a_all.clone().numpy().tobytes() + b_all.clone().numpy().tobytes(), list(a_sz) + list(b_sz)))


def _worker(rank, world, port, outdir, chunk, fuse):
    sys.path.insert(0, os.path.join(ROOT, "rna-bloom_amd"))
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from rnabloom import sharded
    from rnabloom.sharded import run_distributed
    if chunk:
        sharded.A2A_CHUNK = chunk            # force the multi-round path (RCCL's 1 GiB per-peer limit)
    if fuse is not None:
        sharded.FUSE_LIMIT = fuse            # 0: every tensor of a phase travels in its own collective
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    run_distributed(toy(rank, world, log))
    dist.barrier()
    dist.destroy_process_group()
    with open(os.path.join(outdir, "r%d.pkl" % rank), "wb") as fh:
        pickle.dump(log, fh)


@pytest.mark.parametrize("world,chunk,fuse", [(2, 0, None), (2, 13, None), (2, 0, 0), (2, 7, 0)])
def test_run_distributed_equals_loopback(world, chunk, fuse):
    from rnabloom.sharded import run_loopback
    logs = [[] for _ in range(world)]
    run_loopback([toy(r, world, logs[r]) for r in range(world)])
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, 29500 + (os.getpid() + 7 * chunk + (3 if fuse == 0 else 0)) % 400, d, chunk, fuse), nprocs=world, join=True)
        for r in range(world):
            with open(os.path.join(d, "r%d.pkl" % r), "rb") as fh:
                got = pickle.load(fh)
            assert len(got) == len(logs[r])
            for a, b in zip(got, logs[r]):
                assert a[0] == b[0]
                if a[0] == "ints":
                    assert [list(x) for x in a[1]] == [list(x) for x in b[1]]
                else:
                    assert a[1:] == b[1:], (r, a[0])


def test_plan_limits():
    from rnabloom.sharded import plan
    from rnabloom.sharded import default_batch_kmers
    pos_bits, reads = plan(150, 25, 8, 1 << 30)
    assert (1 << pos_bits) > 150 - 25 and reads < (1 << (32 - pos_bits)) and reads * 150 <= (1 << 30)
    pos_bits, reads = plan(100_000, 35, 2, 1 << 28)
    assert (1 << pos_bits) > 100_000 - 35 and reads >= 1
    pos_bits, reads = plan(150, 25, 8, default_batch_kmers(8, "split"))       # 2^32 windows
    assert reads == (1 << 32) // 150 and pos_bits == 7 and reads < (1 << (32 - pos_bits))
    assert default_batch_kmers(8, "replicated") == 1 << 30 and default_batch_kmers(2, "split") == 1 << 31 and default_batch_kmers(4, "split") == 1 << 32
