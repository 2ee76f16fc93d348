"""The sharded engine as it really runs: one PROCESS per rank, torch.distributed collectives between them.
Two (or four) processes share the single GPU of the test box; the byte exchange goes through gloo (RCCL refuses
two ranks on one device), staged through host memory by the same driver code that runs over RCCL.  Every rank
holds the same read batch; the concatenated per-rank filter ranges must equal the sequential oracle's filters."""
import os
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = (300_007, 450_001, 50_021)


def _reads():
    from rnabloom import synth
    return synth.generate_pairs(1800, G=5000, err=0.002, n_rate=1e-3, seed=29, uniform_expr=True)


def _seeds(d):
    from rnabloom import synth
    s, off = synth.flat(d["left"])
    rng = np.random.default_rng(3)
    return [bytes(s[off[r] + p: off[r] + p + 25]) for r, p in zip(rng.integers(0, 1800, 90), rng.integers(0, 120, 90))]


def _rank_main(rank, world, port, outdir, mode):
    for p in (ROOT, os.path.join(ROOT, "rna-bloom_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    torch.cuda.init()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rnabloom import _native as N
    from rnabloom import sharded, synth
    from rnabloom.graph import ReadBatch
    d = _reads()
    sr = sharded.ShardRank((*SIZES, 2, 2, 2, 25, 0, 1, 0, 0, 5, 0), rank, world, 0, mode)
    sr.set_read_pair_distance(115)
    pos_bits, _ = sharded.plan(150, 25, world)
    for name, rc in (("left", False), ("right", True)):
        s, off = synth.flat(d[name]); q, _ = synth.flat(d[name[0] + "qual"])
        batch = ReadBatch.from_ascii(s, q, off, 3)
        flags = N.ADD_STORE_READ_PAIRS | (N.ADD_REVCOMP if rc else 0)
        sharded.run_distributed(sr.add_range(batch, 0, batch.n_reads, flags, 400, pos_bits))
    for which, tag in ((N.DBGBF, "dbg"), (N.CBF, "cbf"), (N.RPKBF, "rpk")):
        np.save(os.path.join(outdir, "%s%d.npy" % (tag, rank)), sr.local_filter(which))
    np.save(os.path.join(outdir, "stats%d.npy" % rank), np.array([sr.stats["kmers"], sr.stats["conflict_ops"], sr.stats["sorted_kmers"]]))
    # maximum-coverage walks on the sharded graph: every rank walks its own slice of the seeds (rank 0: none)
    seeds = _seeds(d)
    mine = seeds[len(seeds) * rank // world: len(seeds) * (rank + 1) // world] if rank else []
    res = []
    sharded.run_distributed(sr.walk(mine, 0, 30, 2.0, res))
    bases, counts, ln, reason = res[0]
    np.save(os.path.join(outdir, "walk%d.npy" % rank), np.concatenate([bases, ln[:, None].astype(np.uint8), reason[:, None]], axis=1))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "replicated"), (2, "split"), (4, "split")])
def test_multiprocess_ranks_match_oracle(world, mode):
    import torch.multiprocessing as mp
    from oracle import rbo
    from rnabloom import synth
    d = _reads()
    og = rbo.Graph(*SIZES, 2, 2, 2, 25, False, True, 5)
    og.set_read_pair_distance(115)
    total = 0
    for name, rc in (("left", False), ("right", True)):
        s, off = synth.flat(d[name]); q, _ = synth.flat(d[name[0] + "qual"])
        st = og.add_reads(s, q, off, 3, rbo.STORE_READ_PAIRS | (rbo.REVCOMP if rc else 0))
        total += st.kmers
    with tempfile.TemporaryDirectory() as out:
        port = 29900 + (os.getpid() * 7 + world * 3 + len(mode)) % 90
        mp.spawn(_rank_main, args=(world, port, out, mode), nprocs=world, join=True)
        for tag, ref in (("dbg", og.dbgbf_bytes()), ("cbf", og.cbf_bytes()), ("rpk", og.rpkbf_bytes())):
            got = np.concatenate([np.load(os.path.join(out, "%s%d.npy" % (tag, r))) for r in range(world)])
            bad = np.nonzero(got != ref)[0]
            assert bad.size == 0, "%s differs at %d bytes: %s" % (tag, bad.size, bad[:6])
        stats = np.sum([np.load(os.path.join(out, "stats%d.npy" % r)) for r in range(world)], axis=0)
        assert stats[0] == total and stats[1] > 0 and stats[2] < stats[0]
        seeds = _seeds(d)
        for r in range(1, world):
            w = np.load(os.path.join(out, "walk%d.npy" % r))
            mine = seeds[len(seeds) * r // world: len(seeds) * (r + 1) // world]
            assert w.shape[0] == len(mine)
            for i, sd in enumerate(mine):
                eb, ec, er = rbo.walk_max_cov(og, sd, 0, 30, 2.0)
                assert (int(w[i, 30]), int(w[i, 31])) == (len(eb), er) and bytes(w[i, :len(eb)]) == eb
        assert np.load(os.path.join(out, "walk0.npy")).shape[0] == 0
    assert og.cbf_bytes().max() > 24
