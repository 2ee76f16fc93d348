"""BASELINE config [3] ("200M 150bp paired-end, pooled, 8 x MI355X — Bloom > HBM of one GPU") as far as ONE GPU can go:
one shard of the 8-way graph at the config's filter sizes (nk = 14 G: dbgbf / rpkbf 265.75 G bits = 33.2 GB, cbf 265.75 GB;
a rank holds 1/8: 4.15 GB + 33.2 GB + 4.15 GB), driven through the owner-side phases of the sharded engine with requests
at the ends of its index range and around the 2^32 / 2^35 boundaries.  What breaks if an index is truncated to 32 bits,
a span is not a multiple of 64, a byte offset is computed in 32 bits or a population count wraps: exactly these checks."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NK, FPR, K, COUNT = 14_000_000_000, 0.01, 25, 8


@pytest.mark.parametrize("rank", [7, 3])
def test_one_rank_of_the_config4_graph(rank):
    import torch
    from rnabloom import _native as N
    from rnabloom._native import check, lib
    size = int(lib.rb_expected_size(NK, FPR, 2))
    assert size > 265_000_000_000
    p = N.GraphParams(size, size, size, 2, 2, 2, K, 0, 1, 0, 0, 1, 0)
    h = C.c_void_p()
    check(lib.rb_graph_create_shard(C.byref(p), rank, COUNT, C.byref(h)))
    try:
        dev = torch.device("cuda", 0)
        span_want = ((size + COUNT - 1) // COUNT + 63) // 64 * 64
        spans = {}
        for which in (N.DBGBF, N.CBF, N.RPKBF):
            sp, lo, hi = C.c_int64(), C.c_int64(), C.c_int64()
            check(lib.rb_shard_span(h, which, C.byref(sp), C.byref(lo), C.byref(hi)))
            assert sp.value == span_want and lo.value == rank * span_want and hi.value == min(size, (rank + 1) * span_want)
            assert lo.value % 64 == 0 and hi.value - lo.value > (1 << 34)          # > 2^32 indices per rank, > 2^32 words of counters
            spans[which] = (lo.value, hi.value)
        lo, hi = spans[N.CBF]
        assert hi > (1 << 37) if rank == 7 else lo > (1 << 36)                      # global indices need > 32 (37-38) bits
        rng = np.random.default_rng(100 + rank)
        edge = np.array([lo, lo + 1, lo + 31, lo + 32, lo + 63, lo + 64, lo + (1 << 32) - 1, lo + (1 << 32), lo + (1 << 32) + 1,
                         lo + (1 << 33) + 7, lo + (1 << 34) + 5, hi - 65, hi - 64, hi - 33, hi - 2, hi - 1], np.uint64)
        rnd = (lo + rng.integers(0, hi - lo, 200_000)).astype(np.uint64)
        idx = np.unique(np.concatenate([edge, rnd]))
        not_set = np.setdiff1d(np.unique(np.concatenate([edge + np.uint64(2), (lo + rng.integers(0, hi - lo, 50_000)).astype(np.uint64)])), idx)
        not_set = not_set[(not_set >= lo) & (not_set < hi)]
        t = lambda a, dt=torch.int64: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(dev)   # noqa: E731
        ptr = lambda x: C.c_void_p(x.data_ptr())                                                                  # noqa: E731
        d_idx = t(idx); d_probe = t((np.arange(idx.size, dtype=np.uint64) << np.uint64(4)))
        # ---- Bloom bits: every request finds its bit clear and is its first setter (reply 2); a second round finds them set (1)
        drep = torch.zeros(idx.size, dtype=torch.uint8, device=dev); crep = torch.zeros(idx.size, dtype=torch.uint8, device=dev)
        check(lib.rb_shard_serve(h, 0, ptr(d_idx), ptr(d_probe), idx.size, ptr(d_idx), idx.size, ptr(d_idx), idx.size, ptr(drep), ptr(crep)))
        assert bool((drep == 2).all()), "a first request did not find its bit clear"
        assert bool((crep == 0).all()), "fresh counters must read 0 and be unclaimed"
        drep2 = torch.zeros_like(drep); crep2 = torch.zeros_like(crep)
        check(lib.rb_shard_serve(h, 0, ptr(d_idx), ptr(d_probe), idx.size, ptr(d_idx), idx.size, None, 0, ptr(drep2), ptr(crep2)))
        assert bool((drep2 == 1).all()), "bits set by the first round must be found set"
        assert bool((crep2 == 0x80).all()), "counters claimed in the first round carry the claim mark (and still count 0)"
        # ---- counter writes: distinct values per index, then read back through the query phase
        vals = ((idx % np.uint64(127)) + np.uint64(1)).astype(np.uint8)             # 1..127
        d_val = torch.from_numpy(vals).to(dev)
        check(lib.rb_shard_apply_writes(h, ptr(d_idx), ptr(d_val), idx.size))
        d_no = t(not_set)
        for which in (N.DBGBF, N.RPKBF):
            brep = torch.zeros(idx.size, dtype=torch.uint8, device=dev); cq = torch.zeros(idx.size, dtype=torch.uint8, device=dev)
            check(lib.rb_shard_query_serve(h, which, ptr(d_idx), idx.size, ptr(d_idx), idx.size, ptr(brep), ptr(cq)))
            assert bool((brep == 1).all()), "filter %d lost bits" % which
            assert bool((cq.cpu().numpy() == vals).all()), "a counter byte did not come back"
            b0 = torch.zeros(not_set.size, dtype=torch.uint8, device=dev); c0 = torch.zeros(not_set.size, dtype=torch.uint8, device=dev)
            check(lib.rb_shard_query_serve(h, which, ptr(d_no), not_set.size, ptr(d_no), not_set.size, ptr(b0), ptr(c0)))
            assert int(b0.sum()) == 0 and int(c0.sum()) == 0, "an index that was never touched is set: some offset aliased"
        # ---- population counts over > 2^32 words: exactly the distinct indices, in every filter
        for which in (N.DBGBF, N.CBF, N.RPKBF):
            n = C.c_int64()
            check(lib.rb_filter_popcount(h, which, C.byref(n)))
            assert n.value == idx.size, (which, n.value, idx.size)
        # ---- the first and last bytes of the local bit range are where the exported concatenation would put them
        nb = C.c_int64(); sz = C.c_int64(); nh = C.c_int()
        check(lib.rb_filter_size(h, N.DBGBF, C.byref(sz), C.byref(nb), C.byref(nh)))
        assert sz.value == size and nb.value == (spans[N.DBGBF][1] - spans[N.DBGBF][0] + 7) // 8
    finally:
        lib.rb_graph_destroy(h)
