"""rnabloom.sharded — multi-GPU stage-1 insert: filters split by index range across ranks.

One rank = one process = one GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI), or, for
single-GPU testing, G virtual ranks inside one process (`LoopbackCluster`).  The per-rank work is
the phase functions of the C ABI (`rb_shard_*`, csrc/rb_shard.hip); this module only moves byte
buffers between ranks (all_to_all / all_gather) in the order the protocol requires.  Results are
identical to the single-GPU path and to the sequential oracle (tests/test_gpu_sharded.py).

Read distribution: replicated.  Every rank holds the same packed read batch (2 bits per base — 1/38 of
the bytes of the (hash, occurrence) records it expands to, so the reads are what a loader broadcasts)
and walks all of it, keeping only the windows whose k-mer it owns; read-paired k-mers are walked by
1/G of the reads per rank.  Insertion order = read order, exactly as on one GPU.
"""
import ctypes as C

import numpy as np
import torch

if torch.cuda.is_available():
    torch.cuda.init()   # torch's bundled HIP runtime must be initialised before librb_hip.so creates its context

from . import _native as N
from ._native import check, lib


# phase tracer: set TRACE = {} to accumulate synchronous wall time (ms) per protocol phase, all ranks of
# this process together (tools/loopback_bench.py, RB_SHARD_TRACE=1 in bench.py)
TRACE = None
_t_last = [0.0]
import os as _os
_DEBUG = bool(_os.environ.get("RB_SHARD_DEBUG"))
# look-ahead hashing of the next sub-batch: 0 off, 1 begin after this sub-batch's cache updates (resolve),
# 2 begin right after this sub-batch's own hashing (maximum overlap, prefilter cache one sub-batch staler)
_OVERLAP = int(_os.environ.get("RB_SHARD_OVERLAP", "1"))
_RAMP = not _os.environ.get("RB_NO_RAMP")
_MODE = _os.environ.get("RB_SHARD_MODE")                  # "split" | "replicated" | unset (by rank count)
_COPY_SLOTS = bool(_os.environ.get("RB_SHARD_COPY"))     # exchange from torch-owned copies instead of zero-copy views


def trace_mark(what):
    if _DEBUG:
        import sys
        print("[shard] after", what, file=sys.stderr, flush=True)
    if TRACE is None:
        return
    import time
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    now = time.perf_counter()
    TRACE[what] = TRACE.get(what, 0.0) + (now - _t_last[0]) * 1e3
    _t_last[0] = now


class _DevView:
    """CUDA-array-interface holder: lets torch wrap device memory owned by the library without a copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = dict(shape=(int(nbytes),), typestr="|u1", data=(int(ptr), False), version=2, strides=None)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None


class ShardRank:
    """One rank's shard of the graph + the coroutine that drives one global sub-batch."""

    def __init__(self, params, rank, count, device, mode=None):
        self.rank, self.count, self.device = rank, count, device
        self.tdev = torch.device("cuda", device)
        p = N.GraphParams(*params)
        p.device = device
        self.p = p
        self.h = C.c_void_p()
        check(lib.rb_graph_create_shard(C.byref(p), rank, count, C.byref(self.h)))
        self.ordinal = 0
        self.stats = dict(kmers=0, pairs=0, distinct=0, conflict_ops=0, reads=0, sorted_kmers=0)
        # "split": every rank hashes its slice of the reads and sends the surviving records to the k-mer owners
        # (hashing 1/G per rank, needs every link); "replicated": every rank hashes all reads and keeps its own
        # k-mers (no record exchange — better when two ranks share ONE xGMI link)
        self.mode = mode or default_mode(count)
        assert self.mode in ("split", "replicated")
        check(lib.rb_shard_set_cache_replication(self.h, int(self.mode == "split")))

    def destroy(self):
        if self.h:
            lib.rb_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def set_read_pair_distance(self, d):
        check(lib.rb_graph_set_read_paired_kmer_distance(self.h, d))

    def clear(self):
        check(lib.rb_graph_clear(self.h, 15))
        self.ordinal = 0

    def _slot(self, slot, nbytes=None):
        """Zero-copy uint8 view of a library slot (valid until the phase that fills it runs again)."""
        p, nb = C.c_void_p(), C.c_int64()
        check(lib.rb_shard_slot(self.h, slot, C.byref(p), C.byref(nb)))
        if nbytes is not None and nb.value != int(nbytes):
            raise RuntimeError("slot %d holds %d bytes, expected %d" % (slot, nb.value, nbytes))
        if not nb.value:
            return torch.empty(0, dtype=torch.uint8, device=self.tdev)
        if _COPY_SLOTS:
            return self._take(slot, nb.value)
        return torch.as_tensor(_DevView(p.value, nb.value), device=self.tdev)

    def _take(self, slot, nbytes):
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.tdev)
        check(lib.rb_shard_take(self.h, slot, _ptr(t), int(nbytes)))
        return t

    def local_filter(self, which):
        s, nb, _ = C.c_int64(), C.c_int64(), C.c_int()
        check(lib.rb_filter_size(self.h, which, C.byref(s), C.byref(nb), C.byref(_)))
        out = np.zeros(nb.value, np.uint8)
        check(lib.rb_filter_export(self.h, which, out.ctypes.data_as(C.c_void_p), nb.value))
        return out

    def local_popcount(self, which):
        v = C.c_int64()
        check(lib.rb_filter_popcount(self.h, which, C.byref(v)))
        return v.value

    def local_fold(self, which):
        v = C.c_uint64()
        check(lib.rb_filter_fold(self.h, which, C.byref(v)))
        return v.value

    # ---- one global sub-batch; yields exchange requests, receives their results ----
    def substep(self, batch, first, n, pos_bits, flags, nxt=None):
        """reads [first, first+n) of `batch` (every rank holds the same batch and passes the same range);
        nxt = (first, n) of the following sub-batch, whose window hashing is enqueued ahead on the
        library's producer stream so that it overlaps this sub-batch's filter phases and exchanges"""
        mark = trace_mark
        G = self.count
        mode = N.MODE_COUNT_IF_PRESENT if (flags & N.ADD_COUNT_IF_PRESENT) else N.MODE_ADD
        cnt = lambda: (C.c_int64 * G)()
        st = N.AddStats()
        # hash + group: the windows whose k-mer this rank owns -> runs -> requests by filter owner;
        # read pairs of this rank's slice of the reads -> probes by rpkbf owner
        p0, p1 = first + n * self.rank // G, first + n * (self.rank + 1) // G
        d_c, c_c, pair_c = cnt(), cnt(), cnt()
        split = self.mode == "split"
        if split:
            # hash own slice (prefilter against the replicated cache) -> records to the k-mer owners
            rec_c = cnt()
            check(lib.rb_shard_hash(self.h, batch.h, first, n, p0, p1 - p0, self.ordinal, pos_bits, flags, rec_c, pair_c, C.byref(st)))
            rec_c, pair_c = list(rec_c), list(pair_c)
            mark("hash")
            send = [self._slot(N.SLOT_REC_KEYS, 8 * sum(rec_c)), self._slot(N.SLOT_REC_OCC, 4 * sum(rec_c)), self._slot(N.SLOT_PAIR_IDX, 8 * sum(pair_c))]
            (rkeys, rocc, rpidx), (rk_c, _, rp_c) = yield ("a2a", send, [[8 * c for c in rec_c], [4 * c for c in rec_c], [8 * c for c in pair_c]])
            check(lib.rb_shard_group(self.h, _ptr(rkeys), _ptr(rocc), sum(rk_c) // 8, self.ordinal, pos_bits, flags, d_c, c_c))
            d_c, c_c = list(d_c), list(c_c)
            mark("group")
            send = [self._slot(N.SLOT_DREQ_IDX, 8 * sum(d_c)), self._slot(N.SLOT_DREQ_PROBE, 8 * sum(d_c)), self._slot(N.SLOT_CREQ_IDX, 8 * sum(c_c))]
            (o_didx, o_dprobe, o_cidx), (o_dc, _, o_cc) = yield ("a2a", send, [[8 * c for c in d_c], [8 * c for c in d_c], [8 * c for c in c_c]])
        else:
            check(lib.rb_shard_hash_group(self.h, batch.h, first, n, p0, p1 - p0, self.ordinal, pos_bits, flags, d_c, c_c, pair_c, C.byref(st)))
            d_c, c_c, pair_c = list(d_c), list(c_c), list(pair_c)
            if nxt and _OVERLAP == 2:
                check(lib.rb_shard_hash_begin(self.h, batch.h, nxt[0], nxt[1], self.ordinal + int(n), pos_bits, flags))
            mark("hash_group")
            send = [self._slot(N.SLOT_DREQ_IDX, 8 * sum(d_c)), self._slot(N.SLOT_DREQ_PROBE, 8 * sum(d_c)), self._slot(N.SLOT_CREQ_IDX, 8 * sum(c_c)),
                    self._slot(N.SLOT_PAIR_IDX, 8 * sum(pair_c))]
            (o_didx, o_dprobe, o_cidx, rpidx), (o_dc, _, o_cc, rp_c) = yield ("a2a", send, [[8 * c for c in d_c], [8 * c for c in d_c], [8 * c for c in c_c],
                                                                                           [8 * c for c in pair_c]])
        # serve: this rank's filter ranges answer
        nd, nc, np_ = sum(o_dc) // 8, sum(o_cc) // 8, sum(rp_c) // 8
        dreply = torch.empty(nd, dtype=torch.uint8, device=self.tdev)
        creply = torch.empty(nc, dtype=torch.uint8, device=self.tdev)
        check(lib.rb_shard_serve(self.h, mode, _ptr(o_didx), _ptr(o_dprobe), nd, _ptr(o_cidx), nc, _ptr(rpidx), np_,
                                 _ptr(dreply), _ptr(creply)))
        if nxt and _OVERLAP == 2 and not split:
            check(lib.rb_shard_hash_emit(self.h))
        mark("serve")
        (my_dreply, my_creply), _ = yield ("a2a", [dreply, creply], [[c // 8 for c in o_dc], [c // 8 for c in o_cc]], [d_c, c_c])
        # resolve: runs that own their counters alone finish here
        w_c, ord_c, nconf, nedge = cnt(), cnt(), C.c_int64(), C.c_int64()
        check(lib.rb_shard_resolve(self.h, mode, _ptr(my_dreply), _ptr(my_creply), w_c, ord_c, C.byref(st)))
        w_c, ord_c = list(w_c), list(ord_c)
        if nxt and _OVERLAP == 1 and not split:   # after the cache updates of this sub-batch: fresher prefilter, less overlap
            check(lib.rb_shard_hash_begin(self.h, batch.h, nxt[0], nxt[1], self.ordinal + int(n), pos_bits, flags))
        mark("resolve")
        # counter writes of the finished runs, and — same exchange — the contested counters of the runs that can reach one: who else can?
        (o_widx, o_wval, o_ord), (o_wc, _, o_oc) = yield ("a2a", [self._slot(N.SLOT_W_IDX, 8 * sum(w_c)), self._slot(N.SLOT_W_VAL, sum(w_c)),
                                                                  self._slot(N.SLOT_ORD_IDX, 8 * sum(ord_c))], [[8 * c for c in w_c], w_c, [8 * c for c in ord_c]])
        check(lib.rb_shard_apply_writes(self.h, _ptr(o_widx), _ptr(o_wval), sum(o_wc) // 8))
        n_ord = sum(o_oc) // 8
        oreply = torch.empty(n_ord, dtype=torch.uint8, device=self.tdev)
        check(lib.rb_shard_order_serve(self.h, _ptr(o_ord), n_ord, _ptr(oreply)))
        if nxt and _OVERLAP == 1 and not split:
            check(lib.rb_shard_hash_emit(self.h))
        mark("writes")
        (my_oreply,), _ = yield ("a2a", [oreply], [[c // 8 for c in o_oc]], [ord_c])
        # the ordered set: runs in it keep their edges, the others are finished here (their remote writes ride behind the edges)
        check(lib.rb_shard_order_finish(self.h, mode, _ptr(my_oreply), C.byref(nconf), C.byref(nedge), C.byref(st)))
        mark("order")
        # one all_gather: the (run, contested counter) edges of the runs that share a counter and, in split mode, what
        # the owners learnt about their k-mers' counters (for every rank's prefilter cache replica)
        if split:
            (all_edges, e_sizes), (upd, u_sizes) = yield ("gather", [self._slot(N.SLOT_CONF_EDGES, 16 * nedge.value), self._slot(N.SLOT_CACHE_UPD)])
            if sum(e_sizes): check(lib.rb_shard_apply_tagged(self.h, _ptr(all_edges), sum(e_sizes) // 16))
            # every rank's updates but this rank's own (its replica got them when they were made, in rb_shard_resolve)
            before, mine = sum(u_sizes[:self.rank]), u_sizes[self.rank]
            if before: check(lib.rb_shard_cache_apply(self.h, _ptr(upd[:before]), before // 16))
            if sum(u_sizes) - before - mine: check(lib.rb_shard_cache_apply(self.h, _ptr(upd[before + mine:]), (sum(u_sizes) - before - mine) // 16))
            if nxt and _OVERLAP >= 1:      # look-ahead: this rank's slice of the next sub-batch walked beside the conflict phases below
                q0, q1 = nxt[0] + nxt[1] * self.rank // G, nxt[0] + nxt[1] * (self.rank + 1) // G
                check(lib.rb_shard_hash_begin_split(self.h, batch.h, nxt[0], nxt[1], q0, q1 - q0, self.ordinal + int(n), pos_bits, flags))
            mark("cache_upd")
        else:
            all_edges, e_sizes = yield ("gather", self._slot(N.SLOT_CONF_EDGES, 16 * nedge.value))
            if sum(e_sizes): check(lib.rb_shard_apply_tagged(self.h, _ptr(all_edges), sum(e_sizes) // 16))
        # components -> component owner -> ordered replay -> counter owners
        if sum(e_sizes):
            run_c, op_c = cnt(), cnt()
            check(lib.rb_shard_conflict_route(self.h, _ptr(all_edges), sum(e_sizes) // 16, G * (max(e_sizes) // 16), run_c, op_c, C.byref(st)))
            run_c, op_c = list(run_c), list(op_c)
            mark("conf_route")
            send = [self._slot(N.SLOT_CONF_RUNS, 24 * sum(run_c)), self._slot(N.SLOT_CONF_OPS, 4 * sum(op_c))]
            (r_runs, r_ops), (r_rc, r_oc) = yield ("a2a", send, [[24 * c for c in run_c], [4 * c for c in op_c]])
            cw_c = cnt()
            check(lib.rb_shard_conflict_replay(self.h, _ptr(r_runs), sum(r_rc) // 24, _ptr(r_ops), sum(r_oc) // 4, cw_c))
            cw_c = list(cw_c)
            mark("conf_replay")
            (o_cwidx, o_cwval), (o_cwc, _) = yield ("a2a", [self._slot(N.SLOT_CW_IDX, 8 * sum(cw_c)), self._slot(N.SLOT_CW_VAL, sum(cw_c))],
                                                    [[8 * c for c in cw_c], cw_c])
            check(lib.rb_shard_apply_writes(self.h, _ptr(o_cwidx), _ptr(o_cwval), sum(o_cwc) // 8))
            mark("conf_writes")
        self.ordinal += int(n)
        for kk in ("kmers", "pairs", "distinct", "conflict_ops", "sorted_kmers", "reads"):
            self.stats[kk] += getattr(st, kk)

    def query(self, what, h0, which_bits=N.DBGBF, out=None):
        """Coroutine: what 0 = lookup in bit filter which_bits, 1 = counting-filter count, 2 = graph count, for the
        hashes h0 (numpy u64; every rank calls this, possibly with an empty array).  The result lands in out[0]."""
        G = self.count
        a = np.ascontiguousarray(h0, dtype=np.uint64)
        b_c, c_c = (C.c_int64 * G)(), (C.c_int64 * G)()
        check(lib.rb_shard_query_make(self.h, what, which_bits, a.ctypes.data_as(C.c_void_p), a.size, b_c, c_c))
        b_c, c_c = list(b_c), list(c_c)
        (o_b, o_c), (o_bc, o_cc) = yield ("a2a", [self._slot(N.SLOT_Q_BIDX, 8 * sum(b_c)), self._slot(N.SLOT_Q_CIDX, 8 * sum(c_c))],
                                          [[8 * c for c in b_c], [8 * c for c in c_c]])
        nb, nc = sum(o_bc) // 8, sum(o_cc) // 8
        brep = torch.empty(nb, dtype=torch.uint8, device=self.tdev)
        crep = torch.empty(nc, dtype=torch.uint8, device=self.tdev)
        check(lib.rb_shard_query_serve(self.h, which_bits, _ptr(o_b), nb, _ptr(o_c), nc, _ptr(brep), _ptr(crep)))
        (my_b, my_c), _ = yield ("a2a", [brep, crep], [[c // 8 for c in o_bc], [c // 8 for c in o_cc]], [b_c, c_c])
        res = np.zeros(a.size, np.uint8 if what == 0 else np.float32)
        check(lib.rb_shard_query_finish(self.h, which_bits, _ptr(my_b), _ptr(my_c), res.ctypes.data_as(C.c_void_p) if what == 0 else None,
                                        res.ctypes.data_as(C.c_void_p) if what != 0 else None))
        if out is not None:
            out.append(res.astype(bool) if what == 0 else res)

    def getKmers(self, reads, out=None):
        """Coroutine: graph.getKmers(String) (R/bloom/hash/CanonicalHashFunction.java:46-78) for this rank's list of sequences
        on the SHARDED graph: the window hashes are rolled on this rank's GPU (rb_graph_kmers on a shard handle: hashes +
        usable flags), the counts come from ONE query exchange.  out gets (koffsets, f, r, count) as graph.getKmers returns."""
        lens = np.fromiter((len(x) for x in reads), np.int64, len(reads))
        off = np.zeros(len(reads) + 1, np.int64); np.cumsum(lens, out=off[1:])
        seq = np.frombuffer(b"".join(reads), np.uint8) if len(reads) else np.zeros(0, np.uint8)
        ko = np.zeros(len(reads) + 1, np.int64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        check(lib.rb_graph_kmers(self.h, vp(seq), vp(off), len(reads), vp(ko), None, None, None))
        t = int(ko[-1])
        f = np.zeros(t, np.uint64); r = np.zeros(t, np.uint64); usable = np.zeros(t, np.float32)
        if t:
            check(lib.rb_graph_kmers(self.h, vp(seq), vp(off), len(reads), vp(ko), vp(f), vp(r), vp(usable)))
        h0 = f if self.p.stranded else np.where(r.view(np.int64) < f.view(np.int64), r, f)
        ask = np.nonzero(usable > 0)[0]                          # windows with a non-ACGTU base count 0 without asking (:73-78)
        res = []
        yield from self.query(2, h0[ask], out=res)
        cnt = np.zeros(t, np.float32)
        cnt[ask] = res[0]
        if out is not None:
            out.append((ko, f, r, cnt))

    def neighbors(self, f, r, char_out, direction, out=None):
        """Coroutine: Kmer.getSuccessors / getPredecessors (direction 0 / 1, R/graph/Kmer.java:199-255) and the left / right
        variants (2 / 3) for this rank's k-mers on the SHARDED graph: candidate hashes on this rank's GPU (rb_graph_neighbors on
        a shard handle), their counts from one query exchange.  out gets (f4, r4, count4), shaped [n, 4]."""
        f = np.ascontiguousarray(np.atleast_1d(f), np.uint64); r = np.ascontiguousarray(np.atleast_1d(r), np.uint64)
        ch = np.ascontiguousarray(np.atleast_1d(char_out), np.uint8)
        n = f.size
        f4 = np.zeros((n, 4), np.uint64); r4 = np.zeros((n, 4), np.uint64); c4 = np.zeros((n, 4), np.float32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        check(lib.rb_graph_neighbors(self.h, vp(f), vp(r), vp(ch), n, direction, vp(f4), vp(r4), vp(c4)))
        h0 = f4 if self.p.stranded else np.where(r4.view(np.int64) < f4.view(np.int64), r4, f4)
        res = []
        yield from self.query(2, h0.reshape(-1), out=res)
        if out is not None:
            out.append((f4, r4, res[0].reshape(n, 4)))

    def traverse(self, kind, seeds, direction, bound=0, mode_or_lookahead=0, min_cov=1.0, targets=None, terminators=None, cap=4096,
                 answer_cap=0, gate=None, out=None):
        """Coroutine: rb_graph_walk (kind 0), rb_graph_greedy_extend (1) or rb_graph_naive_extend (2) for this rank's seed k-mers on
        the SHARDED graph (rb_shard_trav_*).  The walks run on this rank's GPU in the kernels the single-GPU calls use; whenever
        they need counts nobody has told them yet, all ranks do one query exchange (2 all-to-alls) and the walks replay their
        current step with the answers.  Every rank calls this with its own seeds (possibly none).  gate (greedy extension only):
        this rank's ShardRank of ANOTHER sharded graph whose dbgbf is the `bf` of greedyExtendRight(graph, source, lookahead, bound, bf).
        out gets (bases[n, width], f[n, width] or None, r or None, count or None, len[n], reason[n], rounds)."""
        G = self.count
        k = int(self.p.k)
        n = len(seeds)
        sb = np.frombuffer(b"".join(s if isinstance(s, bytes) else s.encode() for s in seeds), np.uint8) if n else np.zeros(1, np.uint8)
        if n and sb.size != n * k: raise ValueError("traverse: every seed must be one k-mer")
        tb = None
        if targets is not None and n:
            tb = np.frombuffer(b"".join(targets), np.uint8)
            if tb.size != n * k: raise ValueError("traverse: every target must be one k-mer")
        tseq = toff = None
        if kind == 2 and mode_or_lookahead == 0:
            from .graph import _pack
            tseq, toff = _pack([t if isinstance(t, bytes) else t.encode() for t in (terminators if terminators is not None else [b""] * n)])
            if tseq.size == 0: tseq = np.zeros(1, np.uint8)
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        check(lib.rb_shard_trav_begin(self.h, kind, vp(sb), vp(tb), n, direction, mode_or_lookahead, bound, cap, C.c_float(min_cov), vp(tseq), vp(toff), answer_cap))
        if gate is not None:
            check(lib.rb_shard_trav_set_gate(self.h, gate.h))
        while True:
            act = C.c_int64()
            b_c, c_c, g_c = (C.c_int64 * G)(), (C.c_int64 * G)(), (C.c_int64 * G)()
            check(lib.rb_shard_trav_advance(self.h, C.byref(act), b_c, c_c, g_c if gate is not None else None))
            flags = yield ("ints", [int(act.value > 0)])
            if not any(x[0] for x in flags):
                break
            b_c, c_c, g_c = list(b_c), list(c_c), list(g_c)
            req = [self._slot(N.SLOT_Q_BIDX, 8 * sum(b_c)), self._slot(N.SLOT_Q_CIDX, 8 * sum(c_c))]
            cnt = [[8 * c for c in b_c], [8 * c for c in c_c]]
            if gate is not None:                               # the gate's lookups travel with the counts
                req.append(gate._slot(N.SLOT_Q_BIDX, 8 * sum(g_c))); cnt.append([8 * c for c in g_c])
            got, got_c = yield ("a2a", req, cnt)
            o_b, o_c, o_bc, o_cc = got[0], got[1], got_c[0], got_c[1]
            nb, nc = sum(o_bc) // 8, sum(o_cc) // 8
            brep = torch.empty(nb, dtype=torch.uint8, device=self.tdev)
            crep = torch.empty(nc, dtype=torch.uint8, device=self.tdev)
            check(lib.rb_shard_query_serve(self.h, N.DBGBF, _ptr(o_b), nb, _ptr(o_c), nc, _ptr(brep), _ptr(crep)))
            rep, rep_c, known = [brep, crep], [[c // 8 for c in o_bc], [c // 8 for c in o_cc]], [b_c, c_c]
            if gate is not None:
                ng = sum(got_c[2]) // 8
                grep = torch.empty(ng, dtype=torch.uint8, device=self.tdev)
                check(lib.rb_shard_query_serve(gate.h, N.DBGBF, _ptr(got[2]), ng, None, 0, _ptr(grep), None))
                rep.append(grep); rep_c.append([c // 8 for c in got_c[2]]); known.append(g_c)
            mine, _ = yield ("a2a", rep, rep_c, known)
            check(lib.rb_shard_trav_absorb(self.h, _ptr(mine[0]), _ptr(mine[1]), _ptr(mine[2]) if gate is not None else None))
        width = max(1, (cap if mode_or_lookahead == 0 else bound + 1) if kind == 2 else bound)
        bases = np.zeros((n, width), np.uint8); ln = np.zeros(n, np.int32); reason = np.zeros(n, np.uint8)
        f = np.zeros((n, width), np.uint64) if kind != 1 else None
        r = np.zeros((n, width), np.uint64) if kind == 0 else None
        c = np.zeros((n, width), np.float32) if kind != 2 else None
        rounds = C.c_int64()
        check(lib.rb_shard_trav_end(self.h, vp(bases), vp(f), vp(r), vp(c), vp(ln), vp(reason), C.byref(rounds)))
        if out is not None:
            out.append((bases, f, r, c, ln, reason, rounds.value))

    def walk(self, seeds, direction, bound, min_cov=1.0, out=None, targets=None):
        """Coroutine: greedy maximum-coverage walks (rb_graph_walk) on the SHARDED graph; out gets (bases[n, bound], count[n, bound],
        len[n], reason[n])."""
        res = []
        yield from self.traverse(0, seeds, direction, bound, min_cov=min_cov, targets=targets, out=res)
        if out is not None:
            bases, _, _, c, ln, reason, _ = res[0]
            out.append((bases, c, ln, reason))

    def add_range(self, batch, first, n, flags, reads_per_substep, pos_bits):
        """Coroutine over all sub-batches of reads [first, first+n) — the same call on every rank."""
        n = int(n)
        # cold start (first insert into cleared filters): short sub-batches first, doubling up to the full
        # size, so the prefilter cache knows the hot k-mers before the big sub-batches arrive
        cur = reads_per_substep
        if self.ordinal == 0 and _RAMP and reads_per_substep >= 64 * 1024:
            cur = max(reads_per_substep // 64, 1024)
        cuts, a = [0], 0
        while a < n:
            a = min(n, a + cur)
            cuts.append(a)
            cur = min(reads_per_substep, cur * 2)
        for i in range(len(cuts) - 1):
            a, b = cuts[i], cuts[i + 1]
            nxt = (first + b, cuts[i + 2] - b) if i + 2 < len(cuts) else None
            yield from self.substep(batch, first + a, b - a, pos_bits, flags, nxt)
        if flags & N.ADD_STORE_READ_PAIRS:
            yield from self.flush_pairs()

    def flush_pairs(self):
        """Coroutine: the ranks' read-pair accumulation copies -> the owners' shards (rb_shard_pairs_flush_*): one all-to-all of
        G pieces of size / G bits, at the end of every collective insert call that stores read pairs (nothing travels on the routed
        path, RB_SHARD_PAIRS=route)"""
        G = self.count
        cnts, p = (C.c_int64 * G)(), C.c_void_p()
        check(lib.rb_shard_pairs_flush_begin(self.h, C.byref(p), cnts))
        cnts = list(cnts)
        send = torch.as_tensor(_DevView(p.value, sum(cnts)), device=self.tdev) if sum(cnts) else torch.empty(0, dtype=torch.uint8, device=self.tdev)
        (recv,), (rc,) = yield ("a2a", [send], [cnts])
        check(lib.rb_shard_pairs_flush_end(self.h, _ptr(recv), (C.c_int64 * G)(*rc)))
        trace_mark("pairs_flush")


# ------------------------------------------------------------------ drivers ----
class NativeComm:
    """rb_shard_comm: the exchange driver below the C ABI (csrc/rb_comm.hip).  loopback(G): a hub for G virtual ranks of this
    process (one host thread each); rccl(dist): one rank per process — rank 0's ncclUniqueId travels through torch.distributed."""

    def __init__(self, h):
        self.h = h

    @classmethod
    def loopback(cls, world):
        h = C.c_void_p()
        check(lib.rb_shard_comm_create_loopback(world, C.byref(h)))
        return cls(h)

    @classmethod
    def rccl(cls, dist, device, group=None):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = np.zeros(128, np.uint8)
        err = None
        if rank == 0:
            try:
                check(lib.rb_shard_comm_unique_id(uid.ctypes.data_as(C.c_void_p)))
            except Exception as e:      # noqa: BLE001 — every rank has to hear of it: the others are waiting in the broadcast below
                err = str(e)
        if world > 1:
            box = [None if err else uid.tobytes()]
            dist.broadcast_object_list(box, src=0, group=group)
            if box[0] is None:          # all ranks raise together (a rank 0 that raised alone would leave its peers in the broadcast for ever)
                raise RuntimeError("rank 0 could not make an RCCL unique id" + (": " + err if err else ""))
            uid = np.frombuffer(box[0], np.uint8).copy()
        elif err:
            raise RuntimeError(err)
        h = C.c_void_p()
        check(lib.rb_shard_comm_create_rccl(uid.ctypes.data_as(C.c_void_p), rank, world, device, C.byref(h)))
        return cls(h)

    def destroy(self):
        if self.h:
            lib.rb_shard_comm_destroy(self.h)
            self.h = None


def add_range_native(rank, comm, batch, first, n, flags, reads_per_substep, pos_bits):
    """ShardRank.add_range through rb_shard_add_range: every phase and every exchange inside the library (blocking call)"""
    st = N.AddStats()
    check(lib.rb_shard_add_range(rank.h, comm.h, batch.h, int(first), int(n), flags, int(reads_per_substep), pos_bits, rank.ordinal, C.byref(st)))
    rank.ordinal += int(n)
    for kk in ("kmers", "pairs", "distinct", "conflict_ops", "sorted_kmers", "reads"):
        rank.stats[kk] += getattr(st, kk)
    return st


def run_native_loopback(ranks, comm, batch, first, n, flags, reads_per_substep, pos_bits):
    """G virtual ranks of one process: one host thread per rank inside rb_shard_add_range (the calls release the GIL and meet
    at the hub's barriers); an error on one rank fails the hub and surfaces on all"""
    import threading
    errs = [None] * len(ranks)

    def work(i):
        try:
            add_range_native(ranks[i], comm, batch, first, n, flags, reads_per_substep, pos_bits)
        except Exception as e:          # noqa: BLE001 — re-raised below
            errs[i] = e
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(ranks))]
    for t in th: t.start()
    for t in th: t.join()
    for e in errs:
        if e is not None:
            raise e


def _split(t, counts):
    out, o = [], 0
    for c in counts:
        out.append(t[o:o + c]); o += c
    return out


def _sync(t):
    """torch ops and collectives are asynchronous on torch's stream; the library works on its own
    (non-blocking) stream, so results must be complete before their pointers are handed over."""
    if t is not None and t.is_cuda:
        torch.cuda.synchronize(t.device)


def run_loopback(gens):
    """Drive G coroutines (virtual ranks on one device) in lock step.
    Requests: ("ints", [..]) -> list over ranks;  ("a2a", tensors, byte counts per tensor [, known receive
    counts]) -> (received tensors, received byte counts per tensor);  ("gather", tensor | [tensors]) ->
    (concatenation over ranks, sizes) | [the same per tensor]."""
    G = len(gens)
    reqs = [next(g) for g in gens]
    while True:
        kind = reqs[0][0]
        assert all(r[0] == kind for r in reqs), "ranks diverged"
        if kind == "ints":
            res = [[r[1] for r in reqs]] * G
        elif kind == "a2a":
            m = len(reqs[0][1])
            res = []
            parts = [[_split(r[1][t], r[2][t]) for t in range(m)] for r in reqs]      # parts[src][t][dst]
            for dst in range(G):
                outs, cnts = [], []
                for t in range(m):
                    segs = [parts[src][t][dst] for src in range(G)]
                    outs.append(torch.cat(segs) if G > 1 else segs[0])
                    cnts.append([int(x.numel()) for x in segs])
                res.append((outs, cnts))
        elif kind == "gather":
            many = isinstance(reqs[0][1], (list, tuple))
            lists = [list(r[1]) if many else [r[1]] for r in reqs]
            out = []
            for t in range(len(lists[0])):
                sizes = [int(l[t].numel()) for l in lists]
                out.append((torch.cat([l[t] for l in lists]) if G > 1 else lists[0][t], sizes))
            res = [out if many else out[0]] * G
        else:
            raise ValueError(kind)
        if kind != "ints":
            first = reqs[0][1][0] if isinstance(reqs[0][1], (list, tuple)) else reqs[0][1]
            _sync(first)
        trace_mark("exchange")
        nxt, done = [], 0
        for g, r in zip(gens, res):
            try:
                nxt.append(g.send(r))
            except StopIteration:
                done += 1
        if done:
            assert done == G, "ranks finished at different points"
            return
        reqs = nxt


# RCCL (2.26, ROCm 7.0) delivers WRONG DATA, silently, when one peer's share of an all_to_all_single
# exceeds 1 GiB (seen on the 1.08 GB pair-probe exchange at world 1, where share = whole call).  Keep
# every per-peer message AND every call's total well below that: rounds of A2A_CHUNK bytes per peer,
# A2A_CHUNK = min(256 MiB, 512 MiB / world).
A2A_CHUNK = 256 << 20
A2A_CALL_TOTAL = 512 << 20
FUSE_LIMIT = 8 << 20      # phases whose largest per-peer message is below this travel as one fused collective


def _chunk(world):
    return max(1, min(A2A_CHUNK, A2A_CALL_TOTAL // max(1, world)))


def _all_to_all_bytes(dist, group, t, sc, rc, big):
    """all_to_all_single of uint8 tensor t with per-peer byte counts sc (send) / rc (receive), in rounds of
    at most A2A_CHUNK bytes per peer; `big` = the largest per-peer count on ANY rank (all ranks must run
    the same number of rounds)."""
    recv = torch.empty(sum(rc), dtype=torch.uint8, device=t.device)
    CH = _chunk(len(sc))
    if big <= CH:
        dist.all_to_all_single(recv, t, output_split_sizes=rc, input_split_sizes=sc, group=group)
        return recv
    soff = [sum(sc[:i]) for i in range(len(sc))]
    roff = [sum(rc[:i]) for i in range(len(rc))]
    for j in range(-(-big // CH)):
        lo = j * CH
        s_len = [max(0, min(c, lo + CH) - lo) for c in sc]
        r_len = [max(0, min(c, lo + CH) - lo) for c in rc]
        send_j = torch.cat([t[o + lo: o + lo + n] for o, n in zip(soff, s_len)])
        recv_j = torch.empty(sum(r_len), dtype=torch.uint8, device=t.device)
        dist.all_to_all_single(recv_j, send_j, output_split_sizes=r_len, input_split_sizes=s_len, group=group)
        p = 0
        for o, n in zip(roff, r_len):
            recv[o + lo: o + lo + n] = recv_j[p: p + n]
            p += n
    return recv


class _Driver:
    """torch.distributed exchange layer for one rank: ONE data collective per protocol phase.  The tensors of a
    phase are fused into one byte buffer per peer ([peer0: t0 t1 ..][peer1: t0 t1 ..]); their byte counts and
    this rank's largest per-peer total travel in one small all_to_all, from which every rank derives the same
    round count (no all_reduce)."""

    def __init__(self, dist, group):
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.last_big = 0          # largest fused per-peer message of the last counted phase (same on every rank)
        # gloo moves host memory: device tensors are staged through the CPU (multi-process tests on one GPU;
        # RCCL refuses two ranks on the same device)
        self.stage = dist.get_backend(group) == "gloo"

    def _cpu(self, t):
        return t.cpu() if (self.stage and t.is_cuda) else t

    def a2a(self, tensors, counts, known_rcs=None):
        dist, world, m = self.dist, self.world, len(tensors)
        home = tensors[0].device
        tensors = [self._cpu(t) for t in tensors]
        dev = tensors[0].device
        outs, rcs = self._a2a(tensors, counts, known_rcs, dist, world, m, dev)
        return [o.to(home) for o in outs] if dev != home else outs, rcs

    def _a2a(self, tensors, counts, known_rcs, dist, world, m, dev):
        tot_s = [sum(counts[t][p] for t in range(m)) for p in range(world)]
        if known_rcs is not None:                       # replies: sizes follow from the requests just exchanged
            rcs = [[int(c) for c in cs] for cs in known_rcs]
            big = max(self.last_big, 1)                 # an upper bound every rank shares (replies are smaller)
        else:
            cin = torch.tensor([[counts[t][p] for t in range(m)] + [max(tot_s, default=0)] for p in range(world)], dtype=torch.int64, device=dev)
            cout = torch.empty_like(cin)
            dist.all_to_all_single(cout, cin, group=self.group)
            rows = cout.tolist()
            rcs = [[int(rows[p][t]) for p in range(world)] for t in range(m)]
            big = max(max(int(r[m]) for r in rows), max(tot_s, default=0))
            self.last_big = big
        tot_r = [sum(rcs[t][p] for t in range(m)) for p in range(world)]
        if m > 1 and big > FUSE_LIMIT:                  # bulky phase: fusing would copy more than a collective costs
            outs = [_all_to_all_bytes(dist, self.group, tensors[t], counts[t], rcs[t], big) for t in range(m)]
            return outs, rcs
        if m == 1:
            send = tensors[0]
        else:
            soffs = [[sum(counts[t][:p]) for p in range(world)] for t in range(m)]
            send = torch.cat([tensors[t][soffs[t][p]: soffs[t][p] + counts[t][p]] for p in range(world) for t in range(m)])
        recv = _all_to_all_bytes(dist, self.group, send, tot_s, tot_r, big)
        if m == 1:
            outs = [recv]
        else:
            outs, base = [], [sum(tot_r[:p]) for p in range(world)]
            for t in range(m):
                inner = [sum(rcs[u][p] for u in range(t)) for p in range(world)]
                segs = [recv[base[p] + inner[p]: base[p] + inner[p] + rcs[t][p]] for p in range(world)]
                outs.append(torch.cat(segs) if world > 1 else segs[0])
        if _DEBUG and world == 1:
            torch.cuda.synchronize()
            import sys
            print("[a2a] %d bytes same=%s" % (send.numel(), all(bool(torch.equal(o, t)) for o, t in zip(outs, tensors))), file=sys.stderr, flush=True)
        return outs, rcs

    def gather(self, tensors):
        """all_gather of a list of byte tensors in one padded collective -> [(concatenation over ranks, sizes)] per tensor"""
        home = tensors[0].device
        tensors = [self._cpu(t) for t in tensors]
        res = self._gather(tensors)
        return [(c.to(home), sz) for c, sz in res] if tensors[0].device != home else res

    def _gather(self, tensors):
        dist, world, m = self.dist, self.world, len(tensors)
        dev = tensors[0].device
        mine = torch.tensor([int(t.numel()) for t in tensors], dtype=torch.int64, device=dev)
        allsz = torch.empty(world * m, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allsz, mine, group=self.group)
        allsz = allsz.view(world, m).tolist()
        tot = [sum(int(x) for x in row) for row in allsz]
        mx = max(tot)
        if not mx:
            return [(t, [0] * world) for t in tensors]
        fused = torch.cat(tensors) if m > 1 else tensors[0]
        parts = [[] for _ in range(world)]
        CH = _chunk(world)
        for lo in range(0, mx, CH):                      # rounds of bounded size (see A2A_CHUNK)
            w = min(CH, mx - lo)
            pad = torch.zeros(w, dtype=torch.uint8, device=dev)
            mine_n = max(0, min(int(fused.numel()), lo + w) - lo)
            pad[:mine_n] = fused[lo: lo + mine_n]
            out = torch.empty(w * world, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(out, pad, group=self.group)
            for r in range(world):
                n_r = max(0, min(tot[r], lo + w) - lo)
                if n_r:
                    parts[r].append(out[r * w: r * w + n_r])
        per_rank = [torch.cat(p) if len(p) > 1 else (p[0] if p else fused[:0]) for p in parts]     # rank r's fused bytes
        res = []
        for t in range(m):
            segs, sizes = [], []
            for r in range(world):
                o = sum(int(allsz[r][u]) for u in range(t))
                n = int(allsz[r][t])
                sizes.append(n)
                if n:
                    segs.append(per_rank[r][o: o + n])
            res.append((torch.cat(segs) if len(segs) > 1 else (segs[0] if segs else fused[:0]), sizes))
        return res


def run_distributed(gen, group=None):
    """Drive one rank's coroutine with torch.distributed collectives (RCCL for CUDA tensors, gloo for CPU)."""
    import torch.distributed as dist
    drv = _Driver(dist, group)
    try:
        req = next(gen)
        while True:
            kind = req[0]
            if kind == "ints":
                out = [None] * drv.world
                dist.all_gather_object(out, req[1], group=group)
                res = out
            elif kind == "a2a":
                counts = [[int(c) for c in cs] for cs in req[2]]
                res = drv.a2a(req[1], counts, req[3] if len(req) > 3 else None)
            elif kind == "gather":
                many = isinstance(req[1], (list, tuple))
                out = drv.gather(list(req[1]) if many else [req[1]])
                res = out if many else out[0]
            else:
                raise ValueError(kind)
            if kind != "ints":
                first = req[1][0] if isinstance(req[1], (list, tuple)) else req[1]
                _sync(first)
            trace_mark("exchange")
            req = gen.send(res)
    except StopIteration:
        return


def default_mode(count):
    """split reads from two ranks on (round 5: 247 against 273 ms per rank at two ranks, loopback — a rank of the split engine walks its half of
    the reads with the cooperative prefilter walker, a rank that hashes everything walks all of them with the ownership test in the loop;
    rounds 3-4 split from four ranks); one rank hashes everything (394 against 439 ms: nothing to route)"""
    return _MODE or ("split" if count >= 2 else "replicated")


def default_batch_kmers(count, mode=None):
    """k-mers per GLOBAL sub-batch when the caller names none.  With split reads a rank hashes 1/count of the
    sub-batch and owns 1/count of its k-mers, so the sub-batch grows with the ranks (2^30 windows per rank, at
    most 2^32): per-rank kernels stay large enough to fill the device (loopback, 8 ranks: 151 -> 126 ms per
    rank from 2^30 to 2^32).  With replicated hashing every rank walks the whole sub-batch: 2^30."""
    mode = mode or default_mode(count)
    if mode != "split":
        return 1 << 30
    # (round 5: 2^30 windows per rank instead of 2^29 below eight ranks — 248 -> 221 ms per rank at two ranks, 124 -> 112 at four, loopback; an
    # occurrence id has 32 bits, and a rank's scratch for a sub-batch none of whose windows the prefilter drops stays at the single-GPU engine's bound)
    return min(1 << 32, (1 << 30) * max(1, count))


def plan(max_len, k, count, max_batch_kmers=1 << 30):
    """(pos_bits, reads per GLOBAL sub-batch).  Every rank walks all reads of a sub-batch and keeps the
    k-mers it owns, so library scratch scales with max_batch_kmers / count per rank.  Bigger sub-batches
    merge more occurrences per run and need fewer exchange rounds; smaller ones keep the prefilter cache
    fresher and conflicts rarer.  An occurrence id is (read << pos_bits) | window start, and window starts
    stop at max_len - k."""
    pos_bits = 1
    while (1 << pos_bits) <= max(1, max_len - k):
        pos_bits += 1
    reads = max(1, max_batch_kmers // max(1, max_len))
    reads = min(reads, (1 << (32 - pos_bits)) - 1)
    return pos_bits, reads


class LoopbackCluster:
    """G virtual ranks on one GPU — exercises the full sharded protocol without a second device."""

    def __init__(self, count, dbgbfNumBits, cbfNumBytes, pkbfNumBits, dbgbfNumHash, cbfNumHash, pkbfNumHash, k, stranded,
                 useReadPairedKmers, device=0, rngSeed=0, maxBatchKmers=0, groupBits=0, mode=None, native=False):
        params = (dbgbfNumBits, cbfNumBytes, pkbfNumBits, dbgbfNumHash, cbfNumHash, pkbfNumHash, k, int(stranded),
                  int(useReadPairedKmers), device, groupBits, rngSeed, maxBatchKmers)
        self.k, self.count = k, count
        self.max_batch = maxBatchKmers or default_batch_kmers(count, mode)
        # the virtual ranks share ONE device: the per-rank copies of the read-pair filter (rb_shard.hip ShardState::rpk_acc) are `count`
        # full-size copies on it — fine at config 2 (8 x 1 GB), not beside filters sized for the whole device (configs[2]'s 17.8 GB each)
        import os
        crowded = count * (pkbfNumBits // 8) > (16 << 30) and "RB_SHARD_PAIRS" not in os.environ
        if crowded: os.environ["RB_SHARD_PAIRS"] = "route"
        try:
            self.ranks = [ShardRank(params, r, count, device, mode) for r in range(count)]
        finally:
            if crowded: del os.environ["RB_SHARD_PAIRS"]
        # native=True: the exchange driver below the C ABI (rb_shard_add_range over a loopback hub, one thread per rank)
        self.comm = NativeComm.loopback(count) if native else None

    def setReadPairedKmerDistance(self, d):
        for r in self.ranks:
            r.set_read_pair_distance(d)

    def addBatch(self, batch, max_len, reverseComplement=False, storeReadPairedKmers=False, reads_per_substep=None, first=0, n=None):
        """every virtual rank walks the same batch; insertion order = read order, as on one GPU"""
        flags = (N.ADD_REVCOMP if reverseComplement else 0) | (N.ADD_STORE_READ_PAIRS if storeReadPairedKmers else 0)
        pos_bits, rps = plan(max_len, self.k, self.count, self.max_batch)
        rps = reads_per_substep or rps
        n = batch.n_reads - first if n is None else n
        if self.comm is not None:
            run_native_loopback(self.ranks, self.comm, batch, first, n, flags, rps, pos_bits)
            return
        run_loopback([r.add_range(batch, first, n, flags, rps, pos_bits) for r in self.ranks])

    def _query(self, what, per_rank_h0, which_bits=N.DBGBF):
        outs = [[] for _ in self.ranks]
        run_loopback([r.query(what, h, which_bits, o) for r, h, o in zip(self.ranks, per_rank_h0, outs)])
        return [o[0] for o in outs]

    def contains(self, per_rank_h0):
        """per_rank_h0[r] = the hashes virtual rank r asks about -> list of bool arrays"""
        return self._query(0, per_rank_h0)

    def getCount(self, per_rank_h0):
        return self._query(2, per_rank_h0)

    def walkMaxCov(self, per_rank_seeds, direction, bound, minKmerCov=1.0):
        """per_rank_seeds[r] = the seed k-mers of virtual rank r -> per rank (bases, count, len, reason), as graph.walkMaxCov"""
        outs = [[] for _ in self.ranks]
        run_loopback([r.walk(sd, direction, bound, minKmerCov, o) for r, sd, o in zip(self.ranks, per_rank_seeds, outs)])
        return [o[0] for o in outs]

    def traverse(self, kind, per_rank_seeds, direction, targets=None, terminators=None, gate=None, **kw):
        """rb_graph_walk / _greedy_extend / _naive_extend (kind 0 / 1 / 2) on the sharded graph; targets / terminators: one list per
        rank (or None); gate: another LoopbackCluster with the same number of ranks whose dbgbf is the greedy extension's `bf`.
        -> per rank (bases, f, r, count, len, reason, rounds)"""
        outs = [[] for _ in self.ranks]
        run_loopback([r.traverse(kind, sd, direction, targets=targets[i] if targets else None, terminators=terminators[i] if terminators else None,
                                 gate=gate.ranks[i] if gate is not None else None, out=o, **kw) for i, (r, sd, o) in enumerate(zip(self.ranks, per_rank_seeds, outs))])
        return [o[0] for o in outs]

    def greedyExtend(self, per_rank_seeds, direction, lookahead, bound, answer_cap=0, bf=None):
        """GraphUtils.greedyExtendRight / Left on the sharded graph (bf: the gated variants' filter, a LoopbackCluster whose dbgbf it
        is) -> per rank (bases, count, len, reason), as graph.greedyExtend"""
        return [(b, c, ln, rs) for b, _, _, c, ln, rs, _ in self.traverse(1, per_rank_seeds, direction, bound=bound, mode_or_lookahead=lookahead, answer_cap=answer_cap, gate=bf)]

    def naiveExtend(self, per_rank_seeds, direction, mode=1, bound=0, minKmerCov=1.0, terminators=None, cap=4096):
        """GraphUtils.naiveExtendRight / Left on the sharded graph -> per rank (list of appended bases, reason), as graph.naiveExtend"""
        res = self.traverse(2, per_rank_seeds, direction, bound=bound, mode_or_lookahead=mode, min_cov=minKmerCov, terminators=terminators, cap=cap)
        return [([bytes(b[i, :ln[i]]) for i in range(len(ln))], rs) for b, _, _, _, ln, rs, _ in res]

    def getKmers(self, per_rank_reads):
        """per_rank_reads[r] = the sequences virtual rank r asks about -> per rank (koffsets, f, r, count), as graph.getKmers"""
        outs = [[] for _ in self.ranks]
        run_loopback([rk.getKmers(rd, o) for rk, rd, o in zip(self.ranks, per_rank_reads, outs)])
        return [o[0] for o in outs]

    def getNeighbors(self, per_rank_frc, direction):
        """per_rank_frc[r] = (f, r, charOut) of virtual rank r's k-mers -> per rank (f4, r4, count4), as graph.getNeighbors"""
        outs = [[] for _ in self.ranks]
        run_loopback([rk.neighbors(a, b, c, direction, o) for rk, (a, b, c), o in zip(self.ranks, per_rank_frc, outs)])
        return [o[0] for o in outs]

    def getCbfCount(self, per_rank_h0):
        return self._query(1, per_rank_h0)

    def lookupReadKmerPair(self, per_rank_h0):
        return self._query(0, per_rank_h0, N.RPKBF)

    def exportFilter(self, which):
        return np.concatenate([r.local_filter(which) for r in self.ranks])

    def popcount(self, which):
        return sum(r.local_popcount(which) for r in self.ranks)

    def fold(self, which):
        """digest of the whole distributed filter = sum of the shards' digests mod 2^64 (rb_filter_fold)"""
        return sum(r.local_fold(which) for r in self.ranks) & 0xFFFFFFFFFFFFFFFF

    def destroy(self):
        for r in self.ranks:
            r.destroy()
        if self.comm is not None:
            self.comm.destroy()
