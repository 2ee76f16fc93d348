"""rnabloom.sharded — multi-GPU stage-1 insert: filters split by index range across ranks.

One rank = one process = one GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI), or, for
single-GPU testing, G virtual ranks inside one process (`LoopbackCluster`).  The per-rank work is
the phase functions of the C ABI (`rb_shard_*`, csrc/rb_shard.hip); this module only moves byte
buffers between ranks (all_to_all / all_gather) in the order the protocol requires.  Results are
identical to the single-GPU path and to the sequential oracle (tests/test_gpu_sharded.py).

Read distribution: a global sub-batch is the concatenation, in rank order, of the slices the ranks
contribute to it; global order = sub-batch after sub-batch.  (A loader deals blocks of reads to the
ranks in turn; bench.py generates each rank's blocks in place.)
"""
import ctypes as C

import numpy as np
import torch

if torch.cuda.is_available():
    torch.cuda.init()   # torch's bundled HIP runtime must be initialised before librb_hip.so creates its context

from . import _native as N
from ._native import check, lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None


class ShardRank:
    """One rank's shard of the graph + the coroutine that drives one global sub-batch."""

    def __init__(self, params, rank, count, device):
        self.rank, self.count, self.device = rank, count, device
        self.tdev = torch.device("cuda", device)
        p = N.GraphParams(*params)
        p.device = device
        self.p = p
        self.h = C.c_void_p()
        check(lib.rb_graph_create_shard(C.byref(p), rank, count, C.byref(self.h)))
        self.ordinal = 0
        self.stats = dict(kmers=0, pairs=0, distinct=0, conflict_ops=0, reads=0)

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

    # ---- one global sub-batch; yields exchange requests, receives their results ----
    def substep(self, batch, first, n, pos_bits, flags, mode=N.MODE_ADD):
        import os, time
        trace = os.environ.get("RB_SHARD_TRACE")
        t_last = [time.perf_counter()]

        def mark(what):
            if trace and self.rank == 0:
                torch.cuda.synchronize()
                now = time.perf_counter()
                print("[shard] %-10s %8.1f ms" % (what, (now - t_last[0]) * 1e3), flush=True)
                t_last[0] = now
        G = self.count
        n_all = yield ("ints", [int(n)])
        n_all = [x[0] for x in n_all]
        rel_base, total_reads = sum(n_all[: self.rank]), sum(n_all)
        assert total_reads < (1 << (32 - pos_bits)), "sub-batch has too many reads for the occurrence id"
        rec_c, pair_c = (C.c_int64 * G)(), (C.c_int64 * G)()
        check(lib.rb_shard_hash(self.h, batch.h, first, n, rel_base, pos_bits, flags, rec_c, pair_c))
        rec_c, pair_c = list(rec_c), list(pair_c)
        mark("hash")
        keys = self._take(N.SLOT_REC_KEYS, 8 * sum(rec_c))
        occ = self._take(N.SLOT_REC_OCC, 4 * sum(rec_c))
        pidx = self._take(N.SLOT_PAIR_IDX, 8 * sum(pair_c))
        rkeys, rk_c = yield ("a2a", keys, [8 * c for c in rec_c])
        rocc, _ = yield ("a2a", occ, [4 * c for c in rec_c])
        rpidx, rp_c = yield ("a2a", pidx, [8 * c for c in pair_c])
        mark("a2a_rec")
        nrec = sum(rk_c) // 8
        d_c, c_c = (C.c_int64 * G)(), (C.c_int64 * G)()
        check(lib.rb_shard_group(self.h, _ptr(rkeys), _ptr(rocc), nrec, self.ordinal, pos_bits, mode, d_c, c_c))
        d_c, c_c = list(d_c), list(c_c)
        mark("group")
        o_didx, o_dc = yield ("a2a", self._take(N.SLOT_DREQ_IDX, 8 * sum(d_c)), [8 * c for c in d_c])
        o_dprobe, _ = yield ("a2a", self._take(N.SLOT_DREQ_PROBE, 8 * sum(d_c)), [8 * c for c in d_c])
        o_cidx, o_cc = yield ("a2a", self._take(N.SLOT_CREQ_IDX, 8 * sum(c_c)), [8 * c for c in c_c])
        mark("a2a_req")
        nd, nc, np_ = sum(o_dc) // 8, sum(o_cc) // 8, sum(rp_c) // 8
        dreply = torch.empty(nd, dtype=torch.uint8, device=self.tdev)
        creply = torch.empty(nc, dtype=torch.uint8, device=self.tdev)
        check(lib.rb_shard_serve(self.h, mode, _ptr(o_didx), _ptr(o_dprobe), nd, _ptr(o_cidx), nc, _ptr(rpidx), np_,
                                 _ptr(dreply), _ptr(creply)))
        mark("serve")
        my_dreply, _ = yield ("a2a", dreply, [c // 8 for c in o_dc])
        my_creply, _ = yield ("a2a", creply, [c // 8 for c in o_cc])
        mark("a2a_reply")
        w_c = (C.c_int64 * G)()
        nco, ncc = C.c_int64(), C.c_int64()
        st = N.AddStats()
        check(lib.rb_shard_resolve(self.h, mode, _ptr(my_dreply), _ptr(my_creply), w_c, C.byref(nco), C.byref(ncc), C.byref(st)))
        w_c = list(w_c)
        mark("resolve")
        o_widx, o_wc = yield ("a2a", self._take(N.SLOT_W_IDX, 8 * sum(w_c)), [8 * c for c in w_c])
        o_wval, _ = yield ("a2a", self._take(N.SLOT_W_VAL, sum(w_c)), w_c)
        check(lib.rb_shard_apply_writes(self.h, _ptr(o_widx), _ptr(o_wval), sum(o_wc) // 8))
        mark("writes")
        all_ops = yield ("gather", self._take(N.SLOT_CONF_OPS, 16 * nco.value))
        all_ctr = yield ("gather", self._take(N.SLOT_CONF_CTR, 16 * ncc.value))
        mark("gather")
        check(lib.rb_shard_conflict_replay(self.h, _ptr(all_ops), all_ops.numel() // 16, _ptr(all_ctr), all_ctr.numel() // 16))
        mark("replay")
        self.ordinal += total_reads
        del keys, occ, pidx, rkeys, rocc, rpidx, o_didx, o_dprobe, o_cidx, dreply, creply, my_dreply, my_creply, o_widx, o_wval, all_ops, all_ctr
        self.stats["kmers"] += sum(rec_c)
        self.stats["pairs"] += sum(pair_c) // max(1, self.p.pkbf_num_hash)
        self.stats["distinct"] += st.distinct
        self.stats["conflict_ops"] += st.conflict_ops
        self.stats["reads"] += int(n)

    def add_range(self, batch, first, n, flags, reads_per_substep, pos_bits):
        """Coroutine over all sub-batches of this rank's read range (ranks may hold different counts)."""
        steps = -(-int(n) // reads_per_substep) if n else 0
        steps_all = yield ("ints", [steps])
        for t in range(max(x[0] for x in steps_all)):
            a = min(int(n), t * reads_per_substep)
            b = min(int(n), (t + 1) * reads_per_substep)
            yield from self.substep(batch, first + a, b - a, pos_bits, flags)


# ------------------------------------------------------------------ drivers ----
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
    """Drive G coroutines (virtual ranks on one device) in lock step."""
    G = len(gens)
    reqs = [next(g) for g in gens]
    while True:
        kind = reqs[0][0]
        assert all(r[0] == kind for r in reqs), "ranks diverged"
        if kind == "ints":
            res = [[r[1] for r in reqs]] * G
        elif kind == "a2a":
            parts = [_split(r[1], r[2]) for r in reqs]
            res = []
            for dst in range(G):
                segs = [parts[src][dst] for src in range(G)]
                res.append((torch.cat(segs) if segs else reqs[dst][1][:0], [int(s.numel()) for s in segs]))
        elif kind == "gather":
            cat = torch.cat([r[1] for r in reqs])
            res = [cat] * G
        else:
            raise ValueError(kind)
        _sync(reqs[0][1] if kind != "ints" else None)
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


def run_distributed(gen, group=None):
    """Drive one rank's coroutine with torch.distributed collectives (RCCL for CUDA tensors, gloo for CPU)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    try:
        req = next(gen)
        while True:
            kind = req[0]
            if kind == "ints":
                out = [None] * world
                dist.all_gather_object(out, req[1], group=group)
                res = out
            elif kind == "a2a":
                send, counts = req[1], [int(c) for c in req[2]]
                cin = torch.tensor(counts, dtype=torch.int64, device=send.device)
                cout = torch.empty(world, dtype=torch.int64, device=send.device)
                dist.all_to_all_single(cout, cin, group=group)
                rc = [int(x) for x in cout.tolist()]
                recv = torch.empty(sum(rc), dtype=torch.uint8, device=send.device)
                dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=counts, group=group)
                res = (recv, rc)
            elif kind == "gather":
                t = req[1]
                sizes = [None] * world
                dist.all_gather_object(sizes, int(t.numel()), group=group)
                mx = max(sizes)
                pad = torch.zeros(mx, dtype=torch.uint8, device=t.device)
                pad[: t.numel()] = t
                outs = [torch.empty(mx, dtype=torch.uint8, device=t.device) for _ in range(world)]
                if mx:
                    dist.all_gather(outs, pad, group=group)
                res = torch.cat([o[:s] for o, s in zip(outs, sizes)]) if mx else t
            else:
                raise ValueError(kind)
            _sync(req[1] if kind != "ints" else None)
            req = gen.send(res)
    except StopIteration:
        return


def plan(max_len, k, count, max_batch_kmers=1 << 30):
    """(pos_bits, reads per rank per sub-batch) so that a global sub-batch stays within the limits.
    Per rank a sub-batch is capped at 2^27 k-mers: exchange buffers (torch) and library scratch both
    scale with it."""
    pos_bits = 1
    while (1 << pos_bits) <= max_len:
        pos_bits += 1
    per_read = max(1, max_len)
    reads = max(1, min(max_batch_kmers // count, 1 << 27) // per_read)   # 8-byte keys: < 2^31 bytes per exchange
    reads = min(reads, ((1 << (32 - pos_bits)) - 1) // count)
    return pos_bits, max(1, reads)


class LoopbackCluster:
    """G virtual ranks on one GPU — exercises the full sharded protocol without a second device."""

    def __init__(self, count, dbgbfNumBits, cbfNumBytes, pkbfNumBits, dbgbfNumHash, cbfNumHash, pkbfNumHash, k, stranded,
                 useReadPairedKmers, device=0, rngSeed=0, maxBatchKmers=0, groupBits=0):
        params = (dbgbfNumBits, cbfNumBytes, pkbfNumBits, dbgbfNumHash, cbfNumHash, pkbfNumHash, k, int(stranded),
                  int(useReadPairedKmers), device, groupBits, rngSeed, maxBatchKmers)
        self.k, self.count = k, count
        self.max_batch = maxBatchKmers or (1 << 30)
        self.ranks = [ShardRank(params, r, count, device) for r in range(count)]

    def setReadPairedKmerDistance(self, d):
        for r in self.ranks:
            r.set_read_pair_distance(d)

    def addBatches(self, batches, max_len, reverseComplement=False, storeReadPairedKmers=False, reads_per_substep=None):
        """batches[r] = this rank's reads; global order = sub-batch by sub-batch, rank by rank."""
        flags = (N.ADD_REVCOMP if reverseComplement else 0) | (N.ADD_STORE_READ_PAIRS if storeReadPairedKmers else 0)
        pos_bits, rps = plan(max_len, self.k, self.count, self.max_batch)
        rps = reads_per_substep or rps
        gens = [r.add_range(b, 0, b.n_reads, flags, rps, pos_bits) for r, b in zip(self.ranks, batches)]
        run_loopback(gens)

    def exportFilter(self, which):
        return np.concatenate([r.local_filter(which) for r in self.ranks])

    def popcount(self, which):
        return sum(r.local_popcount(which) for r in self.ranks)

    def destroy(self):
        for r in self.ranks:
            r.destroy()
