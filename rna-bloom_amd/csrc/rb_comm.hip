// rb_comm.hip — the exchange driver of the sharded engine below the C ABI.
//
// rnabloom/sharded.py drives one global sub-batch as "phase, exchange, phase, exchange, ..." from Python over
// torch.distributed.  This file is the same protocol in the library: rb_shard_add_range() runs every phase of csrc/rb_shard.hip
// and moves the byte buffers between ranks itself, on the handle's own stream, with nothing but the per-peer byte counts
// ever visiting the host.  Two transports:
//   * RCCL (one process per GPU): ncclSend / ncclRecv groups on the handle's stream.  librccl is taken with dlopen — a
//     process that also holds torch has torch's copy loaded already, and one copy must serve both.  Every pair of ranks
//     cuts its own message into pieces of at most 256 MiB (both ends know the byte count, so no agreement round is
//     needed): RCCL 2.26 / ROCm 7.0 delivered wrong data for single transfers above 1 GiB.
//   * loopback hub (G virtual ranks = G host threads of one process on one GPU): device-to-device copies between the
//     ranks' buffers, rendezvous through a barrier.  This is what the tests run the protocol with on a one-GPU box.
// The reference has no counterpart (one shared-memory process); the interface mirrors rnabloom/sharded.py::ShardRank.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "rb_pipeline.hpp"

using namespace rb;

namespace rb {
bool shard_is_split(const rb_graph *g);     // rb_shard.hip: split-reads mode (else replicated hashing)
}

namespace {
constexpr int MAX_WORLD = 64;
constexpr int MAX_PARTS = 4;
constexpr size_t PIECE = (size_t)256 << 20;

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi &rccl() {
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) return;
#define RB_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, #sym))
        RB_SYM(GetUniqueId, ncclGetUniqueId); RB_SYM(CommInitRank, ncclCommInitRank); RB_SYM(CommDestroy, ncclCommDestroy);
        RB_SYM(Send, ncclSend); RB_SYM(Recv, ncclRecv); RB_SYM(GroupStart, ncclGroupStart); RB_SYM(GroupEnd, ncclGroupEnd);
        RB_SYM(GetErrorString, ncclGetErrorString);
#undef RB_SYM
    });
    return a;
}
void need_rccl() {
    RcclApi &a = rccl();
    if (!a.lib || !a.GetUniqueId || !a.CommInitRank || !a.Send || !a.Recv || !a.GroupStart || !a.GroupEnd) {
        set_error("librccl.so could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing");
        throw HipError{RB_ERR_STATE};
    }
}
#define RB_NCCL(x)                                                                                                      \
    do {                                                                                                                \
        ncclResult_t r_ = (x);                                                                                          \
        if (r_ != ncclSuccess) {                                                                                        \
            set_error("%s failed: %s", #x, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?");                    \
            throw HipError{RB_ERR_HIP};                                                                                 \
        }                                                                                                               \
    } while (0)
#define RB_CK(x)                                                                                                        \
    do {                                                                                                                \
        int rc_ = (x);                                                                                                  \
        if (rc_ != RB_OK) throw HipError{rc_};                                                                          \
    } while (0)

// what one rank hands to an exchange: k buffers, each cut into `world` consecutive pieces (bytes per destination)
struct Part {
    const void *send = nullptr;
    int64_t sc[MAX_WORLD];         // bytes to each destination
    void *recv = nullptr;          // out: pieces from rank 0, 1, ... one after the other
    int64_t rc[MAX_WORLD];         // out: bytes from each source
    int64_t rtotal = 0;
};
}  // namespace

struct rb_shard_comm {
    int world = 1;
    bool is_rccl = false;
    // RCCL
    ncclComm_t nc = nullptr;
    int rank = 0, device = 0;
    // loopback hub
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool failed = false;
    // A failure poisons the hub until EVERY rank has entered and left the collective call it happened in.  Ranks make the same sequence of
    // collective calls (rb_shard_add_range, rb_shard_comm_selftest), so a call is known by its number: calls[r] = calls rank r has entered,
    // left[r] = the number of the last call it left, fail_call = the latest call number a failure (its own or a poisoned hub's) was seen in.
    // A rank that fails fast and comes back finds the hub poisoned and fails again — which moves fail_call on to ITS call number, so the
    // hub stays poisoned until its peers have failed that call as well and all ranks are in step again.
    uint64_t calls[MAX_WORLD] = {}, left[MAX_WORLD] = {}, fail_call = 0;
    const Part *pub_parts[MAX_WORLD];
    int pub_k[MAX_WORLD];
    // receive buffers and count staging, per (virtual) rank
    struct Pool {
        DevBuf buf[MAX_PARTS];
        DevBuf cnt_dev;
        int64_t *cnt_host = nullptr;    // pinned, 2 * MAX_PARTS * MAX_WORLD
    } pool[MAX_WORLD];
};

namespace {
struct HubFailed {};
// all `world` threads meet here; a rank that failed marks the hub so that nobody waits forever
void hub_barrier(rb_shard_comm *c) {
    std::unique_lock<std::mutex> lk(c->m);
    if (c->failed) throw HubFailed{};
    const uint64_t gen = c->generation;
    if (++c->arrived == c->world) { c->arrived = 0; ++c->generation; c->cv.notify_all(); return; }
    c->cv.wait(lk, [&] { return c->generation != gen || c->failed; });
    if (c->failed) throw HubFailed{};
}
void hub_fail(rb_shard_comm *c, uint64_t call) {
    std::lock_guard<std::mutex> lk(c->m);
    c->failed = true;
    c->fail_call = std::max(c->fail_call, call);
    c->cv.notify_all();
}
// one collective call of rank `me` on a loopback hub (see rb_shard_comm::calls)
struct InCall {
    rb_shard_comm *c; int me; uint64_t call = 0;
    InCall(rb_shard_comm *c_, int me_) : c(c_ && !c_->is_rccl && me_ >= 0 && me_ < c_->world ? c_ : nullptr), me(me_) {
        if (c) { std::lock_guard<std::mutex> lk(c->m); call = ++c->calls[me]; }
    }
    void failed() { if (c) hub_fail(c, call); }
    ~InCall() {
        if (!c) return;
        std::lock_guard<std::mutex> lk(c->m);
        c->left[me] = call;
        if (!c->failed) return;
        // The hub is cleaned only when NOBODY is inside a collective call (calls[r] == left[r]) and everybody has been through the failed
        // one: a peer that already waits in a barrier of a LATER call (this rank failed after its last barrier) must still find the hub
        // poisoned when it wakes — cleaning here would erase its arrival and it would wait for ever.
        for (int r = 0; r < c->world; ++r) if (c->left[r] < c->fail_call || c->calls[r] != c->left[r]) return;
        c->failed = false; c->arrived = 0; c->fail_call = 0;
    }
};
rb_shard_comm::Pool &pool_of(rb_shard_comm *c, int me) {
    rb_shard_comm::Pool &p = c->pool[c->is_rccl ? 0 : me];
    if (!p.cnt_host) RB_HIP(hipHostMalloc(reinterpret_cast<void **>(&p.cnt_host), sizeof(int64_t) * 2 * MAX_PARTS * MAX_WORLD, hipHostMallocDefault));
    return p;
}

// ---- all-to-all of k buffers.  known[t] != nullptr: the receive counts of buffer t are known (replies to requests) ----
void a2a(rb_shard_comm *c, int me, Part *parts, int k, const int64_t *const *known, hipStream_t st) {
    const int G = c->world;
    RB_REQUIRE(k >= 1 && k <= MAX_PARTS, "exchange of %d buffers", k);
    rb_shard_comm::Pool &P = pool_of(c, me);
    if (!c->is_rccl) {
        RB_HIP(hipStreamSynchronize(st));                       // my send buffers are complete
        { std::lock_guard<std::mutex> lk(c->m); c->pub_parts[me] = parts; c->pub_k[me] = k; }
        hub_barrier(c);
        for (int t = 0; t < k; ++t) {
            int64_t tot = 0;
            for (int src = 0; src < G; ++src) { parts[t].rc[src] = c->pub_parts[src][t].sc[me]; tot += parts[t].rc[src]; }
            parts[t].rtotal = tot;
            P.buf[t].reserve((size_t)std::max<int64_t>(tot, 1));
            parts[t].recv = P.buf[t].p;
            int64_t ro = 0;
            for (int src = 0; src < G; ++src) {
                const Part &sp = c->pub_parts[src][t];
                int64_t so = 0;
                for (int d = 0; d < me; ++d) so += sp.sc[d];
                if (parts[t].rc[src])
                    RB_HIP(hipMemcpyAsync(static_cast<char *>(parts[t].recv) + ro, static_cast<const char *>(sp.send) + so, (size_t)parts[t].rc[src],
                                          hipMemcpyDeviceToDevice, st));
                ro += parts[t].rc[src];
            }
        }
        RB_HIP(hipStreamSynchronize(st));
        hub_barrier(c);                                         // everybody has taken its pieces: send buffers may be reused
        return;
    }
    RcclApi &R = rccl();
    // 1. receive counts (one small group), unless every buffer's are known
    bool all_known = known != nullptr;
    for (int t = 0; t < k && all_known; ++t) all_known = known[t] != nullptr;
    if (!all_known) {
        P.cnt_dev.reserve(sizeof(int64_t) * 2 * MAX_PARTS * MAX_WORLD);
        int64_t *hs = P.cnt_host, *hr = P.cnt_host + MAX_PARTS * MAX_WORLD;
        int64_t *ds = P.cnt_dev.as<int64_t>(), *dr = ds + MAX_PARTS * MAX_WORLD;
        for (int d = 0; d < G; ++d)
            for (int t = 0; t < k; ++t) hs[d * k + t] = parts[t].sc[d];
        RB_HIP(hipMemcpyAsync(ds, hs, sizeof(int64_t) * (size_t)G * k, hipMemcpyHostToDevice, st));
        RB_NCCL(R.GroupStart());
        for (int p = 0; p < G; ++p) {
            RB_NCCL(R.Send(ds + p * k, (size_t)k, ncclInt64, p, c->nc, st));
            RB_NCCL(R.Recv(dr + p * k, (size_t)k, ncclInt64, p, c->nc, st));
        }
        RB_NCCL(R.GroupEnd());
        RB_HIP(hipMemcpyAsync(hr, dr, sizeof(int64_t) * (size_t)G * k, hipMemcpyDeviceToHost, st));
        RB_HIP(hipStreamSynchronize(st));
        for (int src = 0; src < G; ++src)
            for (int t = 0; t < k; ++t) parts[t].rc[src] = hr[src * k + t];
    }
    for (int t = 0; t < k; ++t) {
        if (known && known[t]) for (int src = 0; src < G; ++src) parts[t].rc[src] = known[t][src];
        int64_t tot = 0;
        for (int src = 0; src < G; ++src) tot += parts[t].rc[src];
        parts[t].rtotal = tot;
        P.buf[t].reserve((size_t)std::max<int64_t>(tot, 1));
        parts[t].recv = P.buf[t].p;
    }
    // 2. the data: round r moves piece r of every (buffer, peer) message that has one
    for (size_t round = 0;; ++round) {
        bool any = false;
        const size_t lo = round * PIECE;
        for (int t = 0; t < k && !any; ++t)
            for (int p = 0; p < G && !any; ++p) any = (size_t)parts[t].sc[p] > lo || (size_t)parts[t].rc[p] > lo;
        if (!any) break;
        RB_NCCL(R.GroupStart());
        for (int t = 0; t < k; ++t) {
            int64_t so = 0, ro = 0;
            for (int p = 0; p < G; ++p) {
                const size_t sn = (size_t)parts[t].sc[p] > lo ? std::min(PIECE, (size_t)parts[t].sc[p] - lo) : 0;
                const size_t rn = (size_t)parts[t].rc[p] > lo ? std::min(PIECE, (size_t)parts[t].rc[p] - lo) : 0;
                if (sn) RB_NCCL(R.Send(static_cast<const char *>(parts[t].send) + so + lo, sn, ncclUint8, p, c->nc, st));
                if (rn) RB_NCCL(R.Recv(static_cast<char *>(parts[t].recv) + ro + lo, rn, ncclUint8, p, c->nc, st));
                so += parts[t].sc[p]; ro += parts[t].rc[p];
            }
        }
        RB_NCCL(R.GroupEnd());
    }
    RB_HIP(hipStreamSynchronize(st));       // the phases that follow read the buffers from other streams too
}
// ---- all-gather of one variable-length buffer per rank: every rank ends up with rank 0's, rank 1's, ... bytes in a row ----
void gather(rb_shard_comm *c, int me, const void *send, int64_t nbytes, int pool_slot, void **recv, int64_t *sizes, hipStream_t st) {
    const int G = c->world;
    rb_shard_comm::Pool &P = pool_of(c, me);
    if (!c->is_rccl) {
        Part mine;
        mine.send = send;
        for (int d = 0; d < G; ++d) mine.sc[d] = nbytes;       // sc[d] abused as "my length" (same for every destination)
        RB_HIP(hipStreamSynchronize(st));
        { std::lock_guard<std::mutex> lk(c->m); c->pub_parts[me] = &mine; c->pub_k[me] = 1; }
        hub_barrier(c);
        int64_t tot = 0;
        for (int src = 0; src < G; ++src) { sizes[src] = c->pub_parts[src][0].sc[me]; tot += sizes[src]; }
        P.buf[pool_slot].reserve((size_t)std::max<int64_t>(tot, 1));
        int64_t ro = 0;
        for (int src = 0; src < G; ++src) {
            if (sizes[src]) RB_HIP(hipMemcpyAsync(P.buf[pool_slot].as<char>() + ro, c->pub_parts[src][0].send, (size_t)sizes[src], hipMemcpyDeviceToDevice, st));
            ro += sizes[src];
        }
        RB_HIP(hipStreamSynchronize(st));
        hub_barrier(c);
        *recv = P.buf[pool_slot].p;
        return;
    }
    RcclApi &R = rccl();
    P.cnt_dev.reserve(sizeof(int64_t) * 2 * MAX_PARTS * MAX_WORLD);
    int64_t *hs = P.cnt_host, *hr = P.cnt_host + MAX_PARTS * MAX_WORLD;
    int64_t *ds = P.cnt_dev.as<int64_t>(), *dr = ds + MAX_PARTS * MAX_WORLD;
    hs[0] = nbytes;
    RB_HIP(hipMemcpyAsync(ds, hs, sizeof(int64_t), hipMemcpyHostToDevice, st));
    RB_NCCL(R.GroupStart());
    for (int p2 = 0; p2 < G; ++p2) {
        RB_NCCL(R.Send(ds, 1, ncclInt64, p2, c->nc, st));
        RB_NCCL(R.Recv(dr + p2, 1, ncclInt64, p2, c->nc, st));
    }
    RB_NCCL(R.GroupEnd());
    RB_HIP(hipMemcpyAsync(hr, dr, sizeof(int64_t) * (size_t)G, hipMemcpyDeviceToHost, st));
    RB_HIP(hipStreamSynchronize(st));
    int64_t tot = 0;
    for (int src = 0; src < G; ++src) { sizes[src] = hr[src]; tot += sizes[src]; }
    P.buf[pool_slot].reserve((size_t)std::max<int64_t>(tot, 1));
    for (size_t round = 0;; ++round) {
        const size_t lo = round * PIECE;
        bool any = (size_t)nbytes > lo;
        for (int src = 0; src < G && !any; ++src) any = (size_t)sizes[src] > lo;
        if (!any) break;
        RB_NCCL(R.GroupStart());
        int64_t ro = 0;
        for (int p2 = 0; p2 < G; ++p2) {
            const size_t sn = (size_t)nbytes > lo ? std::min(PIECE, (size_t)nbytes - lo) : 0;
            const size_t rn = (size_t)sizes[p2] > lo ? std::min(PIECE, (size_t)sizes[p2] - lo) : 0;
            if (sn) RB_NCCL(R.Send(static_cast<const char *>(send) + lo, sn, ncclUint8, p2, c->nc, st));
            if (rn) RB_NCCL(R.Recv(P.buf[pool_slot].as<char>() + ro + lo, rn, ncclUint8, p2, c->nc, st));
            ro += sizes[p2];
        }
        RB_NCCL(R.GroupEnd());
    }
    RB_HIP(hipStreamSynchronize(st));
    *recv = P.buf[pool_slot].p;
}

struct SlotView { void *p = nullptr; int64_t nbytes = 0; };
SlotView slot(rb_graph *g, int s, int64_t expect) {
    SlotView v;
    RB_CK(rb_shard_slot(g, s, &v.p, &v.nbytes));
    RB_REQUIRE(expect < 0 || v.nbytes == expect, "slot %d holds %lld bytes, expected %lld", s, (long long)v.nbytes, (long long)expect);
    return v;
}
int64_t sum(const int64_t *a, int n) { int64_t s = 0; for (int i = 0; i < n; ++i) s += a[i]; return s; }
void fill(Part &p, const void *send, const int64_t *counts, int G, int64_t unit) {
    p.send = send;
    for (int d = 0; d < G; ++d) p.sc[d] = counts[d] * unit;
}

// ---- one global sub-batch (rnabloom/sharded.py::ShardRank.substep, statement by statement) ----
void substep(rb_graph *g, rb_shard_comm *c, const rb_batch *b, int64_t first, int64_t n, uint32_t pos_bits, unsigned flags, uint64_t ordinal,
             bool have_next, int64_t nxt_first, int64_t nxt_n, int overlap, rb_add_stats *st_out) {
    const int G = g->shard_count, me = g->shard_rank;
    RB_REQUIRE(c->world == G, "communicator of %d ranks, graph of %d shards", c->world, G);
    hipStream_t st = g->stream;
    const int mode = (flags & RB_ADD_COUNT_IF_PRESENT) ? 2 /* M_COUNT_IF_PRESENT */ : 0 /* M_ADD */;
    const bool split = rb::shard_is_split(g);
    rb_add_stats stats;
    memset(&stats, 0, sizeof stats);
    const int64_t p0 = first + n * me / G, p1 = first + n * (me + 1) / G;
    int64_t d_c[MAX_WORLD], c_c[MAX_WORLD], pair_c[MAX_WORLD], rec_c[MAX_WORLD];
    Part pa[MAX_PARTS];
    // requests of this rank's runs, and what the other ranks ask of this rank's filter ranges
    int64_t o_dc[MAX_WORLD], o_cc[MAX_WORLD];
    void *o_didx, *o_dprobe, *o_cidx, *rpidx;
    int64_t np_bytes = 0;
    // the receive pool is reused by every exchange: what must outlive the next exchange is copied aside
    DevBuf &keep_pairs = g->comm_keep;
    if (split) {
        RB_CK(rb_shard_hash(g, b, first, n, p0, p1 - p0, ordinal, pos_bits, flags, rec_c, pair_c, &stats));
        const int64_t nrec = sum(rec_c, G), npair = sum(pair_c, G);
        fill(pa[0], slot(g, RB_SLOT_REC_KEYS, 8 * nrec).p, rec_c, G, 8);
        fill(pa[1], slot(g, RB_SLOT_REC_OCC, 4 * nrec).p, rec_c, G, 4);
        fill(pa[2], slot(g, RB_SLOT_PAIR_IDX, 8 * npair).p, pair_c, G, 8);
        a2a(c, me, pa, 3, nullptr, st);
        np_bytes = pa[2].rtotal;
        keep_pairs.reserve((size_t)std::max<int64_t>(np_bytes, 1));           // the pair probes wait for the serve phase
        if (np_bytes) RB_HIP(hipMemcpyAsync(keep_pairs.p, pa[2].recv, (size_t)np_bytes, hipMemcpyDeviceToDevice, st));
        RB_HIP(hipStreamSynchronize(st));
        RB_CK(rb_shard_group(g, pa[0].recv, pa[1].recv, pa[0].rtotal / 8, ordinal, pos_bits, flags, d_c, c_c));
        const int64_t nd = sum(d_c, G), ncq = sum(c_c, G);
        fill(pa[0], slot(g, RB_SLOT_DREQ_IDX, 8 * nd).p, d_c, G, 8);
        fill(pa[1], slot(g, RB_SLOT_DREQ_PROBE, 8 * nd).p, d_c, G, 8);
        fill(pa[2], slot(g, RB_SLOT_CREQ_IDX, 8 * ncq).p, c_c, G, 8);
        a2a(c, me, pa, 3, nullptr, st);
        rpidx = keep_pairs.p;
    } else {
        RB_CK(rb_shard_hash_group(g, b, first, n, p0, p1 - p0, ordinal, pos_bits, flags, d_c, c_c, pair_c, &stats));
        if (have_next && overlap == 2) RB_CK(rb_shard_hash_begin(g, b, nxt_first, nxt_n, ordinal + (uint64_t)n, pos_bits, flags));
        const int64_t nd = sum(d_c, G), ncq = sum(c_c, G), npair = sum(pair_c, G);
        fill(pa[0], slot(g, RB_SLOT_DREQ_IDX, 8 * nd).p, d_c, G, 8);
        fill(pa[1], slot(g, RB_SLOT_DREQ_PROBE, 8 * nd).p, d_c, G, 8);
        fill(pa[2], slot(g, RB_SLOT_CREQ_IDX, 8 * ncq).p, c_c, G, 8);
        fill(pa[3], slot(g, RB_SLOT_PAIR_IDX, 8 * npair).p, pair_c, G, 8);
        a2a(c, me, pa, 4, nullptr, st);
        np_bytes = pa[3].rtotal;
        rpidx = pa[3].recv;
    }
    o_didx = pa[0].recv; o_dprobe = pa[1].recv; o_cidx = pa[2].recv;
    for (int s = 0; s < G; ++s) { o_dc[s] = pa[0].rc[s] / 8; o_cc[s] = pa[2].rc[s] / 8; }
    const int64_t nd_in = pa[0].rtotal / 8, nc_in = pa[2].rtotal / 8;
    // serve: this rank's filter ranges answer (one byte per request)
    DevBuf &dreply = g->comm_dreply, &creply = g->comm_creply;
    dreply.reserve((size_t)std::max<int64_t>(nd_in, 1)); creply.reserve((size_t)std::max<int64_t>(nc_in, 1));
    RB_CK(rb_shard_serve(g, mode, o_didx, o_dprobe, nd_in, o_cidx, nc_in, rpidx, np_bytes / 8, dreply.p, creply.p));
    if (have_next && overlap == 2 && !split) RB_CK(rb_shard_hash_emit(g));
    fill(pa[0], dreply.p, o_dc, G, 1);
    fill(pa[1], creply.p, o_cc, G, 1);
    { const int64_t *known[2] = {d_c, c_c}; a2a(c, me, pa, 2, known, st); }
    // resolve: runs that own their counters alone finish here
    int64_t w_c[MAX_WORLD], ord_c[MAX_WORLD], nconf = 0, nedge = 0;
    RB_CK(rb_shard_resolve(g, mode, pa[0].recv, pa[1].recv, w_c, ord_c, &stats));
    if (have_next && overlap == 1 && !split) RB_CK(rb_shard_hash_begin(g, b, nxt_first, nxt_n, ordinal + (uint64_t)n, pos_bits, flags));
    {   // counter writes of the finished runs and, in the same exchange, the ordered-set questions; the answers come back by known counts
        const int64_t nw = sum(w_c, G), no = sum(ord_c, G);
        fill(pa[0], slot(g, RB_SLOT_W_IDX, 8 * nw).p, w_c, G, 8);
        fill(pa[1], slot(g, RB_SLOT_W_VAL, nw).p, w_c, G, 1);
        fill(pa[2], slot(g, RB_SLOT_ORD_IDX, 8 * no).p, ord_c, G, 8);
        a2a(c, me, pa, 3, nullptr, st);
        RB_CK(rb_shard_apply_writes(g, pa[0].recv, pa[1].recv, pa[0].rtotal / 8));
        int64_t o_oc[MAX_WORLD];
        for (int s2 = 0; s2 < G; ++s2) o_oc[s2] = pa[2].rc[s2] / 8;
        const int64_t n_ord = pa[2].rtotal / 8;
        DevBuf &oreply = g->comm_creply;                       // (the claim replies are consumed: rb_shard_resolve is over)
        oreply.reserve((size_t)std::max<int64_t>(n_ord, 1));
        RB_CK(rb_shard_order_serve(g, pa[2].recv, n_ord, oreply.p));
        fill(pa[0], oreply.p, o_oc, G, 1);
        { const int64_t *known[1] = {ord_c}; a2a(c, me, pa, 1, known, st); }
        RB_CK(rb_shard_order_finish(g, mode, pa[0].recv, &nconf, &nedge, &stats));
    }
    if (have_next && overlap == 1 && !split) RB_CK(rb_shard_hash_emit(g));
    // the (run, contested counter) edges of every rank and, in split mode, what the owners learnt for the cache replicas
    void *all_edges = nullptr;
    int64_t e_sizes[MAX_WORLD], u_sizes[MAX_WORLD];
    gather(c, me, slot(g, RB_SLOT_CONF_EDGES, 16 * nedge).p, 16 * nedge, 0, &all_edges, e_sizes, st);
    if (sum(e_sizes, G)) RB_CK(rb_shard_apply_tagged(g, all_edges, sum(e_sizes, G) / 16));
    if (split) {
        SlotView upd = slot(g, RB_SLOT_CACHE_UPD, -1);
        void *all_upd = nullptr;
        gather(c, me, upd.p, upd.nbytes, 1, &all_upd, u_sizes, st);
        {   // every rank's updates but this rank's own (its replica got them when they were made, rb_shard_resolve)
            int64_t before = 0, after = 0;
            for (int r = 0; r < G; ++r) { if (r < me) before += u_sizes[r]; else if (r > me) after += u_sizes[r]; }
            if (before) RB_CK(rb_shard_cache_apply(g, all_upd, before / 16));
            if (after) RB_CK(rb_shard_cache_apply(g, static_cast<const char *>(all_upd) + before + u_sizes[me], after / 16));
        }
        // look-ahead: this rank's slice of the NEXT sub-batch is walked against the cache as it stands now, on the producer stream,
        // beside the conflict phases and exchanges below (RB_SHARD_OVERLAP=0: off)
        if (have_next && overlap >= 1) {
            const int64_t q0 = nxt_first + nxt_n * me / G, q1 = nxt_first + nxt_n * (me + 1) / G;
            RB_CK(rb_shard_hash_begin_split(g, b, nxt_first, nxt_n, q0, q1 - q0, ordinal + (uint64_t)n, pos_bits, flags));
        }
    }
    const int64_t e_total = sum(e_sizes, G);
    if (e_total) {
        int64_t run_c[MAX_WORLD], op_c[MAX_WORLD], cw_c[MAX_WORLD];
        const int64_t e_max = *std::max_element(e_sizes, e_sizes + G);
        RB_CK(rb_shard_conflict_route(g, all_edges, e_total / 16, (int64_t)G * (e_max / 16), run_c, op_c, &stats));
        fill(pa[0], slot(g, RB_SLOT_CONF_RUNS, 24 * sum(run_c, G)).p, run_c, G, 24);
        fill(pa[1], slot(g, RB_SLOT_CONF_OPS, 4 * sum(op_c, G)).p, op_c, G, 4);
        a2a(c, me, pa, 2, nullptr, st);
        RB_CK(rb_shard_conflict_replay(g, pa[0].recv, pa[0].rtotal / 24, pa[1].recv, pa[1].rtotal / 4, cw_c));
        const int64_t ncw = sum(cw_c, G);
        fill(pa[0], slot(g, RB_SLOT_CW_IDX, 8 * ncw).p, cw_c, G, 8);
        fill(pa[1], slot(g, RB_SLOT_CW_VAL, ncw).p, cw_c, G, 1);
        a2a(c, me, pa, 2, nullptr, st);
        RB_CK(rb_shard_apply_writes(g, pa[0].recv, pa[1].recv, pa[0].rtotal / 8));
    }
    if (st_out) {
        st_out->reads += stats.reads; st_out->kmers += stats.kmers; st_out->pairs += stats.pairs; st_out->distinct += stats.distinct;
        st_out->conflict_ops += stats.conflict_ops; st_out->sorted_kmers += stats.sorted_kmers;
    }
}
}  // namespace

extern "C" {

int rb_shard_comm_unique_id(void *out128) {
    return guarded([&] {
        RB_REQUIRE(out128, "rb_shard_comm_unique_id: null argument");
        need_rccl();
        ncclUniqueId id;
        RB_NCCL(rccl().GetUniqueId(&id));
        static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
        memcpy(out128, &id, sizeof id);
    });
}

int rb_shard_comm_create_rccl(const void *id128, int rank, int world, int device, rb_shard_comm **out) {
    return guarded([&] {
        RB_REQUIRE(id128 && out && world >= 1 && world <= MAX_WORLD && rank >= 0 && rank < world, "rb_shard_comm_create_rccl: bad argument");
        need_rccl();
        RB_HIP(hipSetDevice(device));
        rb_shard_comm *c = new rb_shard_comm();
        c->world = world; c->is_rccl = true; c->rank = rank; c->device = device;
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        ncclResult_t r = rccl().CommInitRank(&c->nc, world, id, rank);
        if (r != ncclSuccess) { delete c; set_error("ncclCommInitRank failed: %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "?"); throw HipError{RB_ERR_HIP}; }
        *out = c;
    });
}

int rb_shard_comm_create_loopback(int world, rb_shard_comm **out) {
    return guarded([&] {
        RB_REQUIRE(out && world >= 1 && world <= MAX_WORLD, "rb_shard_comm_create_loopback: bad argument");
        rb_shard_comm *c = new rb_shard_comm();
        c->world = world;
        *out = c;
    });
}

// A small all-to-all and all-gather with known bytes through the communicator's own transport (what rb_shard_add_range uses):
// rank `me` sends (me * 7 + p * 3 + 1) * 1000 + big bytes to rank p, byte i = (me * 31 + p * 17 + i) & 255, and checks what arrives.
// bench.py runs it before the first step with RCCL and falls back to the torch.distributed driver when it fails.
// big_bytes == -2 (fault injection for the hub's tests): run the whole exchange with big_bytes = 0, wait a moment, then fail AFTER the
// last barrier — the peers have left this call by then and may already wait in the next one.
int rb_shard_comm_selftest(rb_shard_comm *c, int me, int device, int64_t big_bytes) {
    const bool fail_late = big_bytes == -2;
    if (fail_late) big_bytes = 0;
    DevBuf sendbuf;
    struct Rel { DevBuf &b; ~Rel() { b.release(); } } rel{sendbuf};
    InCall in_call(c, me);
    int rc = guarded([&] {
        RB_REQUIRE(c && me >= 0 && me < c->world && big_bytes >= 0, "rb_shard_comm_selftest: bad argument");
        RB_HIP(hipSetDevice(device));
        hipStream_t st = nullptr;
        RB_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        struct StreamDrop { hipStream_t s; ~StreamDrop() { (void)hipStreamDestroy(s); } } drop{st};
        const int G = c->world;
        auto len = [&](int from, int to) { return (int64_t)(from * 7 + to * 3 + 1) * 1000 + big_bytes; };
        Part pa[1];
        std::vector<uint8_t> host;
        int64_t tot = 0;
        for (int p2 = 0; p2 < G; ++p2) { pa[0].sc[p2] = len(me, p2); tot += pa[0].sc[p2]; }
        host.resize((size_t)tot);
        int64_t o = 0;
        for (int p2 = 0; p2 < G; ++p2)
            for (int64_t i = 0; i < pa[0].sc[p2]; ++i) host[(size_t)(o++)] = (uint8_t)((me * 31 + p2 * 17 + i) & 255);
        sendbuf.reserve((size_t)tot);
        RB_HIP(hipMemcpyAsync(sendbuf.p, host.data(), (size_t)tot, hipMemcpyHostToDevice, st));
        pa[0].send = sendbuf.p;
        try {
            a2a(c, me, pa, 1, nullptr, st);
        } catch (const HubFailed &) { set_error("rb_shard_comm_selftest: another rank of the loopback hub failed"); throw HipError{RB_ERR_STATE}; }
        std::vector<uint8_t> got((size_t)pa[0].rtotal);
        RB_HIP(hipMemcpy(got.data(), pa[0].recv, got.size(), hipMemcpyDeviceToHost));
        o = 0;
        for (int src = 0; src < G; ++src) {
            RB_REQUIRE(pa[0].rc[src] == len(src, me), "rb_shard_comm_selftest: %lld bytes from rank %d, expected %lld", (long long)pa[0].rc[src], src, (long long)len(src, me));
            for (int64_t i = 0; i < pa[0].rc[src]; ++i, ++o)
                RB_REQUIRE(got[(size_t)o] == (uint8_t)((src * 31 + me * 17 + i) & 255), "rb_shard_comm_selftest: byte %lld from rank %d is wrong", (long long)i, src);
        }
        // all-gather of 1000 * (me + 1) bytes per rank
        void *all = nullptr;
        int64_t sizes[MAX_WORLD];
        try {
            gather(c, me, sendbuf.p, (int64_t)1000 * (me + 1), 0, &all, sizes, st);
        } catch (const HubFailed &) { set_error("rb_shard_comm_selftest: another rank of the loopback hub failed"); throw HipError{RB_ERR_STATE}; }
        int64_t gt = 0;
        for (int src = 0; src < G; ++src) { RB_REQUIRE(sizes[src] == (int64_t)1000 * (src + 1), "rb_shard_comm_selftest: gathered size of rank %d is wrong", src); gt += sizes[src]; }
        got.resize((size_t)gt);
        RB_HIP(hipMemcpy(got.data(), all, got.size(), hipMemcpyDeviceToHost));
        o = 0;
        for (int src = 0; src < G; ++src)
            for (int64_t i = 0; i < sizes[src]; ++i, ++o)
                RB_REQUIRE(got[(size_t)o] == (uint8_t)((src * 31 + 0 * 17 + i) & 255), "rb_shard_comm_selftest: gathered byte %lld of rank %d is wrong", (long long)i, src);
        if (fail_late) {
            struct timespec ts = {0, 300 * 1000 * 1000};
            nanosleep(&ts, nullptr);
            RB_REQUIRE(false, "rb_shard_comm_selftest: injected failure after the last barrier");
        }
    });
    if (rc != RB_OK) in_call.failed();
    return rc;
}

int rb_shard_comm_destroy(rb_shard_comm *c) {
    if (!c) return RB_OK;
    if (c->is_rccl && c->nc && rccl().CommDestroy) (void)rccl().CommDestroy(c->nc);
    for (auto &p : c->pool) {
        for (auto &b : p.buf) b.release();
        p.cnt_dev.release();
        if (p.cnt_host) (void)hipHostFree(p.cnt_host);
    }
    delete c;
    return RB_OK;
}

int rb_shard_add_range(rb_graph *g, rb_shard_comm *c, const rb_batch *b, int64_t first, int64_t n, unsigned flags, int64_t reads_per_substep,
                       uint32_t pos_bits, uint64_t ordinal0, rb_add_stats *stats) {
    if (stats) memset(stats, 0, sizeof *stats);
    InCall in_call(c, g && g->shard ? g->shard_rank : -1);
    int rc = guarded([&] {
        RB_REQUIRE(g && g->shard && c && b && n >= 0 && reads_per_substep > 0, "rb_shard_add_range: bad argument");
        rb::WriteLock wl(g->rw);          // a mutator like every other insert: queries on this shard handle wait (rb_graph_kmers, rb_shard_query_*)
        RB_HIP(hipSetDevice(g->p.device));
        const int overlap = getenv("RB_SHARD_OVERLAP") ? atoi(getenv("RB_SHARD_OVERLAP")) : 1;
        // cold start (first insert into cleared filters): short sub-batches first, doubling up to the full size
        int64_t cur = reads_per_substep;
        // (round 6 measured the ramp's start at 8 ranks: reads / 16 -> 64.9 ms per rank, / 64 (this) -> 66.3, / 4 -> 77.4: the short sub-batches are
        // mostly launch overhead, but starting higher sorts more of the first windows unfiltered)
        if (ordinal0 == 0 && !getenv("RB_NO_RAMP") && reads_per_substep >= 64 * 1024) cur = std::max<int64_t>(reads_per_substep / 64, 1024);
        std::vector<int64_t> cuts{0};
        for (int64_t a = 0; a < n;) { a = std::min(n, a + cur); cuts.push_back(a); cur = std::min(reads_per_substep, cur * 2); }
        try {
            for (size_t i = 0; i + 1 < cuts.size(); ++i) {
                const int64_t a = cuts[i], e = cuts[i + 1];
                const bool have_next = i + 2 < cuts.size();
                substep(g, c, b, first + a, e - a, pos_bits, flags, ordinal0 + (uint64_t)a, have_next, first + e, have_next ? cuts[i + 2] - e : 0, overlap, stats);
            }
            if (flags & RB_ADD_STORE_READ_PAIRS) {   // the ranks' read-pair accumulation copies -> the owners' shards: one all-to-all of G pieces
                Part pa[1];
                void *send = nullptr;
                int64_t cnt[MAX_WORLD];
                RB_CK(rb_shard_pairs_flush_begin(g, &send, cnt));
                fill(pa[0], send, cnt, g->shard_count, 1);
                a2a(c, g->shard_rank, pa, 1, nullptr, g->stream);
                RB_CK(rb_shard_pairs_flush_end(g, pa[0].recv, pa[0].rc));
            }
        } catch (const HubFailed &) {
            set_error("rb_shard_add_range: another rank of the loopback hub failed");
            throw HipError{RB_ERR_STATE};
        }
    });
    if (rc != RB_OK) in_call.failed();
    return rc;
}

}  // extern "C"
