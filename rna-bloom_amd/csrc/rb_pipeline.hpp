// rb_pipeline.hpp — device helpers, per-graph state and host utilities shared by the single-GPU
// engine (rb_graph.hip) and the sharded engine (rb_shard.hip).
#pragma once
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "rb_internal.hpp"
#include "rb_kernels.hpp"

namespace rb {

constexpr int TPB = 256;
constexpr size_t DEVCTR_BYTES = 4096;   // small device-side counter block per graph
inline unsigned blocks_for(int64_t n, int tpb = TPB) { return (unsigned)((n + tpb - 1) / tpb); }

struct BitFilter {
    uint32_t *bits = nullptr;
    int64_t size = 0, nbytes = 0;
    size_t alloc = 0;
    int num_hash = 0;
    Mod mod{1, 0, 0};
    int64_t lo = 0, hi = 0;   // index range held locally ([0,size) unless sharded)
    // seen-pair cache of a paired-k-mer filter (k_pairs_reads, rb_graph.hip): 2^seen_log2b buckets of 16 pair hashes whose bits are known
    // to be set in `bits`.  Not part of the filter's state: whatever clears or replaces `bits` clears it (zero_bits / seen_reset).
    unsigned long long *seen = nullptr;
    uint32_t seen_log2b = 0;
};
// what the pair walker gets of it (by value)
struct PairSeen {
    unsigned long long *tab;
    uint32_t mask;            // buckets - 1
    uint32_t amask;           // an m-mer is an anchor when (mixed hash & amask) == 0: 3 = one in four (RB_PAIR_SEEN_ANCHOR=<log2 density>)
    unsigned long long *dbg;  // RB_DEBUG: [0] pairs the cache did not know, [1] bucket fetches (nullptr: not counted)
};

// everything a kernel needs to address the filters (passed by value)
struct FilterView {
    uint32_t *dbg; Mod dbg_mod; int dbg_h;
    uint8_t *cbf;  Mod cbf_mod; int cbf_h;
    uint64_t kmul;        // k * multiSeed
    uint64_t seed;        // rng seed
    uint64_t ordinal0;    // op ordinal of occurrence value 0
    uint32_t pos_bits;    // occurrence value = (read_rel << pos_bits) | pos
    Npf npf;              // no-op prefilter cache (tab == nullptr: off)
    // minimizer-bucketed variant (k <= 31 insert path): lookups in the window-hash kernel, stores here need
    // the k-mer's bases to find its bucket — the read batch view of the sub-batch being retired
    Mpf mpf;
    const uint64_t *seq_codes;   // packed reads (nullptr: no sequence context, e.g. rb_graph_apply)
    const uint32_t *seq_woff;
    uint32_t seq_wpr;            // words per read when every read of the batch has the same number (no offset lookup then), else 0
    uint32_t seq_first;          // read index of occurrence value 0's read
    int k;
};
// first packed word of read r of the batch view
__device__ __forceinline__ uint64_t seq_word0(const FilterView &fv, uint32_t r) {
    return fv.seq_wpr ? (uint64_t)r * fv.seq_wpr : (uint64_t)fv.seq_woff[r];
}
// remember a k-mer's counter exponent in whichever cache the insert path uses; occ = any occurrence of it.
// Returns whether the cache changed (the sharded engine broadcasts only the stores that did: the replicas are alike, so a store
// that changes nothing on the owner's replica changes nothing anywhere — e.g. the k-mer whose bucket is full of hotter ones tries
// again every sub-batch).
__device__ __forceinline__ bool cache_store(const FilterView &fv, uint64_t h0, uint32_t occ, uint32_t s) {
    if (fv.mpf.tab) {
        if (!fv.seq_codes) return false;
        const uint32_t r = fv.seq_first + (occ >> fv.pos_bits), p = occ & ((1u << fv.pos_bits) - 1u);
        const uint32_t ord = window_min_order(fv.seq_codes + seq_word0(fv, r), p, (uint32_t)fv.k, fv.mpf.m);
        return mpf_store(fv.mpf, mpf_bucket(fv.mpf, ord), h0, s);
    } else if (fv.npf.tab)
        return npf_store(fv.npf, h0, s);
    return false;
}
__device__ __forceinline__ bool cache_on(const FilterView &fv) { return fv.mpf.tab ? fv.seq_codes != nullptr : fv.npf.tab != nullptr; }

// open-addressing table slot: key (empty = ~0) + 64-bit payload (identity of atomicMin = ~0)
struct Slot { unsigned long long key; unsigned long long val; };

__device__ __forceinline__ uint64_t slot_of(uint64_t key, uint32_t log2cap) {
    return (key * 0x9E3779B97F4A7C15ull) >> (64u - log2cap);
}
__device__ __forceinline__ Slot *table_insert(Slot *t, uint32_t log2cap, uint64_t key) {
    const uint64_t mask = (1ull << log2cap) - 1ull;
    uint64_t s = slot_of(key, log2cap);
    for (;;) {
        unsigned long long cur = __hip_atomic_load(&t[s].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return &t[s];
        if (cur == ~0ull) {
            unsigned long long old = atomicCAS(&t[s].key, ~0ull, (unsigned long long)key);
            if (old == ~0ull || old == key) return &t[s];
        }
        s = (s + 1ull) & mask;
    }
}
__device__ __forceinline__ const Slot *table_find(const Slot *t, uint32_t log2cap, uint64_t key) {
    const uint64_t mask = (1ull << log2cap) - 1ull;
    uint64_t s = slot_of(key, log2cap);
    for (;;) {
        unsigned long long cur = t[s].key;
        if (cur == key) return &t[s];
        if (cur == ~0ull) return nullptr;
        s = (s + 1ull) & mask;
    }
}

// op kinds of the counting-Bloom state machine
enum : uint32_t { K_INC = 0, K_INC_IF_POS = 1, K_INC_IF_ZERO = 2 };
// pipeline modes
enum : int { M_ADD = 0, M_ADD_IF_ABSENT = 1, M_COUNT_IF_PRESENT = 2, M_COUNT_ONLY = 4 };

// status word per distinct k-mer: bits 0..7 premask (bit j: probe j was set before the batch),
// bit 8 all_pre, bits 12..13 kind of first op, bits 14..15 kind of the remaining ops, bit 16 conflict
constexpr uint32_t ST_ALLPRE = 1u << 8;

// CountingBloomFilter.increment(long[]) :170-194 on a register copy of the cbf_h bytes.
// c[j] mirrors counts[idx_j]; duplicated indices stay consistent because both copies move together.
__device__ __forceinline__ void cbf_step(uint32_t *c, int h, uint32_t kind, uint32_t rnd31) {
    uint32_t mn = c[0];
    for (int j = 1; j < h; ++j) mn = c[j] < mn ? c[j] : mn;
    if (kind == K_INC_IF_POS && mn == 0u) return;    // addCountIfPresent :424-428
    if (kind == K_INC_IF_ZERO && mn != 0u) return;   // addIfAbsent else-branch :419-421
    uint32_t up = minifloat_inc(mn, rnd31);
    if (up != mn)
        for (int j = 0; j < h; ++j) if (c[j] == mn) c[j] = up;
}
__device__ __forceinline__ uint32_t occ_rnd(const FilterView &fv, uint32_t v) {
    return rng31(fv.seed, fv.ordinal0 + (uint64_t)(v >> fv.pos_bits), v & ((1u << fv.pos_bits) - 1u));
}

// Random-draw "strength" of every occurrence, in sorted order: the number of trailing zero bits of
// its 31-bit draw (capped at 15).  MiniFloat.increment at byte b >= 16 succeeds iff
// rnd % 2^s == 0 with s = (b>>3)-1 <= 14, i.e. iff strength >= s — so the per-run state machines
// only compare bytes and never evaluate the generator inside their sequential loops.
static __global__ void k_strength(FilterView fv, const uint32_t *__restrict__ vals, size_t n, uint8_t *__restrict__ tz) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = occ_rnd(fv, vals[i]) | 0x8000u;          // bit 15 caps the count at 15
    tz[i] = (uint8_t)(__ffs((int)r) - 1);
}
// first index in [pos,end) whose strength byte is >= s (s in 1..15), or end.  8 bytes per load.
__device__ __forceinline__ uint32_t next_success(const uint8_t *__restrict__ tz, uint32_t pos, uint32_t end, uint32_t s) {
    const uint64_t add = (uint64_t)(0x80u - s) * 0x0101010101010101ull;
    while (pos < end) {
        const uint32_t a = pos & ~7u;
        uint64_t w = *reinterpret_cast<const uint64_t *>(tz + a);      // tz is allocated with 8 bytes of slack
        uint64_t hit = (w + add) & 0x8080808080808080ull;              // bytes are <= 15: no carries
        hit &= ~0ull << (8u * (pos - a));                              // ignore bytes below pos
        if (hit) {
            const uint32_t q = a + ((uint32_t)__ffsll((long long)hit) - 1u) / 8u;
            return q < end ? q : end;
        }
        pos = a + 8u;
    }
    return end;
}
// all `ops` increments of one run applied to the register copy c[] of its counters, in order.
// kind0 applies to the first op, kind to the others; tz = strengths of the ops (tz[0] = first op).
__device__ __forceinline__ void run_ops(uint32_t *c, int h, uint32_t kind0, uint32_t kind, const uint8_t *__restrict__ tz,
                                        uint32_t base, uint32_t ops) {
    auto minimum = [&]() { uint32_t mn = c[0]; for (int j = 1; j < h; ++j) mn = c[j] < mn ? c[j] : mn; return mn; };
    auto bump = [&](uint32_t mn) { for (int j = 0; j < h; ++j) if (c[j] == mn) c[j] = mn + 1u; };
    uint32_t i = 0;
    {   // first op (its kind may differ)
        const uint32_t mn = minimum();
        const bool gate = !((kind0 == K_INC_IF_POS && mn == 0u) || (kind0 == K_INC_IF_ZERO && mn != 0u));
        if (gate && mn < 127u && (mn < 16u || tz[base] >= (mn >> 3) - 1u)) bump(mn);
        i = 1;
    }
    while (i < ops) {
        const uint32_t mn = minimum();
        if (mn >= 127u) break;                                    // saturated
        if (kind == K_INC_IF_POS && mn == 0u) break;              // stays zero for the rest of the run
        if (kind == K_INC_IF_ZERO && mn != 0u) break;             // stays positive
        if (mn < 16u) { bump(mn); ++i; continue; }                // deterministic region
        const uint32_t q = next_success(tz, base + i, base + ops, (mn >> 3) - 1u);
        if (q >= base + ops) break;
        bump(mn);
        i = q - base + 1u;
    }
}

// Counter bytes live in 0..127 (MiniFloat saturates at Byte.MAX_VALUE, R/util/MiniFloat.java:32), so
// bit 7 of a counting-Bloom byte is free.  During a sub-batch it serves as a "claimed by a k-mer of
// this sub-batch" marker: the atomicOr that claims a counter also returns its value, and a k-mer that
// finds the marker already set knows it shares the counter with another k-mer (=> ordered replay).
constexpr uint32_t CLAIM = 0x80u;
__device__ __forceinline__ uint32_t cbf_claim(uint8_t *cbf, uint64_t idx) {       // returns old byte
    uint32_t *w = reinterpret_cast<uint32_t *>(cbf) + (idx >> 2);
    const uint32_t sh = 8u * (uint32_t)(idx & 3u);
    return (atomicOr(w, CLAIM << sh) >> sh) & 0xFFu;
}
__device__ __forceinline__ void cbf_release(uint8_t *cbf, uint64_t idx) {
    uint32_t *w = reinterpret_cast<uint32_t *>(cbf) + (idx >> 2);
    atomicAnd(w, ~(CLAIM << (8u * (uint32_t)(idx & 3u))));
}

// status word per distinct run: bits 0..7 premask, bit 8 all_pre, bit 9 claimed counters,
// bit 10 saw a foreign claim, bits 12..13 kind of first op, 14..15 kind of the other ops
constexpr uint32_t ST_CLAIMED = 1u << 9, ST_FOREIGN = 1u << 10;
// bit 11: single-occurrence run of a k-mer that is not in dbgbf yet — it counts only if an earlier probe
// of the sub-batch set all its missing bits (a false positive in the making), so its counters are
// claimed late, by k_late_claim, and only in that rare case; bit 16: that case applies
constexpr uint32_t ST_LATE = 1u << 11, ST_LATE_FOUND = 1u << 16;
// outcome of the resolve stage (lists are built from these flags by stream compaction: a single
// shared atomic cursor saturates at ~88 M increments/s and would dominate the stage)
constexpr uint32_t RUN_CONFLICT = 1u << 17, RUN_RELEASE = 1u << 18, RUN_WRITES = 1u << 19, RUN_HEAVY = 1u << 20;


template <typename F> int guarded(F &&f) {
    try { f(); return RB_OK; }
    catch (const HipError &e) { return e.code; }
    catch (const std::bad_alloc &) { set_error("host allocation failed"); return RB_ERR_NOMEM; }
}
void alloc_bits(BitFilter &f, int64_t bits, int num_hash, int64_t lo, int64_t hi);
void free_bits(BitFilter &f);
// shared by the translation units of the single-GPU engine (rb_graph.hip: pipeline; rb_capi.hip: entry points; rb_query.hip: queries)
const char *last_error_text();                      // this thread's message of the last failing call (rb_last_error)
BitFilter *bit_filter(rb_graph *g, int which);      // RB_DBGBF / RB_RPKBF / RB_FPKBF -> the handle's filter, nullptr otherwise
uint64_t *upload_h0(rb_graph *g, DevBuf &buf, const uint64_t *h0, size_t n, hipStream_t st = nullptr);
void fast_zero(void *p, size_t bytes, hipStream_t s);
void add_reads_streamed(rb_graph *g, const char *seq, const char *qual, const int64_t *offsets, int64_t n_reads, int min_base_qual, int64_t piece_bases,
                        unsigned flags, rb_add_stats *stats);       // rb_packed.hip: host ASCII reads as ONE insert over a batch that is still being uploaded and encoded
void add_range(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, unsigned flags, rb_add_stats *stats);      // the stage-1 insert of reads [first, first + n)
void run_pipeline(rb_graph *g, size_t N, int mode, uint64_t ordinal0, uint32_t pos_bits, rb_add_stats *stats);       // the consumer half on N records in keys0 / vals0
void launch_pairs_reads(rb_graph *g, const rb_batch *b, int64_t w0, int64_t nw, int mode_hash, const BitFilter &f, int dist,
                        uint32_t min_len, bool if_present, const uint32_t *chunk_off, uint64_t *out_idx, unsigned long long *pc, hipStream_t st);
__global__ void k_bits_add(uint32_t *bits, Mod mod, int num_hash, uint64_t kmul, const uint64_t *__restrict__ h0, size_t n);   // rb_query.hip
void alloc_pair_seen(BitFilter &f);                 // seen-pair cache for a paired-k-mer filter (RB_PAIR_SEEN=0: none)
void seen_reset(BitFilter &f, hipStream_t s);       // after anything that clears or replaces f.bits
// one wavefront per high-multiplicity run: lanes fetch 64 occurrences at a time, compute each
// one's random draw, and the increment chain hops from success to success with ballots
static __global__ void __launch_bounds__(64) k_cbf_heavy(FilterView fv, const uint64_t *__restrict__ uniq,
                            const uint32_t *__restrict__ counts, const uint32_t *__restrict__ starts,
                            const uint32_t *__restrict__ vals, const uint32_t *__restrict__ status,
                            const uint32_t *__restrict__ nops, const uint64_t *__restrict__ cvals,
                            const uint8_t *__restrict__ tz, const uint32_t *__restrict__ heavy_list,
                            const uint32_t *__restrict__ counters, uint64_t *__restrict__ cfinal,
                            uint8_t *__restrict__ cache_upd = nullptr /* sharded engine: exponent to broadcast */) {
    const uint32_t n_heavy = counters[0];
    const uint32_t lane = threadIdx.x;
    for (uint32_t hi = blockIdx.x; hi < n_heavy; hi += gridDim.x) {
        const uint32_t d = heavy_list[hi];
        const uint64_t h0 = uniq[d];
        const uint32_t ops = nops[d];
        const uint32_t st = status[d];
        const uint64_t cv = cvals[d];
        uint64_t idx[RB_MAX_HASH];
        uint32_t c[RB_MAX_HASH];
        for (int j = 0; j < fv.cbf_h; ++j) {
            idx[j] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
            c[j] = (uint32_t)(cv >> (8 * j)) & 0xFFu;
        }
        const uint32_t base = starts[d] + counts[d] - ops;
        const uint32_t krest = (st >> 14) & 3u;
        uint32_t done = 0;
        {   // the first op may have its own kind
            uint32_t mn = c[0];
            for (int j = 1; j < fv.cbf_h; ++j) mn = c[j] < mn ? c[j] : mn;
            const uint32_t k0 = (st >> 12) & 3u;
            const bool gate = !((k0 == K_INC_IF_POS && mn == 0u) || (k0 == K_INC_IF_ZERO && mn != 0u));
            if (gate && mn < 127u && (mn < 16u || tz[base] >= (mn >> 3) - 1u))
                for (int j = 0; j < fv.cbf_h; ++j) if (c[j] == mn) c[j] = mn + 1u;
            done = 1;
        }
        while (done < ops) {
            uint32_t mn = c[0];
            for (int j = 1; j < fv.cbf_h; ++j) mn = c[j] < mn ? c[j] : mn;
            if (mn >= 127u) break;                                   // saturated: nothing changes any more
            if (krest == K_INC_IF_POS && mn == 0u) break;            // stays zero for ever
            if (krest == K_INC_IF_ZERO && mn != 0u) break;           // stays positive for ever
            if (mn < 16u) {                                          // deterministic region: one step
                cbf_step(c, fv.cbf_h, krest, 0u);
                ++done;
                continue;
            }
            // probabilistic region: every lane examines 8 pending occurrences (one 8-byte load)
            const uint32_t shift = (mn >> 3) - 1u;
            const uint32_t pos = base + done, end = base + ops;
            const uint32_t a0 = pos & ~7u;                               // aligned start of the 512-byte window
            const uint32_t mine = a0 + 8u * lane;
            uint32_t hit = 0xFFFFFFFFu;                                  // absolute index of my first success
            if (mine < end) {
                uint64_t w = *reinterpret_cast<const uint64_t *>(tz + mine);
                uint64_t m = (w + (uint64_t)(0x80u - shift) * 0x0101010101010101ull) & 0x8080808080808080ull;
                if (mine < pos) m &= ~0ull << (8u * (pos - mine));        // bytes before the cursor
                if (m) { uint32_t q = mine + ((uint32_t)__ffsll((long long)m) - 1u) / 8u; if (q < end) hit = q; }
            }
            const unsigned long long win = __ballot(hit != 0xFFFFFFFFu);
            if (!win) { done = (a0 + 512u) - base; continue; }
            const uint32_t first_lane = (uint32_t)__ffsll((long long)win) - 1u;
            const uint32_t q = __shfl(hit, (int)first_lane, 64);
            cbf_step(c, fv.cbf_h, krest, 0u);                        // rnd 0 always succeeds
            done = q - base + 1u;
        }
        if (lane == 0) {
            if (cfinal) {                        // sharded engine: the counters live on other ranks
                uint64_t out = 0;
                for (int j = 0; j < fv.cbf_h; ++j) out |= (uint64_t)c[j] << (8 * j);
                cfinal[d] = out;
            } else
                for (int j = 0; j < fv.cbf_h; ++j) {
                    // an unchanged counter may be shared with a run that can reach it (k_cs_writers): leave its value alone
                    if (c[j] == ((uint32_t)(cv >> (8 * j)) & 0xFFu)) cbf_release(fv.cbf, idx[j]);
                    else fv.cbf[idx[j]] = (uint8_t)c[j];                             // clears the claim mark too
                }
            if (cache_on(fv) && (st & ST_ALLPRE)) {   // remember how hard this k-mer has become to increment
                uint32_t mn = c[0];
                for (int j = 1; j < fv.cbf_h; ++j) mn = c[j] < mn ? c[j] : mn;
                if (mn >= 16u) { if (cache_store(fv, h0, vals[starts[d]], cache_exp(mn)) && cache_upd) cache_upd[d] = (uint8_t)cache_exp(mn); }
            }
        }
    }
}

inline uint32_t log2_ceil(uint64_t x) { uint32_t l = 0; while ((1ull << l) < x) ++l; return l; }

}  // namespace rb

// ------------------------------------------------------------------ graph object ----
struct ShardState;
struct rb_trav;
using rb::BitFilter; using rb::FilterView; using rb::Mod; using rb::DevBuf; using rb::kmul_of;
// Scratch + stream of one in-flight query.  The reference's stage-2 workers call contains / getCount / getKmers /
// getSuccessors on ONE graph from T threads (R/RNABloom.java, e.g. :1984-2114), so every query call leases a context of
// its own; inserts and everything else that changes a filter take the handle exclusively (rb_graph::rw).
struct rb_query_ctx {
    hipStream_t st = nullptr;
    rb::DevBuf b0, b1, b2, b3;
};
struct rb_graph {
    // one handle, many threads: queries share the handle (shared lock + a leased context each), mutators own it
    std::shared_mutex rw;
    std::mutex qm;
    std::condition_variable qcv;
    std::vector<rb_query_ctx *> qfree;
    int qmade = 0;
    static constexpr int kMaxQueryCtx = 32;
    // sharded mode (rb_shard.hip): this handle owns index range [lo,hi) of every filter
    int shard_rank = 0, shard_count = 1;
    ShardState *shard = nullptr;
    rb_trav *trav = nullptr;          // a traversal in progress on a sharded graph (rb_shard_trav_*, rb_query.hip)
    rb_graph_params p{};
    int k = 0, H = 0;
    bool stranded = false;
    BitFilter dbg, rpk, fpk;
    uint8_t *cbf = nullptr;
    int64_t cbf_size = 0;      // global number of counters
    int64_t cbf_lo = 0, cbf_hi = 0;   // counters held locally ([0,cbf_size) unless sharded)
    size_t cbf_alloc = 0;
    Mod cbf_mod{1, 0, 0};
    int cbf_h = 0;
    int read_d = -1, frag_d = -1;
    uint64_t ordinal = 0;
    int64_t max_batch_kmers = 0;
    int sort_begin_bit = 28;
    uint32_t light_ops = 96;
    uint32_t small_ops = 32;             // a component of at most this many ops is replayed by one lane (k_conf_replay_small), a larger one by a wavefront; RB_SMALL_COMPONENT_OPS
    hipStream_t stream = nullptr;    // consumer stream: everything that touches the filters
    hipStream_t stream2 = nullptr;   // producer stream: hashing + grouping of the NEXT sub-batch (scratch only)
    hipStream_t stream3 = nullptr;   // side stream of the producer: the paired-k-mer walker (rpkbf only) beside the window walk
    // grouped sub-batch, double buffered so that grouping of sub-batch i+1 overlaps the filter stages of i
    struct GroupSlot { DevBuf keys1, valsT, vals1, tz, uniq, counts, starts, brun, bnr /* swept stage: run slots per index range */; uint32_t sweep_T = 0, n_main = 0; size_t N = 0; uint32_t D = 0; int flags = 0; uint32_t live = 0; int bucket_target = 0; /* what group_enqueue planned with */ };
    GroupSlot slots[2];
    int cur = 0;
    DevBuf &keys1() { return slots[cur].keys1; }
    DevBuf &vals1() { return slots[cur].vals1; }
    DevBuf &tz() { return slots[cur].tz; }
    DevBuf &uniq() { return slots[cur].uniq; }
    DevBuf &counts() { return slots[cur].counts; }
    DevBuf &starts() { return slots[cur].starts; }
    DevBuf temp2, devctr2, pairs_ctr;
    DevBuf sw_st, sw_temp;               // swept Bloom-bit stage: what the probes found, partition scratch
    int pf_streak = 0, pf_skip_left = 0;   // sub-batches in a row the prefilter kept (nearly) everything of; sub-batches left to go without it
    float last_present_frac = 0.0f;      // share of the last sub-batch's runs whose Bloom bits were all set before it
    // no-op prefilter
    DevBuf npf, chunk_mask, npf_tot, wstate;
    uint32_t npf_log2 = 0;
    DevBuf mpf;                         // minimizer-bucketed cache (single-GPU k <= 31 insert path)
    uint32_t mpf_log2b = 0, mpf_m = 0;
    const uint64_t *seq_codes = nullptr;   // read batch view of the sub-batch being retired (add_range sets it)
    const uint32_t *seq_woff = nullptr;
    uint32_t seq_wpr = 0;
    uint32_t seq_first = 0;
    uint64_t *group_in_keys = nullptr;   // one-shot: the next group_enqueue takes its records from here instead of keys0 / vals0 (clobbered;
    uint32_t *group_in_vals = nullptr;   // the sharded engine groups the records it received where they arrived)
    uint32_t occ_bits = 32;       // occurrence ids of the sub-batch in flight are below 2^occ_bits (the conflict sort skips the bits above)
    bool use_mpf = false;
    // scratch (grow-only)
    DevBuf chunk_cnt, chunk_off, keys0, vals0, status, nops, temp,
        ftable, ctable, heavy, confk, conf_sizes, conf_off, opk0, opk1, opv0, opv1, label, kk0, kk1, biglist, cvals, foreign, devctr, qbuf0, qbuf1, qbuf2, qbuf3,
        comm_keep, comm_dreply, comm_creply,          // exchange driver below the C ABI (rb_comm.hip)
        cwriters, cshared;                            // per shared counter: runs that can reach it; the runs with a shared counter (k_cs_writers / k_cs_order)
    // profiling: HIP events recorded on the stream a stage runs on; resolved lazily (no host sync
    // inside the pipeline, so the two streams keep overlapping while timing is on)
    bool prof_on = false;
    struct ProfEntry { const char *name; double ms; int64_t launches; };
    std::vector<ProfEntry> prof;
    struct ProfPending { const char *name; hipEvent_t e0, e1; };
    std::vector<ProfPending> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    hipEvent_t prof_open[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    // rb_graph_add_packed: the batch being inserted is still arriving from host memory, piece by piece on a copy stream.  add_range calls this
    // before it enqueues anything that reads words [0, w_end) of the batch on `st`; the hook makes `st` wait for the pieces that hold them.
    std::function<void(int64_t w_end, hipStream_t st)> await_words;
    // ... and what that call keeps between calls (rb_packed.hip): the device batch the host's reads are streamed into, the pinned copy of its word
    // offsets, the copy stream and one event per uploaded piece
    struct PackedIngest {
        DevBuf codes, valid, word_read, woff, len, wc, temp, stats;
        uint32_t *h_woff = nullptr, *h_stats = nullptr;
        size_t h_woff_cap = 0;
        std::vector<hipEvent_t> ev;              // one behind every uploaded piece
        hipEvent_t ev_woff = nullptr;            // the word offsets are back in h_woff
        std::vector<int64_t> wend;               // piece p holds words [wend[p - 1], wend[p])
        // the upload in flight (rb_graph_prefetch_packed, or the add call itself): whose arrays, how many
        const void *src = nullptr;
        int64_t n_reads = 0, n_words = 0;
        bool inflight = false;
        std::vector<void *> pins;                // caller's arrays registered for the upload (hipHostUnregister when it is over)
        // streamed ASCII ingest (rb_graph_add_reads over more than one piece): the caller's base offsets on the device, two staging buffers
        // for the pieces' bases and qualities taking turns, the read each piece ends at
        DevBuf off, stage_seq[2], stage_qual[2];
        std::vector<int64_t> rend;
        // ... and its feeder: a helper thread registers the caller's arrays slab by slab (hipHostRegister: 8 ms per GB — 120 ms for config 2's 15 GB per
        // file if done up front, with the link idle meanwhile) and enqueues a piece's copies + encode as soon as its bytes are registered
        std::thread feeder;
        std::mutex fm;
        std::condition_variable fcv;
        size_t enqueued = 0;                     // pieces whose event has been recorded (a stream may only wait for a RECORDED event)
        bool cancel = false;
        int feeder_rc = 0;
        std::string feeder_err;
    } pk[2];
    IngestHost ingest_host[2];                   // pinned host scratch of the chunked text ingests' preparation, taking turns
    DevPool ingest_pool;                         // device blocks of the chunked text ingests (rb_graph_add_reads): handed from chunk to chunk and call to call, freed with the graph
    hipStream_t pk_stream = nullptr;             // the copy stream both slots upload on (in order: a prefetch queues behind the batch before it)
    std::mutex pk_mutex;                         // slots are handed out under it (a prefetch may come from another thread than the insert)
    int pk_busy = -1;                            // the slot the running rb_graph_add_packed reads
    hipEvent_t prof_event() {
        if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
        hipEvent_t e; RB_HIP(hipEventCreate(&e)); return e;
    }

    // need_all: the call works on dbgbf AND cbf (a filter freed by rb_graph_destroy_filter makes it fail loudly)
    FilterView view(uint64_t ordinal0, uint32_t pos_bits, bool need_all = true) const {
        if (need_all) RB_REQUIRE(dbg.bits && cbf, "this call needs dbgbf and cbf, and one of them has been destroyed");
        FilterView fv;
        fv.dbg = dbg.bits; fv.dbg_mod = dbg.mod; fv.dbg_h = dbg.num_hash;
        fv.cbf = cbf; fv.cbf_mod = cbf_mod; fv.cbf_h = cbf_h;
        fv.kmul = kmul_of(k); fv.seed = p.rng_seed; fv.ordinal0 = ordinal0; fv.pos_bits = pos_bits;
        fv.npf.tab = npf_log2 ? reinterpret_cast<unsigned long long *>(npf.p) : nullptr;
        fv.npf.log2n = npf_log2;
        fv.mpf.tab = (use_mpf && mpf_log2b) ? reinterpret_cast<unsigned long long *>(mpf.p) : nullptr;
        fv.mpf.log2b = mpf_log2b; fv.mpf.m = mpf_m;
        fv.seq_codes = seq_codes; fv.seq_woff = seq_woff; fv.seq_wpr = seq_wpr; fv.seq_first = seq_first; fv.k = k;
        return fv;
    }
    void prof_begin(hipStream_t st = nullptr) {
        if (!prof_on) return;
        const int w = (st && st == stream2) ? 1 : (st && st == stream3) ? 2 : 0;
        prof_open[w] = prof_event();
        RB_HIP(hipEventRecord(prof_open[w], w == 1 ? stream2 : w == 2 ? stream3 : stream));
    }
    void prof_end(const char *name, hipStream_t st = nullptr) {
        if (!prof_on) return;
        const int w = (st && st == stream2) ? 1 : (st && st == stream3) ? 2 : 0;
        if (!prof_open[w]) return;
        hipEvent_t e1 = prof_event();
        RB_HIP(hipEventRecord(e1, w == 1 ? stream2 : w == 2 ? stream3 : stream));
        prof_pending.push_back({name, prof_open[w], e1});
        prof_open[w] = nullptr;
    }
    void prof_collect() {   // call with both streams idle
        for (auto &pp : prof_pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pp.e0, pp.e1) == hipSuccess) {
                bool found = false;
                for (auto &e : prof) if (!strcmp(e.name, pp.name)) { e.ms += ms; e.launches++; found = true; break; }
                if (!found) prof.push_back({pp.name, ms, 1});
            }
            prof_pool.push_back(pp.e0); prof_pool.push_back(pp.e1);
        }
        prof_pending.clear();
    }
};


namespace rb {
// RAII: shared ownership of the handle + a query context (created on demand, at most kMaxQueryCtx per handle)
struct QueryLease {
    rb_graph *g;
    rb_query_ctx *c = nullptr;
    std::shared_lock<std::shared_mutex> lk;
    explicit QueryLease(rb_graph *g_) : g(g_), lk(g_->rw) {
        RB_HIP(hipSetDevice(g->p.device));
        std::unique_lock<std::mutex> q(g->qm);
        for (;;) {
            if (!g->qfree.empty()) { c = g->qfree.back(); g->qfree.pop_back(); return; }
            if (g->qmade < rb_graph::kMaxQueryCtx) {
                ++g->qmade;
                q.unlock();
                c = new rb_query_ctx();
                hipError_t e = hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking);
                if (e != hipSuccess) { delete c; c = nullptr; q.lock(); --g->qmade; q.unlock(); RB_HIP(e); }
                return;
            }
            g->qcv.wait(q);
        }
    }
    ~QueryLease() {
        if (!c) return;
        { std::lock_guard<std::mutex> q(g->qm); g->qfree.push_back(c); }
        g->qcv.notify_one();
    }
    QueryLease(const QueryLease &) = delete;
    QueryLease &operator=(const QueryLease &) = delete;
};
using WriteLock = std::unique_lock<std::shared_mutex>;
// sort + strengths + run-length encode of the N records in g->keys0/vals0 (rb_graph.hip)
uint32_t group_records(rb_graph *g, size_t N, uint64_t ordinal0, uint32_t pos_bits, rb_add_stats *stats, uint32_t **ctr_out);
// asynchronous halves of group_records: enqueue on `st` into slot `slot`; finish reads the run count
void group_enqueue(rb_graph *g, int slot, size_t N, uint64_t ordinal0, uint32_t pos_bits, hipStream_t st, DevBuf &temp, DevBuf &ctrbuf,
                   int flags = 0 /* GR_FLAG_DEAD: the records may hold ones the emit pass cancelled */);
uint32_t group_finish(rb_graph *g, int slot, hipStream_t st, DevBuf &temp, DevBuf &ctrbuf, hipStream_t scan_stream);
// paired k-mer walker: inserts into g->rpk (out_idx == nullptr) or collects global bit indices
void shard_free(rb_graph *g);   // rb_shard.hip
void shard_clear_pairs_acc(rb_graph *g);   // rb_shard.hip
void *alloc_best_placed(size_t bytes, const char *what);   // rb_graph.hip: zeroed device memory, the best placed of a few allocations (counting filters)
uint64_t *shard_query_h0(rb_graph *g, size_t n);                                                 // rb_shard.hip: the query protocol with hashes already on the device
void shard_query_make_dev(rb_graph *g, int what, int which_bits, size_t n, int64_t *bit_counts, int64_t *ctr_counts);
const void *shard_query_combine_dev(rb_graph *g, int which_bits, const void *breply_dev, const void *creply_dev);
void trav_free(rb_graph *g);    // rb_query.hip: state of a traversal on a sharded graph
void cbf_counts_device(rb_graph *g, const uint64_t *d_h0, size_t n, float *d_out);   // rb_query.hip
void launch_pairs(rb_graph *g, const rb_batch *b, int64_t w0, int64_t nw, int mode_hash, const uint32_t *chunk_off,
                  uint64_t *out_idx, unsigned long long *n_pairs_dev, hipStream_t st = nullptr, const BitFilter *into = nullptr /* another bit array of the pair filter's geometry (the sharded engine's accumulation copy) */);
}  // namespace rb
