// rb_shard.hip — the sharded (multi-GPU) insert engine: one rb_graph per rank holding index range
// [lo,hi) of every filter.  The phases mirror the single-GPU pipeline of rb_graph.hip, cut where data
// has to move between ranks (see the protocol comment in include/rb_capi.h and DESIGN.md §6):
//
//   requester = owner of a k-mer = the rank that holds its FIRST COUNTER (idx_0 = (h0 >>> 1) % cbf_size in the rank's range):
//               holds all occurrences of its k-mers in order, decides found-flags / op counts / counter updates.  Probes
//               that fall into its own ranges — probe 0 of the counting filter always, probe 0 of dbgbf too when both
//               filters have the same size, any other probe with probability 1/G — never become requests: the run-based
//               kernels of this file test / set / claim / write them in place (k_shard_probe, k_local_*, k_resolve_shard);
//   owner     = owner of a filter index range: tests & sets bits, arbitrates first setters, hands
//               out counter claims, stores counter bytes;
//   component owner = rank that replays one connected component of the "runs that share a counter"
//               graph in global occurrence order (the rank holding the component's smallest run id).
//
// Exactness argument is unchanged: first-setter arbitration happens at the bit's owner over ALL
// probes of the sub-batch; claim marks and the conflict set live at the counter's owner; runs whose
// counters nobody else claimed commute; everything else is replayed in global occurrence order by
// exactly one rank per component, on a private table of that component's counters.
#include <string.h>

#include "rb_pipeline.hpp"

using namespace rb;

struct ShardState {
    int G = 1, log2G = 0;
    int64_t span[4] = {0, 0, 0, 0};            // per filter (RB_DBGBF..RB_FPKBF)
    DevBuf slot[RB_SLOT_COUNT];
    size_t slot_bytes[RB_SLOT_COUNT] = {0};
    // requester-side state carried from group -> resolve -> conflict_route
    uint32_t D = 0, n_conf = 0;
    int mode = M_ADD;
    uint64_t n_kept = 0;                        // records that survived the prefilter in the last group()
    uint64_t ordinal0 = 0;
    uint32_t pos_bits = 0;
    uint32_t sub_reads_bits = 0, gid_bits = 32;  // the sub-batch in flight: occurrence ids are below 2^(pos_bits + sub_reads_bits), component labels below 2^gid_bits
                                                // (the conflict replay sorts on those bits only)
    DevBuf dreq_pos, creq_pos;                 // [D*h] position of (run, probe) in the bucketed request order
    DevBuf creq_dup;                           // [D*h] for a duplicated counter: the earlier probe it copies
    DevBuf lctr;                               // 32 spread counters: local claims that met a claimed counter (k_shard_probe)
    DevBuf lmask, lcoll, lcv;                  // per run: local probes (dbg mask | cbf mask << 8), local Bloom probes that met another probe, local claim replies (one byte per probe)
    bool has_f = false, has_cs = false;        // serve left a collision table / a contested-counter table for resolve
    uint32_t f_log2 = 1, cs_log2 = 1, csf_log2 = 16;
    DevBuf cfinal, conf_list;
    DevBuf oinfo, ord_pos, conf2, deferred, heavy2;   // ordered-set classification (rb_shard_resolve -> rb_shard_order_finish)
    uint32_t n_cand = 0;                       // runs with a contested counter in the sub-batch in flight
    // routing scratch
    DevBuf stage0, stage1, stage2, stage3, rhist, roffs, bounds, rcnt;
    // owner-side scratch
    DevBuf own_f, own_cs;
    // conflict path scratch
    DevBuf esz, eoff, etab, eslot, elabel, cdesc, cpos, cnops, cnoff;
    DevBuf rk0, rk1, ok0, ok1, ov0, ov1, rtab, rslot, rbig;
    // sub-batch prepared ahead on the producer stream (rb_shard_hash_begin / _emit)
    struct Prep {
        int stage = 0;                          // 0 none, 1 filter pass enqueued, 2 emit + grouping enqueued
        const rb_batch *b = nullptr;
        int64_t first = 0, n = 0, w0 = 0, nw = 0;
        int64_t own_first = 0, own_n = 0;     // split reads: the slice of the sub-batch this rank hashes (w0 / nw are the slice's words)
        bool split = false;
        uint64_t ordinal0 = 0;
        uint32_t pos_bits = 0;
        void *wstate = nullptr;              // per-word rolling state of the prefilter pass (nullptr: none)
        unsigned flags = 0;
        int slot = 0;
        uint32_t N = 0;
        uint64_t owned = 0;
    } prep;
    // Read-pair filter, replicated accumulation (round 4).  rpkbf.add is a pure OR, so nothing about it needs the owner while a file is
    // inserted: every rank ORs the pairs of ITS reads into a private full-size copy (the single-GPU walker, no index list, no routing
    // pass, no exchange per sub-batch, no serve-side scatter) and at the end of the call the copies are merged range by range into the
    // owners' shards — one all-to-all of G pieces of size/G bits (rb_shard_pairs_flush_begin / _end).  288 GB of HBM make the copy free
    // (1 GB at config 2, 33 GB at configs[3]); RB_SHARD_PAIRS=route is the routed path of rounds 1-3.
    BitFilter rpk_acc;                          // full range [0, bits); bits == nullptr: routed path
    bool acc_dirty = false;                     // something was ORed into rpk_acc since the last flush
    bool pairs_direct = false;                  // one rank: its shard IS the filter — the walker ORs straight into g->rpk (round 5; the routed path cost 85 ms per pass there)
    bool replicate_cache = false;               // split-reads mode: cache updates are broadcast to every rank
    DevBuf cache_upd;                           // [D] exponent to broadcast per run (0 = none)
    uint32_t *pinned = nullptr;                 // [0] = kept records, [16 + 16 q] = owned windows (32 spread counters)
    // queries
    DevBuf q_h0, q_bpos, q_cpos, q_out;
    size_t q_n = 0;
    int q_what = 0;
};

namespace {

// index ranges of the filters this rank holds
struct LocalRanges { uint64_t dlo, dhi, clo, chi; };
inline int64_t roundup64(int64_t x) { return (x + 63) / 64 * 64; }

struct Geometry { int64_t span, lo, hi; };
Geometry geom(int64_t size, int rank, int count) {
    Geometry g;
    g.span = roundup64((size + count - 1) / count);
    g.lo = std::min<int64_t>(size, g.span * rank);
    g.hi = std::min<int64_t>(size, g.span * (rank + 1));
    return g;
}

LocalRanges local_ranges(const rb_graph *g) { return LocalRanges{(uint64_t)g->dbg.lo, (uint64_t)g->dbg.hi, (uint64_t)g->cbf_lo, (uint64_t)g->cbf_hi}; }
// the k-mers this rank owns: first counter index inside its range (no test at all on a single rank)
OwnRange own_range(const rb_graph *g) {
    if (g->shard_count <= 1) return OwnRange{Mod{1, 0, 0}, 0, 0};
    return OwnRange{g->cbf_mod, (uint64_t)g->cbf_lo, (uint64_t)g->cbf_hi};
}

void *slot_reserve(ShardState *S, int slot, size_t bytes) {
    S->slot[slot].reserve(std::max<size_t>(bytes, 16));
    S->slot_bytes[slot] = bytes;
    return S->slot[slot].p;
}

// ---------------------------------------------------------------- routing ----
// Stable multi-way partition of n items into G destination buckets in two passes over the items
// (count, scatter).  A tile is 2048 items: 4 wavefronts x 8 rows of 64 consecutive items; ranks inside
// a row come from ballots, row bases from a 32-row prefix per destination, tile bases from a global
// exclusive scan over the [destination][tile] histogram.  No sort, no permutation gather.
constexpr int RT_TPB = 256, RT_IPT = 8, RT_TILE = RT_TPB * RT_IPT;

template <class F, bool SCATTER>
__global__ void __launch_bounds__(RT_TPB) k_route(F f, size_t n, uint32_t G, uint32_t nb, uint32_t *__restrict__ hist_or_offs) {
    __shared__ uint32_t s_cnt[4 * RT_IPT][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t r = threadIdx.x; r < 4u * RT_IPT * 64u; r += RT_TPB) (&s_cnt[0][0])[r] = 0u;
    __syncthreads();
    int dst[RT_IPT];
    uint32_t rank[RT_IPT];
    const size_t base = (size_t)blockIdx.x * RT_TILE + (size_t)wave * (64u * RT_IPT);
#pragma unroll
    for (int q = 0; q < RT_IPT; ++q) {
        const size_t i = base + (size_t)q * 64u + lane;
        const int d = (i < n) ? f.dest(i) : -1;
        dst[q] = d; rank[q] = 0;
        unsigned long long pending = __ballot(d >= 0);
        while (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int dl = __shfl(d, leader, 64);
            const unsigned long long m = __ballot(d == dl);
            if (d == dl) rank[q] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if ((int)lane == leader) s_cnt[wave * RT_IPT + q][dl] = (uint32_t)__popcll(m);
            pending &= ~m;
        }
    }
    __syncthreads();
    if (threadIdx.x < G) {
        uint32_t run = SCATTER ? hist_or_offs[(size_t)threadIdx.x * nb + blockIdx.x] : 0u;
        for (int r = 0; r < 4 * RT_IPT; ++r) { const uint32_t c = s_cnt[r][threadIdx.x]; s_cnt[r][threadIdx.x] = run; run += c; }
        if (!SCATTER) hist_or_offs[(size_t)threadIdx.x * nb + blockIdx.x] = run;
    }
    if (!SCATTER) return;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RT_IPT; ++q) {
        const size_t i = base + (size_t)q * 64u + lane;
        if (i >= n) continue;
        if (dst[q] >= 0) f.emit(i, s_cnt[wave * RT_IPT + q][dst[q]] + rank[q]);
        else f.drop(i);
    }
}
__global__ void k_pick_bounds(const uint32_t *__restrict__ offs, uint32_t G, uint32_t nb, uint64_t *__restrict__ bounds) {
    uint32_t g = threadIdx.x;
    if (g <= G) bounds[g] = offs[(size_t)g * nb];
}
// counts[G] <- items per destination; `place(f, kept)` points the functor at its outputs once the
// totals are known.  Returns the number of items kept.
template <class F, class P>
size_t route(rb_graph *g, F f, size_t n, int64_t *counts, P place, int buckets = 0) {
    ShardState *S = g->shard;
    hipStream_t s = g->stream;
    const int B = buckets ? buckets : S->G;
    for (int r = 0; r < B; ++r) counts[r] = 0;
    if (n == 0) { place(f, (size_t)0); return 0; }
    RB_REQUIRE(n < (1ull << 32), "route: too many items (%zu)", n);
    const uint32_t nb = (uint32_t)((n + RT_TILE - 1) / RT_TILE);
    const size_t nh = (size_t)B * nb + 1;
    S->rhist.reserve(nh * 4); S->roffs.reserve(nh * 4); S->bounds.reserve(2 * (S->G + 2) * 8);
    RB_HIP(hipMemsetAsync(S->rhist.as<uint32_t>() + (nh - 1), 0, 4, s));
    hipLaunchKernelGGL((k_route<F, false>), dim3(nb), dim3(RT_TPB), 0, s, f, n, (uint32_t)B, nb, S->rhist.as<uint32_t>());
    g->temp.reserve(scan_temp_bytes(nh));
    exclusive_scan_u32(g->temp.p, g->temp.cap, S->rhist.as<uint32_t>(), S->roffs.as<uint32_t>(), nh, s);
    hipLaunchKernelGGL(k_pick_bounds, dim3(1), dim3(128), 0, s, S->roffs.as<uint32_t>(), (uint32_t)B, nb, S->bounds.as<uint64_t>());
    std::vector<uint64_t> b(B + 1);
    RB_HIP(hipMemcpyAsync(b.data(), S->bounds.p, (B + 1) * 8, hipMemcpyDeviceToHost, s));
    RB_HIP(hipStreamSynchronize(s));
    for (int r = 0; r < B; ++r) counts[r] = (int64_t)(b[r + 1] - b[r]);
    const size_t kept = (size_t)b[B];
    place(f, kept);
    hipLaunchKernelGGL((k_route<F, true>), dim3(nb), dim3(RT_TPB), 0, s, f, n, (uint32_t)B, nb, S->roffs.as<uint32_t>());
    return kept;
}
// The same partition where the order inside a destination bucket does not matter (requests, counter writes: the owner answers by position and
// arbitrates by probe id, a counter is written by one run) — round 6.  The stable route above pays for its order with a [destination][tile]
// histogram, its scan (three kernels) and a bounds kernel per call, four calls a sub-batch; here a count kernel adds up one number per destination
// (LDS, then one global atomic per destination and block), the host turns the B totals into bases, and the scatter kernel reserves a tile's
// share of a destination's range with one atomicAdd per (tile, destination): tiles land in whatever order they arrive, items keep their order
// inside a tile.  emit(i, pos) still tells the caller where item i went (pos_of).
struct RouteBases { uint32_t b[64]; };
template <class F>
__global__ void __launch_bounds__(RT_TPB) k_route_count(F f, size_t n, uint32_t G, uint32_t *__restrict__ cnt) {
    __shared__ uint32_t s_c[64];
    if (threadIdx.x < 64u) s_c[threadIdx.x] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RT_TILE;
#pragma unroll
    for (int q = 0; q < RT_IPT; ++q) {
        const size_t i = base + (size_t)q * RT_TPB + threadIdx.x;
        const int d = (i < n) ? f.dest(i) : -1;
        // one LDS atomic per distinct destination of the wavefront's 64 items
        unsigned long long pending = __ballot(d >= 0);
        while (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int dl = __shfl(d, leader, 64);
            const unsigned long long m = __ballot(d == dl);
            if ((int)(threadIdx.x & 63u) == leader) atomicAdd(&s_c[dl], (uint32_t)__popcll(m));
            pending &= ~m;
        }
    }
    __syncthreads();
    if (threadIdx.x < G && s_c[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], s_c[threadIdx.x]);
}
template <class F>
__global__ void __launch_bounds__(RT_TPB) k_route_scatter_any(F f, size_t n, uint32_t G, RouteBases bases, uint32_t *__restrict__ cursor) {
    __shared__ uint32_t s_cnt[4 * RT_IPT][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t r = threadIdx.x; r < 4u * RT_IPT * 64u; r += RT_TPB) (&s_cnt[0][0])[r] = 0u;
    __syncthreads();
    int dst[RT_IPT];
    uint32_t rank[RT_IPT];
    const size_t base = (size_t)blockIdx.x * RT_TILE + (size_t)wave * (64u * RT_IPT);
#pragma unroll
    for (int q = 0; q < RT_IPT; ++q) {
        const size_t i = base + (size_t)q * 64u + lane;
        const int d = (i < n) ? f.dest(i) : -1;
        dst[q] = d; rank[q] = 0;
        unsigned long long pending = __ballot(d >= 0);
        while (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int dl = __shfl(d, leader, 64);
            const unsigned long long m = __ballot(d == dl);
            if (d == dl) rank[q] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if ((int)lane == leader) s_cnt[wave * RT_IPT + q][dl] = (uint32_t)__popcll(m);
            pending &= ~m;
        }
    }
    __syncthreads();
    if (threadIdx.x < G) {
        uint32_t total = 0;
        for (int r = 0; r < 4 * RT_IPT; ++r) total += s_cnt[r][threadIdx.x];
        uint32_t run = total ? bases.b[threadIdx.x] + atomicAdd(&cursor[threadIdx.x], total) : 0u;      // this tile's share of the destination's range
        for (int r = 0; r < 4 * RT_IPT; ++r) { const uint32_t c = s_cnt[r][threadIdx.x]; s_cnt[r][threadIdx.x] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RT_IPT; ++q) {
        const size_t i = base + (size_t)q * 64u + lane;
        if (i >= n) continue;
        if (dst[q] >= 0) f.emit(i, s_cnt[wave * RT_IPT + q][dst[q]] + rank[q]);
        else f.drop(i);
    }
}
template <class F, class P>
size_t route_any_order(rb_graph *g, F f, size_t n, int64_t *counts, P place) {
    if (g->shard->G <= 2) return route(g, f, n, counts, place);       // (two destinations: measured 13 ms per pass slower than the stable route at two ranks, 6 ms faster at eight)
    ShardState *S = g->shard;
    hipStream_t s = g->stream;
    const int B = S->G;
    RB_REQUIRE(B <= 64, "route: %d destinations", B);
    for (int r = 0; r < B; ++r) counts[r] = 0;
    if (n == 0) { place(f, (size_t)0); return 0; }
    RB_REQUIRE(n < (1ull << 32), "route: too many items (%zu)", n);
    const uint32_t nb = (uint32_t)((n + RT_TILE - 1) / RT_TILE);
    S->rcnt.reserve(2 * 64 * 4);
    uint32_t *cnt = S->rcnt.as<uint32_t>(), *cursor = cnt + 64;
    RB_HIP(hipMemsetAsync(cnt, 0, 2 * 64 * 4, s));
    hipLaunchKernelGGL((k_route_count<F>), dim3(nb), dim3(RT_TPB), 0, s, f, n, (uint32_t)B, cnt);
    uint32_t h[64];
    RB_HIP(hipMemcpyAsync(h, cnt, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    RB_HIP(hipStreamSynchronize(s));
    RouteBases bases;
    size_t kept = 0;
    for (int r = 0; r < 64; ++r) { bases.b[r] = (uint32_t)kept; if (r < B) { counts[r] = h[r]; kept += h[r]; } }
    place(f, kept);
    hipLaunchKernelGGL((k_route_scatter_any<F>), dim3(nb), dim3(RT_TPB), 0, s, f, n, (uint32_t)B, bases, cursor);
    return kept;
}
// ordered compaction of the received records by the prefilter's verdicts (one bucket)
struct RouteKeep {
    const uint8_t *keep; const uint64_t *keys; const uint32_t *occ;
    uint64_t *out_keys; uint32_t *out_occ;
    __device__ int dest(size_t i) const { return keep[i] ? 0 : -1; }
    __device__ void emit(size_t i, uint32_t pos) const { out_keys[pos] = keys[i]; out_occ[pos] = occ[i]; }
    __device__ void drop(size_t) const {}
};
// the common case: items are global filter indices staged in an array, destination = index / span
// idx / span without a 64-bit division per item and pass (the partition kernels call dest() 8 times per thread, twice): one mulhi + a fix-up
struct SpanDiv {
    uint64_t span, inv;
    SpanDiv(uint64_t s) : span(s), inv(s > 1 ? (uint64_t)(((unsigned __int128)1 << 64) / s) : ~0ull) {}      // (implicit: the call sites pass the span)
    __device__ __forceinline__ int of(uint64_t idx) const {
        uint64_t q = __umul64hi(idx, inv);
        if (idx - q * span >= span) ++q;
        return (int)q;
    }
};
struct RouteIdx {
    const uint64_t *idx; const uint8_t *drop_flag; SpanDiv span;
    const uint64_t *pay64; const uint8_t *pay8;             // optional payload columns
    uint64_t *out_idx, *out64; uint8_t *out8; uint32_t *pos_of;
    __device__ int dest(size_t i) const { return (drop_flag && drop_flag[i]) ? -1 : span.of(idx[i]); }
    __device__ void emit(size_t i, uint32_t pos) const {
        out_idx[pos] = idx[i];
        if (pay64) out64[pos] = pay64[i];
        if (pay8) out8[pos] = pay8[i];
        if (pos_of) pos_of[i] = pos;
    }
    __device__ void drop(size_t i) const { if (pos_of) pos_of[i] = 0xFFFFFFFFu; }
};

// ------------------------------------------------------------- requester ----
// generic window-hash path (k > 31): verdict per record — this rank owns the k-mer, and the no-op
// prefilter does not know that the occurrence's draw cannot move a counter
__global__ void k_rec_keep(FilterView fv, int use_cache, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ occ, size_t n,
                           OwnRange own, uint8_t *__restrict__ keep, uint32_t *__restrict__ owned_spread) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool mine = false, k = false;
    if (i < n) {
        const uint64_t h0 = keys[i];
        mine = own_mine(own, h0);
        if (mine) {
            const uint32_t s = use_cache ? npf_lookup(fv.npf, h0) : 0u;
            k = !s || draw_strength(occ_rnd(fv, occ[i])) >= s;
        }
        keep[i] = k ? 1u : 0u;
    }
    const unsigned long long m = __ballot(mine);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&owned_spread[16u * (blockIdx.x & 31u)], (uint32_t)__popcll(m));
}
// Stage A of a run at its k-mer's owner.  A probe that falls into this rank's own range is handled HERE, on the run: the
// Bloom bit is tested against the pre-batch state (bits are set later, by k_local_dbg_set in the serve phase, after the
// remote requests have been tested too), the counter is claimed at once (claims commute).  Only the other probes become
// requests: (index, probe id) per Bloom bit, the index per DISTINCT counter.  lmask = local Bloom probes | local counter
// probes << 8; lcv = the claim replies of the local counter probes, one byte per probe (bit 7 = claimed before).
__global__ void k_shard_probe(FilterView fv, LocalRanges R, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ starts,
                              const uint32_t *__restrict__ vals, uint32_t D, int mode,
                              uint64_t *__restrict__ d_idx, uint64_t *__restrict__ d_probe, uint8_t *__restrict__ d_drop,
                              uint64_t *__restrict__ c_idx, uint8_t *__restrict__ c_drop, uint8_t *__restrict__ c_dup,
                              uint32_t *__restrict__ status, uint16_t *__restrict__ lmask, uint64_t *__restrict__ lcv,
                              uint32_t *__restrict__ foreign_spread) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nforeign = 0;
    if (d < D) {
        const uint64_t h0 = uniq[d];
        const unsigned long long v_first = vals[starts[d]];
        uint32_t premask = 0, dm = 0, cm = 0;
        if (mode != M_COUNT_ONLY)
            for (int j = 0; j < fv.dbg_h; ++j) {
                const size_t q = (size_t)d * fv.dbg_h + j;
                const uint64_t idx = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod);
                const bool local = idx >= R.dlo && idx < R.dhi;
                d_idx[q] = idx;
                d_probe[q] = (v_first << 4) | (unsigned long long)j;
                d_drop[q] = local ? 1u : 0u;
                if (local) { dm |= 1u << j; if (bit_test(fv.dbg, idx - R.dlo)) premask |= 1u << j; }
            }
        uint64_t cidx[RB_MAX_HASH], lc = 0;
        for (int j = 0; j < fv.cbf_h; ++j) {
            const size_t q = (size_t)d * fv.cbf_h + j;
            cidx[j] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
            int dup = -1;
            for (int p = 0; p < j; ++p) if (cidx[p] == cidx[j] && dup < 0) dup = p;
            const bool local = dup < 0 && cidx[j] >= R.clo && cidx[j] < R.chi;
            c_idx[q] = cidx[j];
            c_drop[q] = (dup >= 0 || local) ? 1u : 0u;
            c_dup[q] = (uint8_t)(dup >= 0 ? dup : j);
            if (local) {
                cm |= 1u << j;
                const uint32_t byte = cbf_claim(fv.cbf, cidx[j] - R.clo);
                lc |= (uint64_t)byte << (8 * j);
                nforeign += (byte & CLAIM) ? 1u : 0u;
            }
        }
        status[d] = premask;
        lmask[d] = (uint16_t)(dm | (cm << 8));
        lcv[d] = lc;
    }
    for (int o = 32; o > 0; o >>= 1) nforeign += __shfl_down(nforeign, o, 64);
    if ((threadIdx.x & 63u) == 0 && nforeign) atomicAdd(&foreign_spread[16u * (blockIdx.x & 31u)], nforeign);
}
// serve phase, local half: set the local Bloom bits that were clear before the sub-batch; a bit found set now was met by
// another probe of the sub-batch (local or remote) -> lcoll, counted
__global__ void k_local_dbg_set(FilterView fv, LocalRanges R, const uint64_t *__restrict__ uniq, uint32_t D, const uint32_t *__restrict__ status,
                                const uint16_t *__restrict__ lmask, uint8_t *__restrict__ lcoll, uint32_t *__restrict__ spread) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ncoll = 0;
    if (d < D) {
        uint32_t need = (lmask[d] & 0xFFu) & ~(status[d] & 0xFFu), coll = 0;
        const uint64_t h0 = uniq[d];
        while (need) {
            const int j = __ffs((int)need) - 1;
            need &= need - 1u;
            const uint64_t b = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod) - R.dlo;
            const uint32_t m = 1u << (uint32_t)(b & 31u);
            if (atomicOr(&fv.dbg[b >> 5], m) & m) { coll |= 1u << j; ++ncoll; }
        }
        lcoll[d] = (uint8_t)coll;
    }
    for (int o = 32; o > 0; o >>= 1) ncoll += __shfl_down(ncoll, o, 64);
    if ((threadIdx.x & 63u) == 0 && ncoll) atomicAdd(&spread[16u * (blockIdx.x & 31u)], ncoll);
}
// the local probes that met another probe enter the collision table (same table, same ids as the remote requests: k_own_collide_*)
__global__ void k_local_collide_insert(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ starts, const uint32_t *__restrict__ vals,
                                       uint32_t D, const uint8_t *__restrict__ lcoll, Slot *ftable, uint32_t f_log2) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    uint32_t coll = lcoll[d];
    if (!coll) return;
    const uint64_t h0 = uniq[d];
    const unsigned long long v_first = vals[starts[d]];
    while (coll) {
        const int j = __ffs((int)coll) - 1;
        coll &= coll - 1u;
        Slot *sl = table_insert(ftable, f_log2, index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod));
        atomicMin(&sl->val, (v_first << 4) | (unsigned long long)j);
    }
}
// ... and the local probes that set their bit first must be in it too, if the bit has an entry
__global__ void k_local_collide_fixup(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ starts, const uint32_t *__restrict__ vals,
                                      uint32_t D, const uint32_t *__restrict__ status, const uint16_t *__restrict__ lmask, const uint8_t *__restrict__ lcoll,
                                      Slot *ftable, uint32_t f_log2) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    uint32_t setter = (lmask[d] & 0xFFu) & ~(status[d] & 0xFFu) & ~(uint32_t)lcoll[d];
    if (!setter) return;
    const uint64_t h0 = uniq[d];
    const unsigned long long v_first = vals[starts[d]];
    while (setter) {
        const int j = __ffs((int)setter) - 1;
        setter &= setter - 1u;
        Slot *sl = const_cast<Slot *>(table_find(ftable, f_log2, index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.dbg_mod)));
        if (sl) atomicMin(&sl->val, (v_first << 4) | (unsigned long long)j);
    }
}
// local claims that found the counter claimed already: into the contested-counter table (next to the remote ones, k_own_cs_build)
__global__ void k_local_cs_build(FilterView fv, const uint64_t *__restrict__ uniq, uint32_t D, const uint16_t *__restrict__ lmask, const uint64_t *__restrict__ lcv,
                                 Slot *cs, uint32_t cs_log2, uint32_t *__restrict__ csf, uint32_t csf_log2) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const uint64_t lc = lcv[d];
    if (!(lc & 0x8080808080808080ull)) return;
    uint32_t cm = (uint32_t)lmask[d] >> 8;
    const uint64_t h0 = uniq[d];
    while (cm) {
        const int j = __ffs((int)cm) - 1;
        cm &= cm - 1u;
        if (!((lc >> (8 * j)) & 0x80ull)) continue;
        const uint64_t idx = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
        table_insert(cs, cs_log2, idx);
        const uint64_t b = slot_of(idx, csf_log2);
        atomicOr(&csf[b >> 5], 1u << (uint32_t)(b & 31u));
    }
}

// status bits 21..28: probes whose counter is claimed by another run of the sub-batch too
constexpr uint32_t ST_CONTESTED_SHIFT = 21;

// where a run's local probes find what the serve phase left: the collision table of the Bloom bits two probes met on and the
// contested-counter table with its bit filter (either may be absent: nothing collided / nothing was contested)
struct LocalTables { const Slot *ftab; uint32_t f_log2; const Slot *cs; uint32_t cs_log2; const uint32_t *csf; uint32_t csf_log2; };
__global__ void k_resolve_shard(FilterView fv, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ starts,
                                uint32_t D, int mode, uint32_t light_ops, const uint32_t *__restrict__ dreq_pos,
                                const uint8_t *__restrict__ dreply, const uint32_t *__restrict__ creq_pos,
                                const uint8_t *__restrict__ c_dup, const uint8_t *__restrict__ creply,
                                const uint8_t *__restrict__ tz, const uint64_t *__restrict__ uniq, uint32_t *__restrict__ status,
                                uint32_t *__restrict__ nops, uint64_t *__restrict__ cvals, uint64_t *__restrict__ cfinal,
                                uint8_t *__restrict__ cache_upd, const uint32_t *__restrict__ vals,
                                const uint16_t *__restrict__ lmask, const uint64_t *__restrict__ lcv, LocalTables T) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    if (cache_upd) cache_upd[d] = 0;
    const uint32_t m = counts[d];
    const uint32_t dm = lmask[d] & 0xFFu, cm = (uint32_t)lmask[d] >> 8, lpre = status[d] & 0xFFu;
    const uint64_t h0r = uniq[d], lc = lcv[d];
    const bool sets = mode == M_ADD || mode == M_ADD_IF_ABSENT;
    uint32_t premask = 0;
    bool all_pre = true, found_first = true;
    if (mode != M_COUNT_ONLY) {
        for (int j = 0; j < fv.dbg_h; ++j) {
            if ((dm >> j) & 1u) {                  // a local probe: pre-batch state from k_shard_probe, first setter from the collision table
                if ((lpre >> j) & 1u) { premask |= 1u << j; continue; }
                all_pre = false;
                if (sets) {
                    const Slot *sl = T.ftab ? table_find(T.ftab, T.f_log2, index_of(multi_hash(h0r, (uint32_t)j, fv.kmul), fv.dbg_mod)) : nullptr;
                    if (!sl || sl->val == (((unsigned long long)vals[starts[d]] << 4) | (unsigned long long)j)) found_first = false;
                }
                continue;
            }
            const uint8_t r = dreply[dreq_pos[(size_t)d * fv.dbg_h + j]];
            if (r & 1u) premask |= 1u << j;
            else { all_pre = false; if (r & 2u) found_first = false; }   // this probe was the first setter => bit was clear
        }
    }
    uint32_t ops = 0, kfirst = K_INC, krest = K_INC;
    if (mode == M_COUNT_ONLY) ops = m;
    else if (mode == M_COUNT_IF_PRESENT) { ops = all_pre ? m : 0u; kfirst = krest = K_INC_IF_POS; }
    else if (mode == M_ADD) ops = (all_pre || found_first) ? m : m - 1u;
    else { ops = m; kfirst = (all_pre || found_first) ? K_INC_IF_ZERO : K_INC; krest = K_INC_IF_ZERO; }
    uint32_t c[RB_MAX_HASH];
    uint32_t contested = 0;
    uint64_t cv = 0;
    for (int j = 0; j < fv.cbf_h; ++j) {
        const size_t q = (size_t)d * fv.cbf_h + j;
        const int src = c_dup[q];
        if (src != j) c[j] = c[src];
        else if ((cm >> j) & 1u) {             // a local counter: claimed by k_shard_probe; contested if anybody else's claim met it
            const uint32_t r = (uint32_t)(lc >> (8 * j)) & 0xFFu;
            c[j] = r & 0x7Fu;
            bool con = (r & 0x80u) != 0u;
            if (!con && T.cs) {
                const uint64_t idx = index_of(multi_hash(h0r, (uint32_t)j, fv.kmul), fv.cbf_mod);
                const uint64_t b = slot_of(idx, T.csf_log2);
                con = ((T.csf[b >> 5] >> (uint32_t)(b & 31u)) & 1u) && table_find(T.cs, T.cs_log2, idx);
            }
            if (con) contested |= 1u << j;
        } else {
            const uint8_t r = creply[creq_pos[q]];
            c[j] = r & 0x7Fu;
            if (r & 0x80u) contested |= 1u << j;
        }
        cv |= (uint64_t)c[j] << (8 * j);
    }
    cvals[d] = cv;
    uint32_t st = premask | (all_pre ? ST_ALLPRE : 0u) | (kfirst << 12) | (krest << 14) | (contested << ST_CONTESTED_SHIFT);
    nops[d] = ops;
    if (ops == 0) { status[d] = st | RUN_RELEASE; return; }
    if (contested) {
        status[d] = st | RUN_CONFLICT;
        // replayed elsewhere; its pre-batch minimum is still a valid lower bound for the prefilter cache
        // (the k-mer is in dbgbf: ops > 0 means present before or inserted by this sub-batch's serve)
        if (cache_on(fv) && mode != M_COUNT_ONLY) {
            uint32_t mn = c[0];
            for (int j = 1; j < fv.cbf_h; ++j) mn = c[j] < mn ? c[j] : mn;
            if (mn >= 16u) { if (cache_store(fv, uniq[d], vals[starts[d]], cache_exp(mn)) && cache_upd) cache_upd[d] = (uint8_t)cache_exp(mn); }
        }
        return;
    }
    if (ops > light_ops) { status[d] = st | RUN_WRITES | RUN_HEAVY; return; }
    status[d] = st | RUN_WRITES;
    uint32_t mn0 = c[0];
    for (int j = 1; j < fv.cbf_h; ++j) mn0 = c[j] < mn0 ? c[j] : mn0;
    run_ops(c, fv.cbf_h, kfirst, krest, tz, starts[d] + m - ops, ops);
    uint64_t out = 0;
    for (int j = 0; j < fv.cbf_h; ++j) out |= (uint64_t)c[j] << (8 * j);
    cfinal[d] = out;
    if (cache_on(fv) && mode != M_COUNT_ONLY) {   // the k-mer is in dbgbf now; remember its counter exponent
        uint32_t mn = c[0];
        for (int j = 1; j < fv.cbf_h; ++j) mn = c[j] < mn ? c[j] : mn;
        // (not when the cache evidently has it: same exponent and every op succeeded — see k_resolve_apply)
        const bool cached = mn0 >= 16u && (mn >> 3) == (mn0 >> 3) && mn - mn0 == ops && !(mn >= 127u && mn0 < 127u);   // (reaching 127 is news: the k-mer is saturated)
        if (mn >= 16u && !cached) { if (cache_store(fv, uniq[d], vals[starts[d]], cache_exp(mn)) && cache_upd) cache_upd[d] = (uint8_t)cache_exp(mn); }
    }
}
// cache updates of this sub-batch, compacted for the broadcast
struct CacheUpd { uint64_t h0, e; };      // e = exponent | minimizer bucket << 8 (bucket: minimizer-bucketed replicas only)
struct RouteUpd {
    const uint8_t *upd; const uint64_t *uniq; const uint32_t *starts, *vals; FilterView fv; CacheUpd *out;
    __device__ int dest(size_t i) const { return upd[i] ? 0 : -1; }
    __device__ void emit(size_t i, uint32_t pos) const {
        CacheUpd u; u.h0 = uniq[i]; u.e = upd[i];
        if (fv.mpf.tab && fv.seq_codes) {          // the receivers do not know which of its reads the k-mer came from: ship the bucket
            const uint32_t occ = vals[starts[i]];
            const uint32_t r = fv.seq_first + (occ >> fv.pos_bits), p = occ & ((1u << fv.pos_bits) - 1u);
            u.e |= mpf_bucket(fv.mpf, window_min_order(fv.seq_codes + seq_word0(fv, r), p, (uint32_t)fv.k, fv.mpf.m)) << 8;
        }
        out[pos] = u;
    }
    __device__ void drop(size_t) const {}
};
__global__ void k_cache_apply(FilterView fv, const CacheUpd *__restrict__ u, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fv.mpf.tab) mpf_store(fv.mpf, u[i].e >> 8, u[i].h0, (uint32_t)(u[i].e & 0xFFull));
    else if (fv.npf.tab) npf_store(fv.npf, u[i].h0, (uint32_t)(u[i].e & 0xFFull));
}
// records of this rank's read slice, bucketed by the rank that owns their k-mer (stable)
struct RouteRec {
    const uint64_t *keys; const uint32_t *occ; const uint8_t *keep; Mod cmod; OwnSpan cspan;
    uint64_t *out_keys; uint32_t *out_occ;
    __device__ int dest(size_t i) const { return (keep && !keep[i]) ? -1 : (int)own_rank_of(cspan, index_of(keys[i], cmod)); }
    __device__ void emit(size_t i, uint32_t pos) const { out_keys[pos] = keys[i]; out_occ[pos] = occ[i]; }
    __device__ void drop(size_t) const {}
};
__global__ void k_emit_writes(FilterView fv, LocalRanges R, const uint64_t *__restrict__ uniq, uint32_t D, const uint32_t *__restrict__ status,
                              const uint8_t *__restrict__ c_dup, const uint64_t *__restrict__ cfinal, const uint16_t *__restrict__ lmask,
                              uint64_t *__restrict__ w_idx, uint8_t *__restrict__ w_val, uint8_t *__restrict__ w_drop) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const uint32_t st = status[d], cm = (uint32_t)lmask[d] >> 8;
    const uint64_t h0 = uniq[d], cf = cfinal[d];
    for (int j = 0; j < fv.cbf_h; ++j) {
        const size_t q = (size_t)d * fv.cbf_h + j;
        bool send = (c_dup[q] == j) && (st & (RUN_RELEASE | RUN_WRITES));
        const uint64_t idx = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
        const uint8_t val = (st & RUN_RELEASE) ? (uint8_t)0xFF : (uint8_t)(cf >> (8 * j));
        if (send && ((cm >> j) & 1u)) {        // this rank's own counter: written (or released) in place — the store drops the claim mark
            if (val == 0xFFu) cbf_release(fv.cbf, idx - R.clo); else fv.cbf[idx - R.clo] = val;
            send = false;
        }
        w_idx[q] = idx;
        w_val[q] = val;
        w_drop[q] = !send;
    }
}

// ---- conflict path, requester side: edges of the (run, contested counter) graph ----
struct ConfEdge { uint64_t cidx; uint32_t gid, pad; };          // gid = local conflict-run number * G + rank
// ... and, in the same 16 bytes and the same all-gather, the counter WRITES of the runs the ordered-set rule finished in place (round 6): gid =
// EDGE_WRITE, pad = the byte (0xFF: drop the claim mark only).  Every rank applies the ones that fall into its range (k_apply_tagged); the
// component kernels skip them.
constexpr uint32_t EDGE_WRITE = 0xFFFFFFFFu;
struct ConfRun { uint64_t h0, cv; uint32_t label, nops_kinds; }; // nops | kfirst << 28 | krest << 30
constexpr uint32_t NOPS_MASK = 0x0FFFFFFFu;

__global__ void k_conf_sizes(const uint32_t *__restrict__ conf_list, const uint32_t *__restrict__ status, uint32_t n,
                             uint32_t *__restrict__ esz) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) esz[i] = (uint32_t)__popc((status[conf_list[i]] >> ST_CONTESTED_SHIFT) & 0xFFu);
    if (i == n) esz[i] = 0;
}
__global__ void k_conf_edges(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ conf_list,
                             const uint32_t *__restrict__ status, const uint32_t *__restrict__ eoff, uint32_t n, uint32_t G,
                             uint32_t rank, ConfEdge *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t d = conf_list[i];
    const uint64_t h0 = uniq[d];
    uint32_t mask = (status[d] >> ST_CONTESTED_SHIFT) & 0xFFu, o = eoff[i];
    while (mask) {
        const int j = __ffs((int)mask) - 1;
        mask &= mask - 1u;
        ConfEdge e;
        e.cidx = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
        e.gid = i * G + rank; e.pad = 0;
        out[o++] = e;
    }
}
// ---- which runs with a contested counter need the ordered replay at all (round 6; the single-GPU engine's k_cs_writers / k_cs_order, rb_graph.hip) ----
// A run X only ever raises its counters; as long as nobody else writes the counters it works on, its minimum after its m ops is at most
// m0 + m, so it writes a counter — and depends on that counter's exact value — only if the pre-batch value is within reach: c <= m0 + m - 1.
// The single-GPU engine closes the set O = {can reach a contested counter two runs can reach, or one a run of O claimed} by rounds over one
// table.  Here the claimants of a counter sit on different ranks and a round would be an exchange, so a SUPERSET that needs no closure is used:
//     any(X)  =  X can reach at least one of its contested counters                                   (known where X lives)
//     X in O* =  X can reach a contested counter c that ANOTHER claimant Y with any(Y) claimed        (one count per counter, at c's owner)
// O is inside O*: "two runs can reach c" gives the other one any(); "a run of O claimed c" too, for every run of O can reach something.  A run
// outside O* is the only writer of everything it can reach (the other claimants of such a counter reach nothing contested, their own bound
// holds, they never get to it), so it is finished in place from the pre-batch values, and it never writes a counter an O* run shares with it
// (it would be in O* through that run).  Hot k-mers whose co-claimants are cold ones stay outside — where most replayed ops were.
// Protocol: the runs with any() send their contested counters to the owners (with the counter writes of the resolve phase), the owners count
// claimants per counter and answer the count, k_order_decide classifies.
__global__ void k_order_requests(FilterView fv, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ conf_list, uint32_t n,
                                 const uint32_t *__restrict__ status, const uint32_t *__restrict__ nops, const uint64_t *__restrict__ cvals,
                                 const uint8_t *__restrict__ c_dup, uint8_t *__restrict__ oinfo, uint64_t *__restrict__ o_idx, uint8_t *__restrict__ o_drop) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t d = conf_list[i], con = (status[d] >> ST_CONTESTED_SHIFT) & 0xFFu;
    const uint64_t cv = cvals[d], h0 = uniq[d];
    uint32_t m0 = 255u;
    for (int j = 0; j < fv.cbf_h; ++j) { const uint32_t c = (uint32_t)(cv >> (8 * j)) & 0xFFu; m0 = c < m0 ? c : m0; }
    const uint32_t reach = m0 + nops[d] - 1u;
    uint32_t rmask = 0;
    for (int j = 0; j < fv.cbf_h; ++j)
        if (((con >> j) & 1u) && ((uint32_t)(cv >> (8 * j)) & 0xFFu) <= reach) rmask |= 1u << j;
    oinfo[i] = (uint8_t)rmask;                                 // the contested counters this run can reach (0: finished in place without asking)
    for (int j = 0; j < fv.cbf_h; ++j) {
        const size_t q = (size_t)i * fv.cbf_h + j;
        o_idx[q] = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
        o_drop[q] = (rmask && ((con >> j) & 1u) && c_dup[(size_t)d * fv.cbf_h + j] == j) ? 0u : 1u;      // every contested counter of a run with any()
    }
}
// owner: claimants with any() per contested counter (the slot's value starts at all ones: n adds leave n - 1)
__global__ void k_order_count(const uint64_t *__restrict__ idx, size_t n, Slot *cs, uint32_t cs_log2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Slot *sl = const_cast<Slot *>(table_find(cs, cs_log2, idx[i]));
    if (sl) atomicAdd(&sl->val, 1ull);
}
__global__ void k_order_reply(const uint64_t *__restrict__ idx, size_t n, const Slot *cs, uint32_t cs_log2, uint8_t *__restrict__ reply) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Slot *sl = cs ? table_find(cs, cs_log2, idx[i]) : nullptr;
    const unsigned long long c = sl ? sl->val + 1ull : 255ull;                // (a counter the table does not know cannot happen; answered as "many")
    reply[i] = (uint8_t)(c > 255ull ? 255ull : c);
}
// requester: in O* (stays RUN_CONFLICT, appended to conf2) or finished in place — light runs here, heavy ones by a second k_cbf_heavy launch.
// A contested counter the run did not change is only released (byte 0xFF): it belongs to whoever can reach it.
__global__ void k_order_decide(FilterView fv, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ starts, const uint32_t *__restrict__ conf_list,
                               uint32_t n, const uint8_t *__restrict__ oinfo, const uint32_t *__restrict__ ord_pos, const uint8_t *__restrict__ reply,
                               uint32_t light_ops, int order_all, uint32_t *__restrict__ status, const uint32_t *__restrict__ nops,
                               const uint64_t *__restrict__ cvals, const uint8_t *__restrict__ tz, uint64_t *__restrict__ cfinal,
                               uint32_t *__restrict__ conf2, uint32_t *__restrict__ deferred, uint32_t *__restrict__ heavy2, uint32_t *__restrict__ ctr3 /* conf2, deferred, heavy2 counts */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t d = conf_list[i], rmask = oinfo[i];
    bool in_o = order_all != 0;
    for (int j = 0; j < fv.cbf_h && !in_o; ++j)
        if ((rmask >> j) & 1u) in_o = reply[ord_pos[(size_t)i * fv.cbf_h + j]] >= 2u;         // this run and at least one other with any()
    if (in_o) { conf2[atomicAdd(&ctr3[0], 1u)] = d; return; }
    const uint32_t st = status[d] & ~RUN_CONFLICT, ops = nops[d], con = (st >> ST_CONTESTED_SHIFT) & 0xFFu;
    deferred[atomicAdd(&ctr3[1], 1u)] = d;
    if (ops > light_ops) { status[d] = st | RUN_WRITES | RUN_HEAVY; heavy2[atomicAdd(&ctr3[2], 1u)] = d; return; }
    status[d] = st | RUN_WRITES;
    const uint64_t cv = cvals[d];
    uint32_t c[RB_MAX_HASH];
    for (int j = 0; j < fv.cbf_h; ++j) c[j] = (uint32_t)(cv >> (8 * j)) & 0xFFu;
    run_ops(c, fv.cbf_h, (st >> 12) & 3u, (st >> 14) & 3u, tz, starts[d] + counts[d] - ops, ops);
    uint64_t out = 0;
    for (int j = 0; j < fv.cbf_h; ++j) {
        const uint32_t c0 = (uint32_t)(cv >> (8 * j)) & 0xFFu;
        out |= (uint64_t)((((con >> j) & 1u) && c[j] == c0) ? 0xFFu : c[j]) << (8 * j);
    }
    cfinal[d] = out;
}
// the heavy ones among them, after k_cbf_heavy left their final bytes: unchanged contested counters become releases
__global__ void k_order_release_fix(int h, const uint32_t *__restrict__ heavy2, const uint32_t *__restrict__ n_heavy2, const uint32_t *__restrict__ status,
                                    const uint64_t *__restrict__ cvals, uint64_t *__restrict__ cfinal) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_heavy2) return;
    const uint32_t d = heavy2[i], con = (status[d] >> ST_CONTESTED_SHIFT) & 0xFFu;
    const uint64_t cv = cvals[d];
    uint64_t cf = cfinal[d];
    for (int j = 0; j < h; ++j)
        if (((con >> j) & 1u) && ((cf >> (8 * j)) & 0xFFull) == ((cv >> (8 * j)) & 0xFFull)) cf |= 0xFFull << (8 * j);
    cfinal[d] = cf;
}
// their counter writes: this rank's own counters in place, the others as tagged records behind the edges (ctr[0] = records so far)
__global__ void k_order_writes(FilterView fv, LocalRanges R, const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ deferred, const uint32_t *__restrict__ n_def,
                               const uint8_t *__restrict__ c_dup, const uint64_t *__restrict__ cfinal, const uint16_t *__restrict__ lmask,
                               ConfEdge *__restrict__ out, uint32_t *__restrict__ n_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *n_def) return;
    const uint32_t d = deferred[i], cm = (uint32_t)lmask[d] >> 8;
    const uint64_t h0 = uniq[d], cf = cfinal[d];
    for (int j = 0; j < fv.cbf_h; ++j) {
        if (c_dup[(size_t)d * fv.cbf_h + j] != j) continue;
        const uint64_t idx = index_of(multi_hash(h0, (uint32_t)j, fv.kmul), fv.cbf_mod);
        const uint8_t val = (uint8_t)(cf >> (8 * j));
        if ((cm >> j) & 1u) { if (val == 0xFFu) cbf_release(fv.cbf, idx - R.clo); else fv.cbf[idx - R.clo] = val; continue; }
        ConfEdge e; e.cidx = idx; e.gid = EDGE_WRITE; e.pad = val;
        out[atomicAdd(n_out, 1u)] = e;
    }
}
__global__ void k_apply_tagged(uint8_t *cbf, uint64_t lo, uint64_t hi, const ConfEdge *__restrict__ e, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || e[i].gid != EDGE_WRITE) return;
    const uint64_t idx = e[i].cidx;
    if (idx < lo || idx >= hi) return;
    if ((e[i].pad & 0xFFu) == 0xFFu) cbf_release(cbf, idx - lo); else cbf[idx - lo] = (uint8_t)e[i].pad;
}

// ---- components of the global edge list by min-label propagation (every rank, identical result) ----
__global__ void k_edge_init(const ConfEdge *__restrict__ e, size_t n, Slot *tab, uint32_t log2cap, uint32_t *__restrict__ eslot,
                            uint32_t *__restrict__ label) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || e[i].gid == EDGE_WRITE) return;
    eslot[i] = (uint32_t)(table_insert(tab, log2cap, e[i].cidx) - tab);
    label[e[i].gid] = e[i].gid;
}
// One round per kernel, one thread per edge (run, contested counter), in no particular order (every update is a min, any order ends at the
// component's smallest run id — the same on every rank, which is all that matters): the edge takes the smaller of its run's label and its
// counter's, follows the label one step (label[l] is a run of the same component with a label at most l: pointer jumping), and hands the result
// to both ends.  A fixed point has equal values on both ends of every edge; `changed` says whether this round moved anything.
__global__ void k_edge_round(const ConfEdge *__restrict__ e, size_t n, Slot *tab, const uint32_t *__restrict__ eslot,
                             uint32_t *__restrict__ label, uint32_t *__restrict__ changed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t gid = e[i].gid;
    if (gid == EDGE_WRITE) return;
    uint32_t *cv = reinterpret_cast<uint32_t *>(&tab[eslot[i]].val);
    const uint32_t lr = __hip_atomic_load(&label[gid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t lc = __hip_atomic_load(cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t l = lc < lr ? lc : lr;
    const uint32_t up = __hip_atomic_load(&label[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    l = up < l ? up : l;
    bool moved = false;
    if (l < lr && atomicMin(&label[gid], l) > l) moved = true;
    if (l < lc && atomicMin(cv, l) > l) moved = true;
    if (moved) *changed = 1u;
}
__global__ void k_conf_desc(const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ conf_list, const uint32_t *__restrict__ status,
                            const uint32_t *__restrict__ nops, const uint64_t *__restrict__ cvals, const uint32_t *__restrict__ label,
                            uint32_t n, uint32_t G, uint32_t rank, ConfRun *__restrict__ desc) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t d = conf_list[i], st = status[d];
    ConfRun r;
    r.h0 = uniq[d]; r.cv = cvals[d];
    r.label = label[i * G + rank];
    r.nops_kinds = nops[d] | (((st >> 12) & 3u) << 28) | (((st >> 14) & 3u) << 30);
    desc[i] = r;
}
struct RouteRuns {   // destination = rank of the component's smallest run id
    const ConfRun *desc; uint32_t G;
    ConfRun *out; uint32_t *pos_of, *nops_routed;
    __device__ int dest(size_t i) const { return (int)(desc[i].label % G); }
    __device__ void emit(size_t i, uint32_t pos) const { out[pos] = desc[i]; pos_of[i] = pos; nops_routed[pos] = desc[i].nops_kinds & NOPS_MASK; }
    __device__ void drop(size_t) const {}
};
// one wavefront per conflicting run: its pending occurrence ids, in order, at the run's routed offset
__global__ void k_conf_ops_out(const uint32_t *__restrict__ conf_list, const uint32_t *__restrict__ counts,
                               const uint32_t *__restrict__ starts, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ nops,
                               const uint32_t *__restrict__ pos_of, const uint32_t *__restrict__ noff, uint32_t n,
                               uint32_t *__restrict__ ops_out) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (wave >= n) return;
    const uint32_t d = conf_list[wave], ops = nops[d];
    const uint32_t base = starts[d] + counts[d] - ops, out = noff[pos_of[wave]];
    for (uint32_t i = lane; i < ops; i += 64u) ops_out[out + i] = vals[base + i];
}
__global__ void k_pick_u32(const uint32_t *__restrict__ a, const uint64_t *__restrict__ at, uint32_t n, uint64_t *__restrict__ out) {
    uint32_t i = threadIdx.x;
    if (i < n) out[i] = a[at[i]];
}

// ------------------------------------------------------------------ owner ----
// Bloom-bit requests at the owner: same arbitration as the single-GPU engine (rb_graph.hip, k_set_bits): test against
// the pre-batch state, set with a returning atomicOr, and give a table entry only to the bits two requests of the
// sub-batch met on.  reply bit 0 = set before the sub-batch, bit 1 = this probe is the sequentially first setter.
__global__ void k_own_dbg_test(const uint32_t *bits, uint64_t lo, const uint64_t *__restrict__ idx, size_t n, uint8_t *__restrict__ reply) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reply[i] = bit_test(bits, idx[i] - lo) ? 1u : 0u;
}
__global__ void k_own_dbg_set(uint32_t *bits, uint64_t lo, const uint64_t *__restrict__ idx, size_t n, uint8_t *__restrict__ reply,
                              uint32_t *__restrict__ counters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (reply[i] & 1u)) return;
    const uint64_t b = idx[i] - lo;
    const uint32_t m = 1u << (uint32_t)(b & 31u);
    if (atomicOr(&bits[b >> 5], m) & m) {          // another request of this sub-batch got there first
        reply[i] = 4u;
        atomicAdd(&counters[16 * (blockIdx.x & 31u)], 1u);
    }
}
__global__ void k_own_collide_insert(const uint64_t *__restrict__ idx, const uint64_t *__restrict__ probe, size_t n,
                                     const uint8_t *__restrict__ reply, Slot *ftable, uint32_t f_log2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !(reply[i] & 4u)) return;
    Slot *s = table_insert(ftable, f_log2, idx[i]);
    atomicMin(&s->val, (unsigned long long)probe[i]);
}
__global__ void k_own_collide_fixup(const uint64_t *__restrict__ idx, const uint64_t *__restrict__ probe, size_t n,
                                    const uint8_t *__restrict__ reply, Slot *ftable, uint32_t f_log2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || reply[i] != 0u) return;          // clear before, and set by this request
    Slot *s = const_cast<Slot *>(table_find(ftable, f_log2, idx[i]));
    if (s) atomicMin(&s->val, (unsigned long long)probe[i]);
}
__global__ void k_own_dbg_first(const uint64_t *__restrict__ idx, const uint64_t *__restrict__ probe, size_t n,
                                const Slot *ftable /* null: no two requests met */, uint32_t f_log2, uint8_t *__restrict__ reply) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = reply[i];
    if (r & 1u) return;
    const Slot *s = ftable ? table_find(ftable, f_log2, idx[i]) : nullptr;
    reply[i] = (!s || s->val == (unsigned long long)probe[i]) ? 2u : 0u;
}
__global__ void k_own_claim(uint8_t *cbf, uint64_t lo, const uint64_t *__restrict__ idx, size_t n, uint8_t *__restrict__ reply,
                            uint32_t *__restrict__ spread /* 32 counters, 16 words apart */) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const uint32_t byte = live ? cbf_claim(cbf, idx[i] - lo) : 0u;
    const unsigned long long m = __ballot(live && (byte & CLAIM));    // one add per wavefront (adds to one address queue up at ~10 ns apiece)
    if (m && (threadIdx.x & 63u) == 0u) atomicAdd(&spread[16 * (blockIdx.x & 31u)], (uint32_t)__popcll(m));
    if (live) reply[i] = (uint8_t)byte;                               // bit 7 = claimed before by another run
}
// contested counters: a table of their indices and, in front of it, a cache-resident bit filter over the same keys (most
// claims are not contested and never reach the table; same arrangement as stage B of the single-GPU engine)
__global__ void k_own_cs_build(const uint64_t *__restrict__ idx, const uint8_t *__restrict__ reply, size_t n, Slot *cs, uint32_t cs_log2,
                               uint32_t *__restrict__ csf, uint32_t csf_log2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !(reply[i] & 0x80u)) return;
    table_insert(cs, cs_log2, idx[i]);
    const uint64_t b = slot_of(idx[i], csf_log2);
    atomicOr(&csf[b >> 5], 1u << (uint32_t)(b & 31u));
}
__global__ void k_own_claim_fin(const uint64_t *__restrict__ idx, size_t n, const Slot *cs, uint32_t cs_log2, const uint32_t *__restrict__ csf,
                                uint32_t csf_log2, uint8_t *__restrict__ reply) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (reply[i] & 0x80u)) return;
    const uint64_t b = slot_of(idx[i], csf_log2);
    if (((csf[b >> 5] >> (uint32_t)(b & 31u)) & 1u) && table_find(cs, cs_log2, idx[i])) reply[i] |= 0x80u;
}
__global__ void k_own_bits(uint32_t *bits, uint64_t lo, const uint64_t *__restrict__ idx, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bit_set(bits, idx[i] - lo);
}
__global__ void k_own_writes(uint8_t *cbf, uint64_t lo, const uint64_t *__restrict__ idx, const uint8_t *__restrict__ val, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (val[i] == 0xFFu) cbf_release(cbf, idx[i] - lo); else cbf[idx[i] - lo] = val[i];
}

// ------------------------------------------ component owner: ordered replay ----
__global__ void k_run_nops(const ConfRun *__restrict__ runs, uint32_t R, uint32_t *__restrict__ sizes, uint64_t *__restrict__ kk) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) { sizes[i] = runs[i].nops_kinds & NOPS_MASK; kk[i] = ((uint64_t)runs[i].label << 32) | i; }
    if (i == R) sizes[i] = 0;
}
__global__ void k_run_expand(const ConfRun *__restrict__ runs, const uint32_t *__restrict__ noff, const uint32_t *__restrict__ occ,
                             uint32_t R, uint64_t *__restrict__ op_key, uint32_t *__restrict__ op_val) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (wave >= R) return;
    const ConfRun r = runs[wave];
    const uint32_t ops = r.nops_kinds & NOPS_MASK, o = noff[wave];
    const uint64_t hi = (uint64_t)r.label << 32;
    for (uint32_t i = lane; i < ops; i += 64u) {
        op_key[o + i] = hi | occ[o + i];
        op_val[o + i] = wave | ((i == 0 ? (r.nops_kinds >> 28) & 3u : r.nops_kinds >> 30) << 30);
    }
}
// the component owner's private copy of every counter its runs touch: table slot <- pre-batch value
__global__ void k_run_slots(FilterView fv, const ConfRun *__restrict__ runs, uint32_t R, Slot *tab, uint32_t log2cap,
                            uint32_t *__restrict__ rslot) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const ConfRun r = runs[i];
    for (int j = 0; j < fv.cbf_h; ++j) {
        Slot *s = table_insert(tab, log2cap, index_of(multi_hash(r.h0, (uint32_t)j, fv.kmul), fv.cbf_mod));
        s->val = (r.cv >> (8 * j)) & 0xFFull;          // every claimer saw the same pre-batch byte
        rslot[(size_t)i * fv.cbf_h + j] = (uint32_t)(s - tab);
    }
}
__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t *a, uint32_t n, uint64_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ void replay_serial_tab(const FilterView &fv, Slot *tab, const uint32_t *__restrict__ rslot, const uint64_t *__restrict__ op_key,
                                  const uint32_t *__restrict__ op_val, uint32_t os, uint32_t oe) {
    for (uint32_t i = os; i < oe; ++i) {
        const uint32_t v = (uint32_t)op_key[i];
        const uint32_t run = op_val[i] & 0x3FFFFFFFu, kind = op_val[i] >> 30;
        uint32_t sl[RB_MAX_HASH], c[RB_MAX_HASH], c0[RB_MAX_HASH];
        for (int j = 0; j < fv.cbf_h; ++j) {
            sl[j] = rslot[(size_t)run * fv.cbf_h + j];
            c0[j] = c[j] = (uint32_t) * (volatile unsigned long long *)&tab[sl[j]].val;
        }
        uint32_t mn = c[0];
        for (int j = 1; j < fv.cbf_h; ++j) mn = c[j] < mn ? c[j] : mn;
        cbf_step(c, fv.cbf_h, kind, (mn >= 16u && mn < 127u) ? occ_rnd(fv, v) : 0u);
        for (int j = 0; j < fv.cbf_h; ++j)
            if (c[j] != c0[j]) *(volatile unsigned long long *)&tab[sl[j]].val = c[j];
    }
}
constexpr uint32_t MAX_COMPONENT_KMERS = 8;
// one thread per run (sorted by component label): component heads either replay a small component
// themselves or flag it for the wave-cooperative kernel
__global__ void k_replay_small(FilterView fv, Slot *tab, const uint32_t *__restrict__ rslot, const uint64_t *__restrict__ run_keys,
                               uint32_t R, const uint64_t *__restrict__ op_key, const uint32_t *__restrict__ op_val, uint32_t n_ops,
                               uint32_t *__restrict__ big_flag, uint32_t small_ops) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    big_flag[i] = 0;
    const uint32_t lab = (uint32_t)(run_keys[i] >> 32);
    if (i > 0 && (uint32_t)(run_keys[i - 1] >> 32) == lab) return;   // not a component head
    const uint32_t os = lower_bound_u64(op_key, n_ops, (uint64_t)lab << 32);
    const uint32_t oe = lower_bound_u64(op_key, n_ops, ((uint64_t)lab + 1ull) << 32);
    if (oe - os > small_ops) { big_flag[i] = 1u; return; }
    replay_serial_tab(fv, tab, rslot, op_key, op_val, os, oe);
}
// one wavefront per large component: the component's counters live in LDS; 64 ops are examined at
// a time against the current state and the chain hops from one state-changing op to the next
__global__ void __launch_bounds__(64) k_replay_big(FilterView fv, Slot *tab, const uint32_t *__restrict__ rslot,
                                    const ConfRun *__restrict__ runs, const uint64_t *__restrict__ run_keys, uint32_t R,
                                    const uint64_t *__restrict__ op_key, const uint32_t *__restrict__ op_val, uint32_t n_ops,
                                    const uint32_t *__restrict__ big_list, const uint32_t *__restrict__ n_big) {
    __shared__ uint32_t s_tslot[MAX_COMPONENT_KMERS * RB_MAX_HASH];  // unique table slots
    __shared__ uint32_t s_val[MAX_COMPONENT_KMERS * RB_MAX_HASH];    // their current bytes
    __shared__ uint32_t s_val0[MAX_COMPONENT_KMERS * RB_MAX_HASH];
    __shared__ uint32_t s_slot[MAX_COMPONENT_KMERS][RB_MAX_HASH];    // k-mer probe -> unique counter
    __shared__ uint64_t s_h0[MAX_COMPONENT_KMERS];
    __shared__ uint32_t s_run[MAX_COMPONENT_KMERS];                  // one run carrying that hash
    __shared__ uint32_t s_nu, s_nk;
    const uint32_t lane = threadIdx.x;
    const int H = fv.cbf_h;
    for (uint32_t bi = blockIdx.x; bi < *n_big; bi += gridDim.x) {
        const uint32_t i0 = big_list[bi];
        const uint32_t lab = (uint32_t)(run_keys[i0] >> 32);
        const uint32_t i1 = lower_bound_u64(run_keys, R, ((uint64_t)lab + 1ull) << 32);   // runs [i0,i1)
        const uint32_t os = lower_bound_u64(op_key, n_ops, (uint64_t)lab << 32);
        const uint32_t oe = lower_bound_u64(op_key, n_ops, ((uint64_t)lab + 1ull) << 32);
        // distinct hashes of the component (a hash split into many runs is still one k-mer)
        __syncthreads();
        if (lane == 0) s_nk = 0;
        __syncthreads();
        bool overflow = false;
        for (uint32_t rb0 = i0; rb0 < i1 && !overflow; rb0 += 64u) {
            const bool have = rb0 + lane < i1;
            const uint32_t myrun = have ? (uint32_t)run_keys[rb0 + lane] : 0u;
            const uint64_t h = have ? runs[myrun].h0 : 0ull;
            unsigned long long pending = __ballot(have);
            while (pending) {
                const int leader = __ffsll((long long)pending) - 1;
                const uint64_t hl = __shfl(h, leader, 64);
                const uint32_t rl = __shfl(myrun, leader, 64);
                pending &= ~__ballot(have && h == hl);
                uint32_t q = 0, nkc = s_nk;
                while (q < nkc && s_h0[q] != hl) ++q;
                if (q == nkc) {
                    if (nkc == MAX_COMPONENT_KMERS) { overflow = true; break; }
                    __syncthreads();
                    if (lane == 0) { s_h0[nkc] = hl; s_run[nkc] = rl; s_nk = nkc + 1u; }
                    __syncthreads();
                }
            }
        }
        if (overflow) {                          // rare: many DIFFERENT k-mers in one component -> plain ordered replay
            if (lane == 0) replay_serial_tab(fv, tab, rslot, op_key, op_val, os, oe);
            continue;
        }
        const uint32_t nk = s_nk;
        __syncthreads();
        if (lane == 0) {                         // build the component's counter table (tiny)
            uint32_t nu = 0;
            for (uint32_t q = 0; q < nk; ++q)
                for (int j = 0; j < H; ++j) {
                    const uint32_t ts = rslot[(size_t)s_run[q] * H + j];
                    uint32_t u = 0;
                    while (u < nu && s_tslot[u] != ts) ++u;
                    if (u == nu) { s_tslot[u] = ts; s_val0[u] = s_val[u] = (uint32_t)tab[ts].val; ++nu; }
                    s_slot[q][j] = u;
                }
            s_nu = nu;
        }
        __syncthreads();
        for (uint32_t base = os; base < oe; base += 64u) {
            const uint32_t i = base + lane;
            uint32_t q = 0, kind = 0, rnd = 0;
            const bool live = i < oe;
            if (live) {
                const uint32_t ov = op_val[i];
                const uint64_t h = runs[ov & 0x3FFFFFFFu].h0;
                kind = ov >> 30;
                while (q < nk && s_h0[q] != h) ++q;
                rnd = occ_rnd(fv, (uint32_t)op_key[i]);
            }
            uint32_t cursor = 0;                 // ops below cursor are settled
            while (cursor < 64u) {
                bool changes = false;
                if (live && lane >= cursor) {
                    uint32_t mn = s_val[s_slot[q][0]];
                    for (int j = 1; j < H; ++j) { uint32_t c = s_val[s_slot[q][j]]; mn = c < mn ? c : mn; }
                    const bool gate = !((kind == K_INC_IF_POS && mn == 0u) || (kind == K_INC_IF_ZERO && mn != 0u));
                    changes = gate && minifloat_inc(mn, rnd) != mn;
                }
                const unsigned long long win = __ballot(changes);
                if (!win) break;
                const uint32_t first = (uint32_t)__ffsll((long long)win) - 1u;
                if (lane == first) {             // apply: every probe equal to the minimum moves up
                    uint32_t mn = s_val[s_slot[q][0]];
                    for (int j = 1; j < H; ++j) { uint32_t c = s_val[s_slot[q][j]]; mn = c < mn ? c : mn; }
                    for (int j = 0; j < H; ++j) if (s_val[s_slot[q][j]] == mn) s_val[s_slot[q][j]] = mn + 1u;
                }
                __syncthreads();
                cursor = first + 1u;
            }
            __syncthreads();
        }
        if (lane < s_nu && s_val[lane] != s_val0[lane]) tab[s_tslot[lane]].val = s_val[lane];
        __syncthreads();
    }
}
// every counter of the table goes back to its owner (the store also clears the claim mark)
__global__ void k_tab_emit(const Slot *__restrict__ tab, size_t cap, uint64_t *__restrict__ idx, uint8_t *__restrict__ val,
                           uint8_t *__restrict__ drop) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const bool used = tab[i].key != ~0ull;
    idx[i] = used ? tab[i].key : 0ull;
    val[i] = (uint8_t)tab[i].val;
    drop[i] = !used;
}

// ---------------------------------------------------------------- queries ----
__global__ void k_query_idx(uint64_t kmul, Mod bmod, int bh, Mod cmod, int ch, const uint64_t *__restrict__ h0, size_t n,
                            uint64_t *__restrict__ bidx, uint64_t *__restrict__ cidx) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = h0[i];
    for (int j = 0; j < bh; ++j) bidx[i * bh + j] = index_of(multi_hash(h, (uint32_t)j, kmul), bmod);
    for (int j = 0; j < ch; ++j) cidx[i * ch + j] = index_of(multi_hash(h, (uint32_t)j, kmul), cmod);
}
__global__ void k_query_bits(const uint32_t *__restrict__ bits, uint64_t lo, const uint64_t *__restrict__ idx, size_t n, uint8_t *__restrict__ reply) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reply[i] = bit_test(bits, idx[i] - lo) ? 1u : 0u;
}
__global__ void k_query_ctrs(const uint8_t *__restrict__ cbf, uint64_t lo, const uint64_t *__restrict__ idx, size_t n, uint8_t *__restrict__ reply) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reply[i] = cbf[idx[i] - lo];
}
// what: 0 bit-filter lookup (all bits set), 1 counting-filter count (MiniFloat of the minimum byte),
//       2 graph count = lookup ? count + 1 : 0   (BloomFilterDeBruijnGraph.getCount :562-570)
__global__ void k_query_combine(int what, int bh, int ch, const uint32_t *__restrict__ bpos, const uint8_t *__restrict__ breply,
                                const uint32_t *__restrict__ cpos, const uint8_t *__restrict__ creply, size_t n,
                                uint8_t *__restrict__ out8, float *__restrict__ outf) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool present = true;
    for (int j = 0; j < bh; ++j) present = present && breply[bpos[i * bh + j]] != 0;
    uint32_t mn = 0xFFu;
    for (int j = 0; j < ch; ++j) { const uint32_t c = creply[cpos[i * ch + j]]; mn = c < mn ? c : mn; }
    if (what == 0) out8[i] = present ? 1u : 0u;
    else if (what == 1) outf[i] = minifloat_to_float(mn);
    else outf[i] = present ? minifloat_to_float(mn) + 1.0f : 0.0f;   // +1 for the insert that only set the dbgbf bits (:565)
}

}  // namespace

// ---- window hashing of a sub-batch on the producer stream (k <= 31): ownership + prefilter pass,
//      then masked emit + grouping into the other GroupSlot; no host wait except for the record count ----
namespace {
void prep_filter(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, uint64_t ordinal0, uint32_t pos_bits, unsigned flags, hipStream_t st,
                 bool split = false, int64_t own_first = 0, int64_t own_n = 0) {
    ShardState *S = g->shard;
    ShardState::Prep &P = S->prep;
    P = ShardState::Prep();
    P.b = b; P.first = first; P.n = n; P.ordinal0 = ordinal0; P.pos_bits = pos_bits; P.flags = flags;
    P.split = split; P.own_first = own_first; P.own_n = own_n;
    const int64_t r0 = split ? own_first : first, rn = split ? own_n : n;       // split reads: this rank's slice, every window (no ownership test)
    P.w0 = b->h_woff[(size_t)r0]; P.nw = (int64_t)b->h_woff[(size_t)(r0 + rn)] - P.w0;
    P.slot = 1 - g->cur;
    P.stage = 1;
    if (!S->pinned) RB_HIP(hipHostMalloc(reinterpret_cast<void **>(&S->pinned), 4096, hipHostMallocDefault));
    memset(S->pinned, 0, 4096);
    if (P.nw <= 0) return;
    const int mode_hash = g->stranded ? ((flags & RB_ADD_REVCOMP) ? 2 : 0) : 1;
    const size_t nw = (size_t)P.nw;
    g->chunk_cnt.reserve((nw + 1) * 4); g->chunk_off.reserve((nw + 1) * 4); g->chunk_mask.reserve((nw + 1) * 4);
    g->temp2.reserve(scan_temp_bytes(nw + 1));
    g->npf_tot.reserve(2048);
    RB_HIP(hipMemsetAsync(g->npf_tot.p, 0, 2048, st));
    RB_HIP(hipMemsetAsync(g->chunk_cnt.as<uint32_t>() + nw, 0, 4, st));
    FilterView fv = g->view(ordinal0, pos_bits);
    Npf cache = fv.npf;
    if (!g->npf_log2) cache.tab = nullptr;
    void *wstate = nullptr;                                   // rolling state per word for the resuming emit pass (prep_emit)
    if (filter_saves_state(b, P.nw, g->k)) { g->wstate.reserve(((size_t)P.nw + 1) * 16); wstate = g->wstate.p; }
    P.wstate = wstate;
    launch_filter_windows(b, P.w0, P.nw, g->k, mode_hash, (uint32_t)first, pos_bits, g->p.rng_seed, ordinal0, cache,
                          g->chunk_cnt.as<uint32_t>(), g->chunk_mask.as<uint32_t>(), g->npf_tot.as<uint32_t>(), st,
                          split ? OwnRange{Mod{1, 0, 0}, 0, 0} : own_range(g), fv.mpf, wstate);
    exclusive_scan_u32(g->temp2.p, g->temp2.cap, g->chunk_cnt.as<uint32_t>(), g->chunk_off.as<uint32_t>(), nw + 1, st);
    RB_HIP(hipMemcpyAsync(&S->pinned[0], g->chunk_off.as<uint32_t>() + nw, 4, hipMemcpyDeviceToHost, st));
    RB_HIP(hipMemcpyAsync(&S->pinned[16], g->npf_tot.p, 2048, hipMemcpyDeviceToHost, st));
}
// runs of the grouped sub-batch in slot g->cur -> Bloom-bit requests (index + probe id) and counter claims,
// bucketed by filter owner
void make_and_route_requests(rb_graph *g, uint32_t D, int mode, uint64_t ordinal0, uint32_t pos_bits, int64_t *dreq_counts, int64_t *creq_counts) {
    ShardState *S = g->shard;
    hipStream_t s = g->stream;
    FilterView fv = g->view(ordinal0, pos_bits);
    const size_t nd = mode == M_COUNT_ONLY ? 0 : (size_t)D * fv.dbg_h, nc = (size_t)D * fv.cbf_h;
    S->stage0.reserve(nd * 8 + 16); S->stage1.reserve(nd * 8 + 16); S->stage3.reserve(nc * 8 + 16); S->stage2.reserve(nc + nd + 32);
    S->creq_dup.reserve(nc + 16);
    S->dreq_pos.reserve(nd * 4 + 16); S->creq_pos.reserve(nc * 4 + 16);
    S->lmask.reserve((size_t)D * 2 + 16); S->lcv.reserve((size_t)D * 8 + 16); S->lcoll.reserve((size_t)D + 16);
    S->lctr.reserve(2048);
    g->status.reserve((size_t)D * 4);
    RB_HIP(hipMemsetAsync(S->lctr.p, 0, 2048, s));
    uint8_t *c_drop = S->stage2.as<uint8_t>(), *d_drop = c_drop + nc + 16;
    // local probes are tested / claimed on the spot, the others become requests (k_shard_probe)
    hipLaunchKernelGGL(k_shard_probe, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, local_ranges(g), g->uniq().as<uint64_t>(), g->starts().as<uint32_t>(),
                       g->vals1().as<uint32_t>(), D, mode, S->stage0.as<uint64_t>(), S->stage1.as<uint64_t>(), d_drop,
                       S->stage3.as<uint64_t>(), c_drop, S->creq_dup.as<uint8_t>(), g->status.as<uint32_t>(), S->lmask.as<uint16_t>(),
                       S->lcv.as<uint64_t>(), S->lctr.as<uint32_t>());
    if (S->G == 1) {      // one rank holds every range: every probe was local, nothing becomes a request — no routing passes over lists of dropped items
        dreq_counts[0] = creq_counts[0] = 0;
        (void)slot_reserve(S, RB_SLOT_DREQ_IDX, 0); (void)slot_reserve(S, RB_SLOT_DREQ_PROBE, 0); (void)slot_reserve(S, RB_SLOT_CREQ_IDX, 0);
        if (nd) RB_HIP(hipMemsetAsync(S->dreq_pos.p, 0xFF, nd * 4, s));          // "no request slot", as route() marks a dropped item
        if (nc) RB_HIP(hipMemsetAsync(S->creq_pos.p, 0xFF, nc * 4, s));
        return;
    }
    RouteIdx fd{S->stage0.as<uint64_t>(), d_drop, (uint64_t)S->span[RB_DBGBF], S->stage1.as<uint64_t>(), nullptr,
                nullptr, nullptr, nullptr, S->dreq_pos.as<uint32_t>()};
    route_any_order(g, fd, nd, dreq_counts, [&](RouteIdx &ff, size_t kept) {
        ff.out_idx = (uint64_t *)slot_reserve(S, RB_SLOT_DREQ_IDX, kept * 8);
        ff.out64 = (uint64_t *)slot_reserve(S, RB_SLOT_DREQ_PROBE, kept * 8);
    });
    RouteIdx fc{S->stage3.as<uint64_t>(), c_drop, (uint64_t)S->span[RB_CBF], nullptr, nullptr,
                nullptr, nullptr, nullptr, S->creq_pos.as<uint32_t>()};
    route_any_order(g, fc, nc, creq_counts, [&](RouteIdx &ff, size_t kept) { ff.out_idx = (uint64_t *)slot_reserve(S, RB_SLOT_CREQ_IDX, kept * 8); });
}
void prep_emit(rb_graph *g, hipStream_t st) {
    ShardState *S = g->shard;
    ShardState::Prep &P = S->prep;
    if (P.stage != 1) return;
    RB_HIP(hipStreamSynchronize(st));                      // the filter pass (usually long finished)
    P.N = S->pinned[0];
    P.owned = 0;
    for (int q = 0; q < 32; ++q) P.owned += S->pinned[16 + 16 * q];
    if (P.N) {
        const int mode_hash = g->stranded ? ((P.flags & RB_ADD_REVCOMP) ? 2 : 0) : 1;
        g->keys0.reserve((size_t)P.N * 8); g->vals0.reserve((size_t)P.N * 4);
        launch_hash_windows_masked(P.b, P.w0, P.nw, g->k, mode_hash, g->chunk_off.as<uint32_t>(), g->chunk_mask.as<uint32_t>(),
                                   (uint32_t)P.first, P.pos_bits, g->keys0.as<uint64_t>(), g->vals0.as<uint32_t>(), st, P.wstate);
    }
    group_enqueue(g, P.slot, P.N, P.ordinal0, P.pos_bits, st, g->temp2, g->devctr2);
    P.stage = 2;
}
}  // namespace

// ------------------------------------------------------------------------ C ABI ----
extern "C" {

int rb_graph_create_shard(const rb_graph_params *p, int shard_rank, int shard_count, rb_graph **out) {
    rb_graph *g = nullptr;
    int rc = guarded([&] {
        RB_REQUIRE(p && out, "rb_graph_create_shard: null argument");
        RB_REQUIRE(shard_count >= 1 && shard_count <= 64 && (shard_count & (shard_count - 1)) == 0,
                   "rb_graph_create_shard: shard_count must be a power of two in [1,64]");
        RB_REQUIRE(shard_rank >= 0 && shard_rank < shard_count, "rb_graph_create_shard: bad shard_rank");
        RB_REQUIRE(p->k >= 1 && p->k <= RB_MAX_K && p->dbgbf_bits > 0 && p->cbf_bytes > 0, "rb_graph_create_shard: bad parameters");
        RB_REQUIRE(p->dbgbf_num_hash >= 1 && p->dbgbf_num_hash <= RB_MAX_HASH && p->cbf_num_hash >= 1 && p->cbf_num_hash <= RB_MAX_HASH,
                   "rb_graph_create_shard: numHash out of range");
        int ndev = 0;
        RB_HIP(hipGetDeviceCount(&ndev));
        RB_REQUIRE(p->device >= 0 && p->device < ndev, "rb_graph_create_shard: device %d not present", p->device);
        RB_HIP(hipSetDevice(p->device));
        g = new rb_graph();
        g->p = *p; g->k = p->k; g->stranded = p->stranded != 0;
        g->H = std::max(p->dbgbf_num_hash, p->cbf_num_hash);
        g->max_batch_kmers = p->max_batch_kmers > 0 ? p->max_batch_kmers : ((int64_t)1 << 30);
        if (p->group_bits) g->sort_begin_bit = 64 - p->group_bits;
        g->shard_rank = shard_rank; g->shard_count = shard_count;
        ShardState *S = g->shard = new ShardState();
        S->G = shard_count; S->log2G = (int)log2_ceil((uint64_t)shard_count);
        RB_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
        RB_HIP(hipStreamCreateWithFlags(&g->stream2, hipStreamNonBlocking));
        RB_HIP(hipEventCreate(&g->ev0));
        RB_HIP(hipEventCreate(&g->ev1));
        Geometry gd = geom(p->dbgbf_bits, shard_rank, shard_count);
        S->span[RB_DBGBF] = gd.span;
        alloc_bits(g->dbg, p->dbgbf_bits, p->dbgbf_num_hash, gd.lo, gd.hi);
        Geometry gc = geom(p->cbf_bytes, shard_rank, shard_count);
        S->span[RB_CBF] = gc.span;
        g->cbf_size = p->cbf_bytes; g->cbf_lo = gc.lo; g->cbf_hi = gc.hi;
        g->cbf_alloc = (((size_t)(gc.hi - gc.lo) + 3) / 4 + 1) * 4;
        g->cbf_h = p->cbf_num_hash;
        g->cbf_mod = make_mod((uint64_t)p->cbf_bytes);
        g->cbf = static_cast<uint8_t *>(rb::alloc_best_placed(g->cbf_alloc, "cbf shard"));
        if (p->use_read_paired_kmers) {
            RB_REQUIRE(p->pkbf_bits > 0 && p->pkbf_num_hash >= 1 && p->pkbf_num_hash <= RB_MAX_HASH, "rb_graph_create_shard: pair filter parameters invalid");
            Geometry gp = geom(p->pkbf_bits, shard_rank, shard_count);
            S->span[RB_RPKBF] = gp.span;
            alloc_bits(g->rpk, p->pkbf_bits, p->pkbf_num_hash, gp.lo, gp.hi);
            const char *pm = getenv("RB_SHARD_PAIRS");
            if (shard_count == 1 && !(pm && !strcmp(pm, "route"))) { S->pairs_direct = true; alloc_pair_seen(g->rpk); }
            if (shard_count > 1 && !(pm && !strcmp(pm, "route"))) {
                size_t free_b = 0, total_b = 0;
                RB_HIP(hipMemGetInfo(&free_b, &total_b));
                // room for the copy with half of the device still free afterwards, else this rank routes its pairs as before (each rank decides
                // for itself: the owners take pair bits both ways — virtual ranks that share one GPU run out of room at configs[2]'s sizes)
                if ((size_t)(p->pkbf_bits / 8) + (total_b >> 1) < free_b) {
                    alloc_bits(S->rpk_acc, p->pkbf_bits, p->pkbf_num_hash, 0, p->pkbf_bits);
                    alloc_pair_seen(S->rpk_acc);         // the walker's seen-pair cache (rb_graph.hip), for this rank's copy
                }
            }
        }
        {   // no-op prefilter cache over this rank's k-mers (applied to the records it receives)
            const char *e = getenv("RB_NPF");
            uint32_t l2 = log2_ceil((uint64_t)std::max<int64_t>(p->cbf_bytes / 64, 1));   // full size: in split-reads mode every rank holds a replica of all ranks' entries
            l2 = std::max(16u, std::min(28u, l2));
            if (e) l2 = (uint32_t)atoi(e);
            if (l2 >= 8 && l2 <= 30) {
                g->npf.reserve(sizeof(uint64_t) << l2);
                RB_HIP(hipMemset(g->npf.p, 0, sizeof(uint64_t) << l2));
                g->npf_log2 = l2;
            }
        }
        {   // minimizer-bucketed variant for the k <= 31 window-hash kernels (see rb_graph_create)
            const char *e = getenv("RB_MPF");
            uint32_t lb = log2_ceil((uint64_t)std::max<int64_t>(p->cbf_bytes / 256, 1));
            lb = std::max(12u, std::min(25u, lb));
            if (e) lb = (uint32_t)atoi(e);
            if (lb >= 8 && lb <= 28 && p->k <= 31 && p->k >= 8 && !getenv("RB_NO_MPF")) {
                g->mpf.reserve((size_t)128 << lb);
                RB_HIP(hipMemset(g->mpf.p, 0, (size_t)128 << lb));
                g->mpf_log2b = lb;
                g->mpf_m = (uint32_t)std::min(16, p->k);
                g->use_mpf = shard_count == 1 && (uint32_t)g->k - g->mpf_m + 1u <= 16u;   // split-reads mode turns it on (rb_shard_set_cache_replication)
            }
        }
        RB_HIP(hipDeviceSynchronize());
        *out = g;
    });
    if (rc != RB_OK && g) rb_graph_destroy(g);
    return rc;
}

int rb_shard_span(rb_graph *g, int which, int64_t *span, int64_t *lo, int64_t *hi) {
    if (!g || !g->shard) { set_error("rb_shard_span: not a sharded graph"); return RB_ERR_INVALID; }
    int64_t l, h;
    if (which == RB_CBF) { l = g->cbf_lo; h = g->cbf_hi; }
    else if (which == RB_DBGBF) { l = g->dbg.lo; h = g->dbg.hi; }
    else if (which == RB_RPKBF) { l = g->rpk.lo; h = g->rpk.hi; }
    else { set_error("rb_shard_span: unsupported filter %d", which); return RB_ERR_INVALID; }
    if (span) *span = g->shard->span[which];
    if (lo) *lo = l;
    if (hi) *hi = h;
    return RB_OK;
}

int rb_shard_take(rb_graph *g, int slot, void *dst_dev, int64_t nbytes) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && slot >= 0 && slot < RB_SLOT_COUNT, "rb_shard_take: bad argument");
        RB_REQUIRE((size_t)nbytes == g->shard->slot_bytes[slot], "rb_shard_take: slot %d holds %zu bytes, asked for %lld", slot,
                   g->shard->slot_bytes[slot], (long long)nbytes);
        RB_HIP(hipSetDevice(g->p.device));
        if (nbytes) RB_HIP(hipMemcpyAsync(dst_dev, g->shard->slot[slot].p, (size_t)nbytes, hipMemcpyDeviceToDevice, g->stream));
        RB_HIP(hipStreamSynchronize(g->stream));
    });
}

int rb_shard_slot(rb_graph *g, int slot, void **dev_ptr, int64_t *nbytes) {
    if (!g || !g->shard || slot < 0 || slot >= RB_SLOT_COUNT || !dev_ptr || !nbytes) { set_error("rb_shard_slot: bad argument"); return RB_ERR_INVALID; }
    *dev_ptr = g->shard->slot_bytes[slot] ? g->shard->slot[slot].p : nullptr;
    *nbytes = (int64_t)g->shard->slot_bytes[slot];
    return RB_OK;
}

int rb_shard_hash_begin(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, uint64_t ordinal0, uint32_t pos_bits, unsigned flags) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && b, "rb_shard_hash_begin: bad argument");
        RB_REQUIRE(b->device == g->p.device, "batch lives on device %d, shard on %d", b->device, g->p.device);
        RB_REQUIRE(first >= 0 && n >= 0 && first + n <= b->n_reads, "rb_shard_hash_begin: bad read range");
        RB_REQUIRE(pos_bits >= 1 && pos_bits <= 31 && ((uint64_t)(b->max_len >= (uint32_t)g->k ? b->max_len - (uint32_t)g->k : 0u) >> pos_bits) == 0, "rb_shard_hash_begin: pos_bits too small for the reads");
        RB_HIP(hipSetDevice(g->p.device));
        g->shard->prep.stage = 0;
        if (g->k > 31) return;                       // generic window-hash path: hash_group does it all
        prep_filter(g, b, first, n, ordinal0, pos_bits, flags, g->stream2);
    });
}
// split reads: the window walk + prefilter of THIS rank's slice of the next sub-batch, enqueued on the producer stream.  Call it when the
// cache updates of the current sub-batch are in (rb_shard_cache_apply): the walk then runs beside the conflict phases and the remaining
// exchanges of the current sub-batch; rb_shard_hash of the same sub-batch picks the result up instead of walking again.
int rb_shard_hash_begin_split(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, int64_t own_first, int64_t own_n, uint64_t ordinal0,
                              uint32_t pos_bits, unsigned flags) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && b, "rb_shard_hash_begin_split: bad argument");
        RB_REQUIRE(b->device == g->p.device, "batch lives on device %d, shard on %d", b->device, g->p.device);
        RB_REQUIRE(first >= 0 && n >= 0 && first + n <= b->n_reads && own_first >= first && own_n >= 0 && own_first + own_n <= first + n, "rb_shard_hash_begin_split: bad read range");
        RB_REQUIRE(pos_bits >= 1 && pos_bits <= 31 && ((uint64_t)(b->max_len >= (uint32_t)g->k ? b->max_len - (uint32_t)g->k : 0u) >> pos_bits) == 0, "rb_shard_hash_begin_split: pos_bits too small for the reads");
        RB_HIP(hipSetDevice(g->p.device));
        g->shard->prep.stage = 0;
        if (g->k > 31) return;                       // generic window-hash path: rb_shard_hash does it all
        prep_filter(g, b, first, n, ordinal0, pos_bits, flags, g->stream2, true, own_first, own_n);
    });
}
int rb_shard_hash_emit(rb_graph *g) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard, "rb_shard_hash_emit: bad argument");
        RB_HIP(hipSetDevice(g->p.device));
        prep_emit(g, g->stream2);
    });
}

int rb_shard_hash_group(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, int64_t pair_first, int64_t pair_n,
                        uint64_t ordinal0, uint32_t pos_bits, unsigned flags, int64_t *dreq_counts, int64_t *creq_counts,
                        int64_t *pair_counts, rb_add_stats *stats) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && b && dreq_counts && creq_counts && pair_counts, "rb_shard_hash_group: bad argument");
        RB_REQUIRE(b->device == g->p.device, "batch lives on device %d, shard on %d", b->device, g->p.device);
        RB_REQUIRE(first >= 0 && n >= 0 && first + n <= b->n_reads, "rb_shard_hash_group: bad read range");
        RB_REQUIRE(pair_first >= first && pair_n >= 0 && pair_first + pair_n <= first + n, "rb_shard_hash_group: pair slice outside the sub-batch");
        RB_REQUIRE(pos_bits >= 1 && pos_bits <= 31 && ((uint64_t)(b->max_len >= (uint32_t)g->k ? b->max_len - (uint32_t)g->k : 0u) >> pos_bits) == 0, "rb_shard_hash_group: pos_bits too small for the reads");
        RB_REQUIRE((uint64_t)n < (1ull << (32 - pos_bits)), "rb_shard_hash_group: too many reads for the occurrence id");
        g->shard->sub_reads_bits = std::max(g->shard->sub_reads_bits, log2_ceil((uint64_t)std::max<int64_t>(1, n)));   // (never shrinks: a look-ahead may already be on the next, shorter sub-batch)
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        for (int r = 0; r < S->G; ++r) dreq_counts[r] = creq_counts[r] = pair_counts[r] = 0;
        S->D = 0; S->n_conf = 0; S->n_kept = 0; S->ordinal0 = ordinal0; S->pos_bits = pos_bits;
        g->seq_codes = b->codes; g->seq_woff = b->woff; g->seq_wpr = b->wpr_uniform; g->seq_first = (uint32_t)first;   // occurrence ids -> bases, for the cache stores
        S->slot_bytes[RB_SLOT_PAIR_IDX] = S->slot_bytes[RB_SLOT_DREQ_IDX] = S->slot_bytes[RB_SLOT_DREQ_PROBE] = S->slot_bytes[RB_SLOT_CREQ_IDX] = 0;
        const int mode_hash = g->stranded ? ((flags & RB_ADD_REVCOMP) ? 2 : 0) : 1;
        const int mode = (flags & RB_ADD_COUNT_IF_PRESENT) ? M_COUNT_IF_PRESENT : M_ADD;
        S->mode = mode;
        const bool pairs = (flags & RB_ADD_STORE_READ_PAIRS) != 0;
        if (pairs) RB_REQUIRE(g->rpk.bits && g->read_d > 0, "STORE_READ_PAIRS needs use_read_paired_kmers and a read pair distance > 0");
        FilterView fv = g->view(ordinal0, pos_bits);
        const bool use_cache = g->npf_log2 != 0;
        // ---- every rank walks ALL reads of the sub-batch and keeps the windows whose k-mer it owns ----
        const int64_t w0 = b->h_woff[(size_t)first], nw = (int64_t)b->h_woff[(size_t)(first + n)] - w0;
        uint64_t owned = 0;
        uint32_t N = 0;
        bool prepared = false;
        if (nw > 0) {
            if (g->k <= 31) {   // fast kernels on the producer stream: ownership + prefilter inside the window walk,
                // then a masked emit + grouping (the one-pass k_filter_emit measured 1.5x slower here: see
                // rb_graph.hip add_range).  Usually rb_shard_hash_begin/_emit did this ahead of time.
                ShardState::Prep &P = S->prep;
                const bool ready = P.stage && P.b == b && P.first == first && P.n == n && P.ordinal0 == ordinal0 && P.pos_bits == pos_bits && P.flags == flags;
                if (!ready) prep_filter(g, b, first, n, ordinal0, pos_bits, flags, g->stream2);   // nothing prepared ahead: do it now
                prep_emit(g, g->stream2);
                prepared = true;
                N = P.N; owned = P.owned;
            } else {            // generic hash of every window, then one ordered compaction
                RB_HIP(hipStreamSynchronize(g->stream2));
                S->prep.stage = 0;
                g->chunk_cnt.reserve(((size_t)nw + 1) * 4); g->chunk_off.reserve(((size_t)nw + 1) * 4);
                g->temp.reserve(scan_temp_bytes((size_t)nw + 1));
                g->npf_tot.reserve(2048);
                RB_HIP(hipMemsetAsync(g->npf_tot.p, 0, 2048, s));
                RB_HIP(hipMemsetAsync(g->chunk_cnt.as<uint32_t>() + nw, 0, 4, s));
                uint32_t spread[16 * 32];
                uint32_t NA = 0;
                launch_count_windows(b, w0, nw, g->k, g->chunk_cnt.as<uint32_t>(), s);
                exclusive_scan_u32(g->temp.p, g->temp.cap, g->chunk_cnt.as<uint32_t>(), g->chunk_off.as<uint32_t>(), (size_t)nw + 1, s);
                RB_HIP(hipMemcpyAsync(&NA, g->chunk_off.as<uint32_t>() + nw, 4, hipMemcpyDeviceToHost, s));
                RB_HIP(hipStreamSynchronize(s));
                if (NA) {
                    S->stage0.reserve((size_t)NA * 8 + 16); S->stage1.reserve((size_t)NA * 4 + 16); S->stage2.reserve((size_t)NA + 16);
                    launch_hash_windows(b, w0, nw, g->k, mode_hash, g->chunk_off.as<uint32_t>(), (uint32_t)first, pos_bits,
                                        S->stage0.as<uint64_t>(), S->stage1.as<uint32_t>(), nullptr, nullptr, s);
                    hipLaunchKernelGGL(k_rec_keep, dim3(blocks_for(NA)), dim3(TPB), 0, s, fv, (int)use_cache, S->stage0.as<uint64_t>(), S->stage1.as<uint32_t>(),
                                       (size_t)NA, own_range(g), S->stage2.as<uint8_t>(), g->npf_tot.as<uint32_t>());
                    g->keys0.reserve((size_t)NA * 8); g->vals0.reserve((size_t)NA * 4);
                    RouteKeep fk{S->stage2.as<uint8_t>(), S->stage0.as<uint64_t>(), S->stage1.as<uint32_t>(), g->keys0.as<uint64_t>(), g->vals0.as<uint32_t>()};
                    int64_t kept_c[1];
                    N = (uint32_t)route(g, fk, (size_t)NA, kept_c, [](RouteKeep &, size_t) {}, 1);
                    RB_HIP(hipMemcpyAsync(spread, g->npf_tot.p, sizeof spread, hipMemcpyDeviceToHost, s));
                    RB_HIP(hipStreamSynchronize(s));
                    for (int q = 0; q < 32; ++q) owned += spread[16 * q];
                }
            }
        }
        uint32_t D_prepared = 0;
        if (prepared) {   // drain the producer stream (it owns chunk_* / keys0 / the other GroupSlot) and switch slots
            D_prepared = group_finish(g, S->prep.slot, g->stream2, g->temp, g->devctr2, s);
            g->cur = S->prep.slot;
            S->prep.stage = 0;
        }
        S->n_kept = N;
        if (stats) { stats->kmers += (int64_t)owned; stats->sorted_kmers += (int64_t)N; }
        // ---- read-paired k-mers of this rank's slice of the reads: ORed into this rank's accumulation copy (merged into the owners'
        //      shards when the call ends, rb_shard_pairs_flush_*), or — routed path — bit indices bucketed by rpkbf owner ----
        if (pairs && pair_n > 0 && (S->rpk_acc.bits || S->pairs_direct)) {
            const int64_t pw0 = b->h_woff[(size_t)pair_first], pnw = (int64_t)b->h_woff[(size_t)(pair_first + pair_n)] - pw0;
            if (pnw > 0) {
                g->devctr.reserve(DEVCTR_BYTES);
                unsigned long long *pc = reinterpret_cast<unsigned long long *>(g->devctr.as<uint32_t>() + 12), np_host = 0;
                RB_HIP(hipMemsetAsync(pc, 0, 8, s));
                launch_pairs(g, b, pw0, pnw, mode_hash, nullptr, nullptr, pc, s, S->pairs_direct ? &g->rpk : &S->rpk_acc);
                RB_HIP(hipMemcpyAsync(&np_host, pc, 8, hipMemcpyDeviceToHost, s));
                RB_HIP(hipStreamSynchronize(s));
                S->acc_dirty = !S->pairs_direct;
                if (stats) stats->pairs += (int64_t)np_host;
            }
        } else
        if (pairs && pair_n > 0) {
            const int64_t pw0 = b->h_woff[(size_t)pair_first], pnw = (int64_t)b->h_woff[(size_t)(pair_first + pair_n)] - pw0;
            if (pnw > 0) {
                uint32_t P = 0;
                g->chunk_cnt.reserve(((size_t)pnw + 1) * 4); g->chunk_off.reserve(((size_t)pnw + 1) * 4);
                g->temp.reserve(scan_temp_bytes((size_t)pnw + 1));
                RB_HIP(hipMemsetAsync(g->chunk_cnt.as<uint32_t>() + pnw, 0, 4, s));
                launch_count_windows(b, pw0, pnw, g->k + g->read_d, g->chunk_cnt.as<uint32_t>(), s);
                exclusive_scan_u32(g->temp.p, g->temp.cap, g->chunk_cnt.as<uint32_t>(), g->chunk_off.as<uint32_t>(), (size_t)pnw + 1, s);
                RB_HIP(hipMemcpyAsync(&P, g->chunk_off.as<uint32_t>() + pnw, 4, hipMemcpyDeviceToHost, s));
                RB_HIP(hipStreamSynchronize(s));
                const size_t np = (size_t)P * (size_t)g->rpk.num_hash;
                if (np) {
                    S->stage0.reserve(np * 8);
                    g->devctr.reserve(DEVCTR_BYTES);
                    unsigned long long *pc = reinterpret_cast<unsigned long long *>(g->devctr.as<uint32_t>() + 12);
                    RB_HIP(hipMemsetAsync(pc, 0, 8, s));
                    launch_pairs(g, b, pw0, pnw, mode_hash, g->chunk_off.as<uint32_t>(), S->stage0.as<uint64_t>(), pc);
                    RouteIdx f{S->stage0.as<uint64_t>(), nullptr, (uint64_t)S->span[RB_RPKBF], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
                    route(g, f, np, pair_counts, [&](RouteIdx &ff, size_t kept) { ff.out_idx = (uint64_t *)slot_reserve(S, RB_SLOT_PAIR_IDX, kept * 8); });
                    if (stats) stats->pairs += (int64_t)P;
                }
            }
        }
        if (stats) stats->reads += pair_n;
        // ---- group this rank's records into runs; requests bucketed by filter owner ----
        if (N) {
            const uint32_t D = prepared ? D_prepared : group_records(g, (size_t)N, ordinal0, pos_bits, nullptr, nullptr);
            S->D = D;
            make_and_route_requests(g, D, mode, ordinal0, pos_bits, dreq_counts, creq_counts);
        }
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

// ---- split-reads mode: each rank hashes ITS slice of the reads (prefilter against a replicated cache),
//      the surviving records travel to the k-mer owners ----
int rb_shard_set_cache_replication(rb_graph *g, int on) {
    if (!g || !g->shard) { set_error("rb_shard_set_cache_replication: not a sharded graph"); return RB_ERR_INVALID; }
    g->shard->replicate_cache = on != 0;
    // the minimizer-bucketed cache pays where a large share of the hashed windows is looked up (split reads, one rank, or
    // replicated hashing on two ranks: 408 -> 397 ms per rank); with more ranks hashing everything, most windows are only
    // hashed, and its LDS footprint slows exactly that part down
    g->use_mpf = g->mpf_log2b && (uint32_t)g->k >= g->mpf_m && (uint32_t)g->k - g->mpf_m + 1u <= 16u && (on || g->shard_count <= 2 || (getenv("RB_SHARD_MPF") && atoi(getenv("RB_SHARD_MPF"))));
    return RB_OK;
}
int rb_shard_hash(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, int64_t own_first, int64_t own_n, uint64_t ordinal0,
                  uint32_t pos_bits, unsigned flags, int64_t *rec_counts, int64_t *pair_counts, rb_add_stats *stats) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && b && rec_counts && pair_counts, "rb_shard_hash: bad argument");
        RB_REQUIRE(b->device == g->p.device, "batch lives on device %d, shard on %d", b->device, g->p.device);
        RB_REQUIRE(first >= 0 && n >= 0 && first + n <= b->n_reads, "rb_shard_hash: bad read range");
        RB_REQUIRE(own_first >= first && own_n >= 0 && own_first + own_n <= first + n, "rb_shard_hash: own slice outside the sub-batch");
        RB_REQUIRE(pos_bits >= 1 && pos_bits <= 31 && ((uint64_t)(b->max_len >= (uint32_t)g->k ? b->max_len - (uint32_t)g->k : 0u) >> pos_bits) == 0, "rb_shard_hash: pos_bits too small for the reads");
        RB_REQUIRE((uint64_t)n < (1ull << (32 - pos_bits)), "rb_shard_hash: too many reads for the occurrence id");
        g->shard->sub_reads_bits = std::max(g->shard->sub_reads_bits, log2_ceil((uint64_t)std::max<int64_t>(1, n)));   // (never shrinks: a look-ahead may already be on the next, shorter sub-batch)
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        RB_HIP(hipStreamSynchronize(g->stream2));
        // the window walk of this very slice, done ahead on the producer stream (rb_shard_hash_begin_split)?
        const ShardState::Prep &PP = S->prep;
        const bool prepared = PP.stage == 1 && PP.split && PP.b == b && PP.first == first && PP.n == n && PP.own_first == own_first && PP.own_n == own_n &&
                              PP.ordinal0 == ordinal0 && PP.pos_bits == pos_bits && PP.flags == flags && g->k <= 31;
        void *prepared_wstate = prepared ? PP.wstate : nullptr;
        S->prep.stage = 0;
        g->seq_codes = b->codes; g->seq_woff = b->woff; g->seq_wpr = b->wpr_uniform; g->seq_first = (uint32_t)first;   // occurrence ids -> bases, for the cache stores
        hipStream_t s = g->stream;
        for (int r = 0; r < S->G; ++r) rec_counts[r] = pair_counts[r] = 0;
        S->slot_bytes[RB_SLOT_REC_KEYS] = S->slot_bytes[RB_SLOT_REC_OCC] = S->slot_bytes[RB_SLOT_PAIR_IDX] = 0;
        const int mode_hash = g->stranded ? ((flags & RB_ADD_REVCOMP) ? 2 : 0) : 1;
        const bool pairs = (flags & RB_ADD_STORE_READ_PAIRS) != 0;
        if (pairs) RB_REQUIRE(g->rpk.bits && g->read_d > 0, "STORE_READ_PAIRS needs use_read_paired_kmers and a read pair distance > 0");
        const int64_t w0 = b->h_woff[(size_t)own_first], nw = (int64_t)b->h_woff[(size_t)(own_first + own_n)] - w0;
        if (stats) stats->reads += own_n;
        if (nw <= 0) return;
        FilterView fv = g->view(ordinal0, pos_bits);
        const bool use_cache = g->npf_log2 != 0;
        g->chunk_cnt.reserve(((size_t)nw + 1) * 4); g->chunk_off.reserve(((size_t)nw + 1) * 4);
        g->temp.reserve(scan_temp_bytes((size_t)nw + 1));
        g->npf_tot.reserve(2048);
        if (!prepared) {
            RB_HIP(hipMemsetAsync(g->npf_tot.p, 0, 2048, s));
            RB_HIP(hipMemsetAsync(g->chunk_cnt.as<uint32_t>() + nw, 0, 4, s));
        }
        uint32_t spread[16 * 32], N = 0;
        uint64_t windows = 0;
        const uint8_t *keep = nullptr;
        const uint64_t *rk = nullptr; const uint32_t *ro = nullptr;
        // occurrence id and generator ordinal are relative to the GLOBAL sub-batch start `first`
        if (g->k <= 31) {
            g->chunk_mask.reserve(((size_t)nw + 1) * 4);
            void *wstate = prepared_wstate;
            if (prepared) {            // counts, masks, offsets and rolling states are in place (the producer stream was drained above)
                N = S->pinned[0];
                for (int q = 0; q < 32; ++q) windows += S->pinned[16 + 16 * q];
            } else {
                Npf cache = fv.npf;
                if (!use_cache) cache.tab = nullptr;
                g->prof_begin();
                if (filter_saves_state(b, nw, g->k)) { g->wstate.reserve(((size_t)nw + 1) * 16); wstate = g->wstate.p; }
                launch_filter_windows(b, w0, nw, g->k, mode_hash, (uint32_t)first, pos_bits, g->p.rng_seed, ordinal0, cache,
                                      g->chunk_cnt.as<uint32_t>(), g->chunk_mask.as<uint32_t>(), g->npf_tot.as<uint32_t>(), s, OwnRange{Mod{1, 0, 0}, 0, 0}, fv.mpf, wstate);
                exclusive_scan_u32(g->temp.p, g->temp.cap, g->chunk_cnt.as<uint32_t>(), g->chunk_off.as<uint32_t>(), (size_t)nw + 1, s);
                RB_HIP(hipMemcpyAsync(&N, g->chunk_off.as<uint32_t>() + nw, 4, hipMemcpyDeviceToHost, s));
                RB_HIP(hipMemcpyAsync(spread, g->npf_tot.p, sizeof spread, hipMemcpyDeviceToHost, s));
                g->prof_end("filter_windows");
                RB_HIP(hipStreamSynchronize(s));
                for (int q = 0; q < 32; ++q) windows += spread[16 * q];
            }
            if (N) {
                g->keys0.reserve((size_t)N * 8); g->vals0.reserve((size_t)N * 4);
                g->prof_begin();
                launch_hash_windows_masked(b, w0, nw, g->k, mode_hash, g->chunk_off.as<uint32_t>(), g->chunk_mask.as<uint32_t>(),
                                           (uint32_t)first, pos_bits, g->keys0.as<uint64_t>(), g->vals0.as<uint32_t>(), s, wstate);
                g->prof_end("hash_windows");
                rk = g->keys0.as<uint64_t>(); ro = g->vals0.as<uint32_t>();
            }
        } else {            // generic hash of every window; the prefilter verdict rides along into the bucketing
            launch_count_windows(b, w0, nw, g->k, g->chunk_cnt.as<uint32_t>(), s);
            exclusive_scan_u32(g->temp.p, g->temp.cap, g->chunk_cnt.as<uint32_t>(), g->chunk_off.as<uint32_t>(), (size_t)nw + 1, s);
            RB_HIP(hipMemcpyAsync(&N, g->chunk_off.as<uint32_t>() + nw, 4, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
            windows = N;
            if (N) {
                g->keys0.reserve((size_t)N * 8); g->vals0.reserve((size_t)N * 4); S->stage2.reserve((size_t)N + 16);
                launch_hash_windows(b, w0, nw, g->k, mode_hash, g->chunk_off.as<uint32_t>(), (uint32_t)first, pos_bits,
                                    g->keys0.as<uint64_t>(), g->vals0.as<uint32_t>(), nullptr, nullptr, s);
                hipLaunchKernelGGL(k_rec_keep, dim3(blocks_for(N)), dim3(TPB), 0, s, fv, (int)use_cache, g->keys0.as<uint64_t>(), g->vals0.as<uint32_t>(),
                                   (size_t)N, OwnRange{Mod{1, 0, 0}, 0, 0}, S->stage2.as<uint8_t>(), g->npf_tot.as<uint32_t>());
                rk = g->keys0.as<uint64_t>(); ro = g->vals0.as<uint32_t>(); keep = S->stage2.as<uint8_t>();
            }
        }
        size_t kept = 0;
        if (N) {
            RouteRec fr{rk, ro, keep, g->cbf_mod, make_own_span((uint64_t)S->span[RB_CBF]), nullptr, nullptr};
            kept = route(g, fr, (size_t)N, rec_counts, [&](RouteRec &ff, size_t k_) {
                ff.out_keys = (uint64_t *)slot_reserve(S, RB_SLOT_REC_KEYS, k_ * 8);
                ff.out_occ = (uint32_t *)slot_reserve(S, RB_SLOT_REC_OCC, k_ * 4);
            });
        }
        if (stats) { stats->kmers += (int64_t)windows; stats->sorted_kmers += (int64_t)kept; }
        if (pairs && (S->rpk_acc.bits || S->pairs_direct)) {   // replicated accumulation: the single-GPU walker into this rank's copy (see ShardState::rpk_acc) — or, one rank, into the filter
            g->devctr.reserve(DEVCTR_BYTES);
            unsigned long long *pc = reinterpret_cast<unsigned long long *>(g->devctr.as<uint32_t>() + 12), np_host = 0;
            RB_HIP(hipMemsetAsync(pc, 0, 8, s));
            launch_pairs(g, b, w0, nw, mode_hash, nullptr, nullptr, pc, s, S->pairs_direct ? &g->rpk : &S->rpk_acc);
            RB_HIP(hipMemcpyAsync(&np_host, pc, 8, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
            S->acc_dirty = !S->pairs_direct;
            if (stats) stats->pairs += (int64_t)np_host;
        } else if (pairs) {
            uint32_t P = 0;
            RB_HIP(hipMemsetAsync(g->chunk_cnt.as<uint32_t>() + nw, 0, 4, s));
            launch_count_windows(b, w0, nw, g->k + g->read_d, g->chunk_cnt.as<uint32_t>(), s);
            exclusive_scan_u32(g->temp.p, g->temp.cap, g->chunk_cnt.as<uint32_t>(), g->chunk_off.as<uint32_t>(), (size_t)nw + 1, s);
            RB_HIP(hipMemcpyAsync(&P, g->chunk_off.as<uint32_t>() + nw, 4, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
            const size_t np = (size_t)P * (size_t)g->rpk.num_hash;
            if (np) {
                S->stage0.reserve(np * 8);
                g->devctr.reserve(DEVCTR_BYTES);
                unsigned long long *pc = reinterpret_cast<unsigned long long *>(g->devctr.as<uint32_t>() + 12);
                RB_HIP(hipMemsetAsync(pc, 0, 8, s));
                launch_pairs(g, b, w0, nw, mode_hash, g->chunk_off.as<uint32_t>(), S->stage0.as<uint64_t>(), pc);
                RouteIdx f{S->stage0.as<uint64_t>(), nullptr, (uint64_t)S->span[RB_RPKBF], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
                route(g, f, np, pair_counts, [&](RouteIdx &ff, size_t k_) { ff.out_idx = (uint64_t *)slot_reserve(S, RB_SLOT_PAIR_IDX, k_ * 8); });
                if (stats) stats->pairs += (int64_t)P;
            }
        }
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}
// the records this rank received (concatenated in source-rank order = read order) -> runs -> requests
int rb_shard_group(rb_graph *g, void *keys_dev, void *occ_dev, int64_t n, uint64_t ordinal0, uint32_t pos_bits,
                   unsigned flags, int64_t *dreq_counts, int64_t *creq_counts) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && dreq_counts && creq_counts && n >= 0, "rb_shard_group: bad argument");
        RB_REQUIRE(n < ((int64_t)1 << 30), "rb_shard_group: %lld records in one sub-batch", (long long)n);
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        for (int r = 0; r < S->G; ++r) dreq_counts[r] = creq_counts[r] = 0;
        S->D = 0; S->n_conf = 0; S->n_kept = (uint64_t)n; S->ordinal0 = ordinal0; S->pos_bits = pos_bits;
        S->slot_bytes[RB_SLOT_DREQ_IDX] = S->slot_bytes[RB_SLOT_DREQ_PROBE] = S->slot_bytes[RB_SLOT_CREQ_IDX] = 0;
        const int mode = (flags & RB_ADD_COUNT_IF_PRESENT) ? M_COUNT_IF_PRESENT : M_ADD;
        S->mode = mode;
        if (n == 0) return;
        // grouped where they arrived: the receive buffers are the caller's scratch until its next exchange (both drivers), and the
        // grouping clobbers its input anyway — no copy into keys0 / vals0 (24 GB per config-2 pass over all ranks)
        // (one-shot pointers: dropped on every way out of this call, so that a failure before group_enqueue consumes them cannot leave
        //  them for the next, unrelated grouping to read and clobber)
        struct DropInputs { rb_graph *g; ~DropInputs() { g->group_in_keys = nullptr; g->group_in_vals = nullptr; } } drop_inputs{g};
        g->group_in_keys = static_cast<uint64_t *>(keys_dev);
        g->group_in_vals = static_cast<uint32_t *>(occ_dev);
        const uint32_t D = group_records(g, (size_t)n, ordinal0, pos_bits, nullptr, nullptr);
        S->D = D;
        make_and_route_requests(g, D, mode, ordinal0, pos_bits, dreq_counts, creq_counts);
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}
int rb_shard_cache_apply(rb_graph *g, const void *upd_dev, int64_t n) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && n >= 0, "rb_shard_cache_apply: bad argument");
        if (!n || !(g->npf_log2 || g->mpf_log2b)) return;
        RB_HIP(hipSetDevice(g->p.device));
        FilterView fv = g->view(0, 0);
        hipLaunchKernelGGL(k_cache_apply, dim3(blocks_for(n)), dim3(TPB), 0, g->stream, fv, (const CacheUpd *)upd_dev, (size_t)n);
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(g->stream));
    });
}

// ---- read-pair filter: merge of the ranks' accumulation copies into the owners' shards (end of an insert call) ----
namespace {
__global__ void k_or_u64(unsigned long long *__restrict__ dst, const unsigned long long *__restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const unsigned long long v = src[i]; if (v) dst[i] |= v; }
}
__global__ void k_or_u8(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint8_t v = src[i]; if (v) dst[i] |= v; }
}
}  // namespace
int rb_shard_pairs_flush_begin(rb_graph *g, void **send_dev, int64_t *counts) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && send_dev && counts, "rb_shard_pairs_flush_begin: bad argument");
        ShardState *S = g->shard;
        *send_dev = nullptr;
        for (int r = 0; r < S->G; ++r) counts[r] = 0;
        if (!S->rpk_acc.bits) return;            // routed path: nothing to merge.  (Every rank of a collective insert call flushes: the
        RB_HIP(hipSetDevice(g->p.device));       //  pieces travel whether or not this rank's slice held a pair.)
        RB_HIP(hipStreamSynchronize(g->stream));
        const int64_t span_bytes = S->span[RB_RPKBF] / 8, total = S->rpk_acc.nbytes;      // spans are multiples of 64 bits
        for (int r = 0; r < S->G; ++r) counts[r] = std::max<int64_t>(0, std::min<int64_t>(total, span_bytes * (r + 1)) - std::min<int64_t>(total, span_bytes * r));
        *send_dev = S->rpk_acc.bits;
    });
}
int rb_shard_pairs_flush_end(rb_graph *g, const void *recv_dev, const int64_t *recv_counts) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && recv_counts, "rb_shard_pairs_flush_end: bad argument");
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        const int64_t mine = g->rpk.bits ? g->rpk.nbytes : 0;
        int64_t off = 0;
        for (int r = 0; r < S->G; ++r) {
            const int64_t n = recv_counts[r];                 // (0: that rank routes its pairs and has nothing to merge)
            RB_REQUIRE(n == mine || n == 0, "rb_shard_pairs_flush_end: rank %d sent %lld bytes of this rank's range, which has %lld", r, (long long)n, (long long)mine);
            const char *src = static_cast<const char *>(recv_dev) + off;
            if (n && (n % 8) == 0 && (reinterpret_cast<uintptr_t>(src) % 8) == 0)
                hipLaunchKernelGGL(k_or_u64, dim3(blocks_for(n / 8)), dim3(TPB), 0, s, reinterpret_cast<unsigned long long *>(g->rpk.bits),
                                   reinterpret_cast<const unsigned long long *>(src), (size_t)(n / 8));
            else if (n)
                hipLaunchKernelGGL(k_or_u8, dim3(blocks_for(n)), dim3(TPB), 0, s, reinterpret_cast<uint8_t *>(g->rpk.bits), reinterpret_cast<const uint8_t *>(src), (size_t)n);
            off += n;
        }
        // The copy KEEPS its bits (round 5; it was cleared here before): they are in the owners' shards now, merging them again with the next
        // call changes nothing (OR), and the walker's seen-pair cache — which vouches for bits of this copy — stays valid from one file of a
        // library to the next.  Whatever empties or replaces the pair filter empties the copy and the cache with it (shard_clear_pairs_acc:
        // rb_graph_clear, rb_filter_import).
        S->acc_dirty = false;
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

int rb_shard_serve(rb_graph *g, int mode, const void *dreq_idx_dev, const void *dreq_probe_dev, int64_t nd,
                   const void *creq_idx_dev, int64_t nc, const void *pair_idx_dev, int64_t np, void *dreply_dev, void *creply_dev) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && nd >= 0 && nc >= 0 && np >= 0, "rb_shard_serve: bad argument");
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        // this rank's own runs take part too: their local probes were tested / claimed by k_shard_probe and are arbitrated here
        // together with the requests of the other ranks (same collision table, same ids, same contested-counter table)
        const uint32_t D = S->D;
        const FilterView fv = g->view(S->ordinal0, S->pos_bits);
        const LocalRanges R = local_ranges(g);
        const uint64_t *uniq = g->uniq().as<uint64_t>();
        const uint32_t *starts = g->starts().as<uint32_t>(), *vals = g->vals1().as<uint32_t>(), *status = g->status.as<uint32_t>();
        const int uses_f = (mode == M_ADD || mode == M_ADD_IF_ABSENT);
        S->has_f = S->has_cs = false;
        g->devctr.reserve(DEVCTR_BYTES);
        uint32_t *ctr = g->devctr.as<uint32_t>();
        const uint64_t *didx = (const uint64_t *)dreq_idx_dev, *dprobe = (const uint64_t *)dreq_probe_dev;
        uint8_t *dreply = (uint8_t *)dreply_dev;
        if (nd) hipLaunchKernelGGL(k_own_dbg_test, dim3(blocks_for(nd)), dim3(TPB), 0, s, g->dbg.bits, (uint64_t)g->dbg.lo, didx, (size_t)nd, dreply);
        if (uses_f && (nd || D)) {
            RB_HIP(hipMemsetAsync(ctr, 0, DEVCTR_BYTES, s));
            if (nd) hipLaunchKernelGGL(k_own_dbg_set, dim3(blocks_for(nd)), dim3(TPB), 0, s, g->dbg.bits, (uint64_t)g->dbg.lo, didx, (size_t)nd, dreply, ctr + 16);
            if (D) hipLaunchKernelGGL(k_local_dbg_set, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, R, uniq, D, status, S->lmask.as<uint16_t>(), S->lcoll.as<uint8_t>(), ctr + 17);
            uint32_t n_collide = 0, spread[16 * 32];
            RB_HIP(hipMemcpyAsync(spread, ctr + 16, sizeof spread, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
            for (int q = 0; q < 32; ++q) n_collide += spread[16 * q] + ((16 * q + 1 < 16 * 32) ? spread[16 * q + 1] : 0u);
            Slot *ftab = nullptr;
            uint32_t f_log2 = 1;
            if (n_collide) {
                f_log2 = log2_ceil(4ull * (uint64_t)n_collide + 2);
                S->own_f.reserve(sizeof(Slot) << f_log2);
                RB_HIP(hipMemsetAsync(S->own_f.p, 0xFF, sizeof(Slot) << f_log2, s));
                ftab = S->own_f.as<Slot>();
                if (nd) hipLaunchKernelGGL(k_own_collide_insert, dim3(blocks_for(nd)), dim3(TPB), 0, s, didx, dprobe, (size_t)nd, dreply, ftab, f_log2);
                if (D) hipLaunchKernelGGL(k_local_collide_insert, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, uniq, starts, vals, D, S->lcoll.as<uint8_t>(), ftab, f_log2);
                if (nd) hipLaunchKernelGGL(k_own_collide_fixup, dim3(blocks_for(nd)), dim3(TPB), 0, s, didx, dprobe, (size_t)nd, dreply, ftab, f_log2);
                if (D) hipLaunchKernelGGL(k_local_collide_fixup, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, uniq, starts, vals, D, status, S->lmask.as<uint16_t>(),
                                          S->lcoll.as<uint8_t>(), ftab, f_log2);
                S->has_f = true; S->f_log2 = f_log2;
            }
            if (nd) hipLaunchKernelGGL(k_own_dbg_first, dim3(blocks_for(nd)), dim3(TPB), 0, s, didx, dprobe, (size_t)nd, ftab, f_log2, dreply);
        }
        if (nc || D) {
            RB_HIP(hipMemsetAsync(ctr, 0, DEVCTR_BYTES, s));
            if (nc) hipLaunchKernelGGL(k_own_claim, dim3(blocks_for(nc)), dim3(TPB), 0, s, g->cbf, (uint64_t)g->cbf_lo, (const uint64_t *)creq_idx_dev,
                                       (size_t)nc, (uint8_t *)creply_dev, ctr + 16);
            uint32_t nf = 0, spread[16 * 32], lspread[16 * 32];
            RB_HIP(hipMemcpyAsync(spread, ctr + 16, sizeof spread, hipMemcpyDeviceToHost, s));
            if (D) RB_HIP(hipMemcpyAsync(lspread, S->lctr.p, sizeof lspread, hipMemcpyDeviceToHost, s));      // local claims that met a mark (k_shard_probe)
            RB_HIP(hipStreamSynchronize(s));
            for (int q = 0; q < 32; ++q) nf += spread[16 * q] + (D ? lspread[16 * q] : 0u);
            if (nf) {
                const uint32_t cs_log2 = log2_ceil(2ull * nf + 2);
                const uint32_t csf_log2 = std::max(16u, std::min(25u, log2_ceil(8ull * (uint64_t)nf)));
                const size_t tab_bytes = sizeof(Slot) << cs_log2;
                S->own_cs.reserve(tab_bytes + ((size_t)1 << (csf_log2 - 3)));
                uint32_t *csf = reinterpret_cast<uint32_t *>(static_cast<char *>(S->own_cs.p) + tab_bytes);
                RB_HIP(hipMemsetAsync(S->own_cs.p, 0xFF, tab_bytes, s));
                RB_HIP(hipMemsetAsync(csf, 0, (size_t)1 << (csf_log2 - 3), s));
                if (nc) hipLaunchKernelGGL(k_own_cs_build, dim3(blocks_for(nc)), dim3(TPB), 0, s, (const uint64_t *)creq_idx_dev, (const uint8_t *)creply_dev,
                                           (size_t)nc, S->own_cs.as<Slot>(), cs_log2, csf, csf_log2);
                if (D) hipLaunchKernelGGL(k_local_cs_build, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, uniq, D, S->lmask.as<uint16_t>(), S->lcv.as<uint64_t>(),
                                          S->own_cs.as<Slot>(), cs_log2, csf, csf_log2);
                if (nc) hipLaunchKernelGGL(k_own_claim_fin, dim3(blocks_for(nc)), dim3(TPB), 0, s, (const uint64_t *)creq_idx_dev, (size_t)nc,
                                           S->own_cs.as<Slot>(), cs_log2, csf, csf_log2, (uint8_t *)creply_dev);
                S->has_cs = true; S->cs_log2 = cs_log2; S->csf_log2 = csf_log2;
            }
        }
        if (np) {
            RB_REQUIRE(g->rpk.bits, "rb_shard_serve: pair probes but no pair filter");
            hipLaunchKernelGGL(k_own_bits, dim3(blocks_for(np)), dim3(TPB), 0, s, g->rpk.bits, (uint64_t)g->rpk.lo, (const uint64_t *)pair_idx_dev, (size_t)np);
        }
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

int rb_shard_resolve(rb_graph *g, int mode, const void *dreply_dev, const void *creply_dev, int64_t *w_counts,
                     int64_t *ord_counts, rb_add_stats *stats) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && w_counts && ord_counts, "rb_shard_resolve: bad argument");
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        for (int r = 0; r < S->G; ++r) w_counts[r] = ord_counts[r] = 0;
        S->n_conf = 0; S->n_cand = 0;
        S->slot_bytes[RB_SLOT_W_IDX] = S->slot_bytes[RB_SLOT_W_VAL] = S->slot_bytes[RB_SLOT_CONF_EDGES] = S->slot_bytes[RB_SLOT_CACHE_UPD] = S->slot_bytes[RB_SLOT_ORD_IDX] = 0;
        const uint32_t D = S->D;
        if (!D) return;
        FilterView fv = g->view(S->ordinal0, S->pos_bits);
        g->status.reserve((size_t)D * 4); g->nops.reserve((size_t)D * 4); g->cvals.reserve((size_t)D * 8);
        g->heavy.reserve((size_t)D * 4); S->conf_list.reserve((size_t)D * 4); S->cfinal.reserve((size_t)D * 8);
        if (S->replicate_cache) S->cache_upd.reserve((size_t)D + 16);
        g->devctr.reserve(DEVCTR_BYTES);
        uint32_t *ctr = g->devctr.as<uint32_t>();
        RB_HIP(hipMemsetAsync(ctr, 0, DEVCTR_BYTES, s));
        LocalTables T{S->has_f ? S->own_f.as<Slot>() : nullptr, S->f_log2, S->has_cs ? S->own_cs.as<Slot>() : nullptr, S->cs_log2,
                      S->has_cs ? reinterpret_cast<const uint32_t *>(static_cast<const char *>(S->own_cs.p) + (sizeof(Slot) << S->cs_log2)) : nullptr, S->csf_log2};
        hipLaunchKernelGGL(k_resolve_shard, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, g->counts().as<uint32_t>(), g->starts().as<uint32_t>(), D, mode,
                           g->light_ops, S->dreq_pos.as<uint32_t>(), (const uint8_t *)dreply_dev, S->creq_pos.as<uint32_t>(),
                           S->creq_dup.as<uint8_t>(), (const uint8_t *)creply_dev, g->tz().as<uint8_t>(), g->uniq().as<uint64_t>(),
                           g->status.as<uint32_t>(), g->nops.as<uint32_t>(), g->cvals.as<uint64_t>(), S->cfinal.as<uint64_t>(),
                           S->replicate_cache ? S->cache_upd.as<uint8_t>() : (uint8_t *)nullptr, g->vals1().as<uint32_t>(),
                           S->lmask.as<uint16_t>(), S->lcv.as<uint64_t>(), T);
        g->temp.reserve(select2_temp_bytes(D));
        select_flagged2(g->temp.p, g->temp.cap, g->status.as<uint32_t>(), D, RUN_HEAVY, g->heavy.as<uint32_t>(), RUN_CONFLICT, S->conf_list.as<uint32_t>(), ctr + 0, s);
        uint32_t hc[2];
        RB_HIP(hipMemcpyAsync(hc, ctr, 8, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        if (hc[0])
            hipLaunchKernelGGL(k_cbf_heavy, dim3(std::min<uint32_t>(hc[0], 262144u)), dim3(64), 0, s, fv, g->uniq().as<uint64_t>(), g->counts().as<uint32_t>(),
                               g->starts().as<uint32_t>(), g->vals1().as<uint32_t>(), g->status.as<uint32_t>(), g->nops.as<uint32_t>(),
                               g->cvals.as<uint64_t>(), g->tz().as<uint8_t>(), g->heavy.as<uint32_t>(), ctr, S->cfinal.as<uint64_t>(),
                               S->replicate_cache ? S->cache_upd.as<uint8_t>() : (uint8_t *)nullptr);
        if (S->replicate_cache) {   // what this sub-batch taught the prefilter cache, for every other rank's replica
            RouteUpd fu{S->cache_upd.as<uint8_t>(), g->uniq().as<uint64_t>(), g->starts().as<uint32_t>(), g->vals1().as<uint32_t>(), fv, nullptr};
            int64_t uc[1];
            route(g, fu, (size_t)D, uc, [&](RouteUpd &ff, size_t kept) { ff.out = (CacheUpd *)slot_reserve(S, RB_SLOT_CACHE_UPD, kept * sizeof(CacheUpd)); }, 1);
        }
        // counter writes / releases of the runs that are finished, bucketed by counter owner
        const size_t nc = (size_t)D * fv.cbf_h;
        S->stage0.reserve(nc * 8 + 16); S->stage2.reserve(2 * nc + 32);
        uint8_t *w_val = S->stage2.as<uint8_t>(), *w_drop = w_val + nc;
        hipLaunchKernelGGL(k_emit_writes, dim3(blocks_for(D)), dim3(TPB), 0, s, fv, local_ranges(g), g->uniq().as<uint64_t>(), D, g->status.as<uint32_t>(),
                           S->creq_dup.as<uint8_t>(), S->cfinal.as<uint64_t>(), S->lmask.as<uint16_t>(), S->stage0.as<uint64_t>(), w_val, w_drop);
        RouteIdx fw{S->stage0.as<uint64_t>(), w_drop, (uint64_t)S->span[RB_CBF], nullptr, w_val, nullptr, nullptr, nullptr, nullptr};
        route_any_order(g, fw, nc, w_counts, [&](RouteIdx &ff, size_t kept) {
            ff.out_idx = (uint64_t *)slot_reserve(S, RB_SLOT_W_IDX, kept * 8);
            ff.out8 = (uint8_t *)slot_reserve(S, RB_SLOT_W_VAL, kept);
        });
        // the runs with a contested counter: which of them can reach one — those ask the counters' owners who else can (rb_shard_order_*)
        if (hc[1]) {
            const uint32_t nck = hc[1];
            RB_REQUIRE((uint64_t)nck * (uint64_t)S->G < (1ull << 32), "too many runs with a contested counter in one sub-batch (%u)", nck);
            S->n_cand = nck;
            const size_t nq = (size_t)nck * fv.cbf_h;
            S->oinfo.reserve((size_t)nck + 16); S->ord_pos.reserve(nq * 4 + 16);
            S->stage3.reserve(nq * 8 + 16); S->stage1.reserve(nq + 16);
            hipLaunchKernelGGL(k_order_requests, dim3(blocks_for(nck)), dim3(TPB), 0, s, fv, g->uniq().as<uint64_t>(), S->conf_list.as<uint32_t>(), nck,
                               g->status.as<uint32_t>(), g->nops.as<uint32_t>(), g->cvals.as<uint64_t>(), S->creq_dup.as<uint8_t>(), S->oinfo.as<uint8_t>(),
                               S->stage3.as<uint64_t>(), S->stage1.as<uint8_t>());
            RouteIdx fo{S->stage3.as<uint64_t>(), S->stage1.as<uint8_t>(), (uint64_t)S->span[RB_CBF], nullptr, nullptr, nullptr, nullptr, nullptr, S->ord_pos.as<uint32_t>()};
            route_any_order(g, fo, nq, ord_counts, [&](RouteIdx &ff, size_t kept) { ff.out_idx = (uint64_t *)slot_reserve(S, RB_SLOT_ORD_IDX, kept * 8); });
        }
        if (stats) stats->distinct += D;
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

// owner: how many claimants that can reach something contested does each asked counter have (the table of contested counters of rb_shard_serve)
int rb_shard_order_serve(rb_graph *g, const void *ord_idx_dev, int64_t n, void *reply_dev) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && n >= 0 && (n == 0 || (ord_idx_dev && reply_dev)), "rb_shard_order_serve: bad argument");
        if (!n) return;
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        RB_REQUIRE(S->has_cs, "rb_shard_order_serve: %lld requests but this rank found no contested counter in the serve phase", (long long)n);
        hipLaunchKernelGGL(k_order_count, dim3(blocks_for(n)), dim3(TPB), 0, s, (const uint64_t *)ord_idx_dev, (size_t)n, S->own_cs.as<Slot>(), S->cs_log2);
        hipLaunchKernelGGL(k_order_reply, dim3(blocks_for(n)), dim3(TPB), 0, s, (const uint64_t *)ord_idx_dev, (size_t)n, (const Slot *)S->own_cs.as<Slot>(), S->cs_log2,
                           (uint8_t *)reply_dev);
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

// requester: the answers are in — the runs in O* keep RUN_CONFLICT (their edges go to RB_SLOT_CONF_EDGES as before), the others are finished
// here; their writes to other ranks' counters follow the edges in the same slot as tagged records (every rank applies its own: rb_shard_apply_tagged)
int rb_shard_order_finish(rb_graph *g, int mode, const void *ord_reply_dev, int64_t *n_conf_runs, int64_t *n_conf_records, rb_add_stats *stats) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && n_conf_runs && n_conf_records, "rb_shard_order_finish: bad argument");
        (void)mode; (void)stats;
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        *n_conf_runs = *n_conf_records = 0;
        S->n_conf = 0;
        S->slot_bytes[RB_SLOT_CONF_EDGES] = 0;
        const uint32_t nck = S->n_cand;
        if (!nck) return;
        FilterView fv = g->view(S->ordinal0, S->pos_bits);
        uint32_t *ctr = g->devctr.as<uint32_t>();
        S->conf2.reserve((size_t)nck * 4); S->deferred.reserve((size_t)nck * 4); S->heavy2.reserve((size_t)nck * 4);
        RB_HIP(hipMemsetAsync(ctr + 940, 0, 32, s));            // [0] O* runs, [1] finished here, [2] heavy among those, [4] tagged records
        const int order_all = (getenv("RB_SHARD_ORDER_ALL") && atoi(getenv("RB_SHARD_ORDER_ALL"))) ? 1 : 0;      // (A/B: every run with a contested counter is replayed, rounds 1-5)
        hipLaunchKernelGGL(k_order_decide, dim3(blocks_for(nck)), dim3(TPB), 0, s, fv, g->counts().as<uint32_t>(), g->starts().as<uint32_t>(), S->conf_list.as<uint32_t>(), nck,
                           S->oinfo.as<uint8_t>(), S->ord_pos.as<uint32_t>(), (const uint8_t *)ord_reply_dev, g->light_ops, order_all, g->status.as<uint32_t>(),
                           g->nops.as<uint32_t>(), g->cvals.as<uint64_t>(), g->tz().as<uint8_t>(), S->cfinal.as<uint64_t>(), S->conf2.as<uint32_t>(),
                           S->deferred.as<uint32_t>(), S->heavy2.as<uint32_t>(), ctr + 940);
        uint32_t c3[3] = {0, 0, 0};
        RB_HIP(hipMemcpyAsync(c3, ctr + 940, 12, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        const uint32_t n_o = c3[0], n_def = c3[1], n_h2 = c3[2];
        if (n_h2) {
            hipLaunchKernelGGL(k_cbf_heavy, dim3(std::min<uint32_t>(n_h2, 262144u)), dim3(64), 0, s, fv, g->uniq().as<uint64_t>(), g->counts().as<uint32_t>(),
                               g->starts().as<uint32_t>(), g->vals1().as<uint32_t>(), g->status.as<uint32_t>(), g->nops.as<uint32_t>(),
                               g->cvals.as<uint64_t>(), g->tz().as<uint8_t>(), S->heavy2.as<uint32_t>(), ctr + 942, S->cfinal.as<uint64_t>(), (uint8_t *)nullptr);
            hipLaunchKernelGGL(k_order_release_fix, dim3(blocks_for(n_h2)), dim3(TPB), 0, s, (int)fv.cbf_h, S->heavy2.as<uint32_t>(), ctr + 942, g->status.as<uint32_t>(),
                               g->cvals.as<uint64_t>(), S->cfinal.as<uint64_t>());
        }
        // edges of the O* runs (conf_list <- conf2: local run number i of the edge ids counts inside THIS list)
        uint32_t ne = 0;
        if (n_o) {
            RB_HIP(hipMemcpyAsync(S->conf_list.p, S->conf2.p, (size_t)n_o * 4, hipMemcpyDeviceToDevice, s));
            S->n_conf = n_o;
            S->esz.reserve(((size_t)n_o + 1) * 4); S->eoff.reserve(((size_t)n_o + 1) * 4);
            hipLaunchKernelGGL(k_conf_sizes, dim3(blocks_for(n_o + 1)), dim3(TPB), 0, s, S->conf_list.as<uint32_t>(), g->status.as<uint32_t>(), n_o, S->esz.as<uint32_t>());
            g->temp.reserve(scan_temp_bytes((size_t)n_o + 1));
            exclusive_scan_u32(g->temp.p, g->temp.cap, S->esz.as<uint32_t>(), S->eoff.as<uint32_t>(), (size_t)n_o + 1, s);
            RB_HIP(hipMemcpyAsync(&ne, S->eoff.as<uint32_t>() + n_o, 4, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
        }
        ConfEdge *eo = (ConfEdge *)slot_reserve(S, RB_SLOT_CONF_EDGES, ((size_t)ne + (size_t)n_def * fv.cbf_h) * sizeof(ConfEdge));
        if (n_o)
            hipLaunchKernelGGL(k_conf_edges, dim3(blocks_for(n_o)), dim3(TPB), 0, s, fv, g->uniq().as<uint64_t>(), S->conf_list.as<uint32_t>(),
                               g->status.as<uint32_t>(), S->eoff.as<uint32_t>(), n_o, (uint32_t)S->G, (uint32_t)g->shard_rank, eo);
        uint32_t n_tag = 0;
        if (n_def) {
            hipLaunchKernelGGL(k_order_writes, dim3(blocks_for(n_def)), dim3(TPB), 0, s, fv, local_ranges(g), g->uniq().as<uint64_t>(), S->deferred.as<uint32_t>(), ctr + 941,
                               S->creq_dup.as<uint8_t>(), S->cfinal.as<uint64_t>(), S->lmask.as<uint16_t>(), eo + ne, ctr + 944);
            RB_HIP(hipMemcpyAsync(&n_tag, ctr + 944, 4, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
        }
        S->slot_bytes[RB_SLOT_CONF_EDGES] = ((size_t)ne + n_tag) * sizeof(ConfEdge);
        *n_conf_runs = n_o; *n_conf_records = (int64_t)ne + n_tag;
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

// every rank, after the all-gather of the edge slots: the tagged counter writes that fall into this rank's range
int rb_shard_apply_tagged(rb_graph *g, const void *records_dev, int64_t n) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && n >= 0, "rb_shard_apply_tagged: bad argument");
        RB_HIP(hipSetDevice(g->p.device));
        if (n) hipLaunchKernelGGL(k_apply_tagged, dim3(blocks_for(n)), dim3(TPB), 0, g->stream, g->cbf, (uint64_t)g->cbf_lo, (uint64_t)g->cbf_hi, (const ConfEdge *)records_dev, (size_t)n);
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(g->stream));
    });
}

int rb_shard_apply_writes(rb_graph *g, const void *w_idx_dev, const void *w_val_dev, int64_t n) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && n >= 0, "rb_shard_apply_writes: bad argument");
        RB_HIP(hipSetDevice(g->p.device));
        if (n) hipLaunchKernelGGL(k_own_writes, dim3(blocks_for(n)), dim3(TPB), 0, g->stream, g->cbf, (uint64_t)g->cbf_lo, (const uint64_t *)w_idx_dev,
                                  (const uint8_t *)w_val_dev, (size_t)n);
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(g->stream));
    });
}

int rb_shard_conflict_route(rb_graph *g, const void *edges_dev, int64_t n_edges, int64_t gid_bound, int64_t *run_counts,
                            int64_t *op_counts, rb_add_stats *stats) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && run_counts && op_counts && n_edges >= 0 && gid_bound >= 0, "rb_shard_conflict_route: bad argument");
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        for (int r = 0; r < S->G; ++r) run_counts[r] = op_counts[r] = 0;
        S->slot_bytes[RB_SLOT_CONF_RUNS] = S->slot_bytes[RB_SLOT_CONF_OPS] = 0;
        if (n_edges == 0) { RB_REQUIRE(S->n_conf == 0, "rb_shard_conflict_route: conflicting runs but no edges"); return; }
        RB_REQUIRE(gid_bound < ((int64_t)1 << 32) && (int64_t)S->n_conf * S->G <= gid_bound, "rb_shard_conflict_route: gid_bound too small");
        const ConfEdge *e = (const ConfEdge *)edges_dev;
        const size_t ne = (size_t)n_edges;
        // components of the global (run, contested counter) graph — identical on every rank
        const uint32_t log2cap = log2_ceil(2ull * ne + 2);
        S->gid_bits = std::max(1u, log2_ceil((uint64_t)std::max<int64_t>(2, gid_bound)));
        S->etab.reserve(sizeof(Slot) << log2cap); S->eslot.reserve(ne * 4); S->elabel.reserve((size_t)gid_bound * 4 + 16);
        RB_HIP(hipMemsetAsync(S->etab.p, 0xFF, sizeof(Slot) << log2cap, s));
        g->devctr.reserve(DEVCTR_BYTES);
        uint32_t *ctr = g->devctr.as<uint32_t>();
        hipLaunchKernelGGL(k_edge_init, dim3(blocks_for((int64_t)ne)), dim3(TPB), 0, s, e, ne, S->etab.as<Slot>(), log2cap, S->eslot.as<uint32_t>(),
                           S->elabel.as<uint32_t>());
        for (int it = 0;; ++it) {
            RB_REQUIRE(it < 100000, "conflict component labelling did not converge");
            uint32_t *flag = ctr + 900 + (it & 31);                          // a fresh flag per look: 32 of them zeroed by one fill, not one fill per look
            if ((it & 31) == 0) RB_HIP(hipMemsetAsync(ctr + 900, 0, 128, s));
            for (int q = 0; q < (ne >= ((size_t)1 << 20) ? 1 : 2); ++q)      // (a short list: two rounds per look at the flag — most components are pairs)
                hipLaunchKernelGGL(k_edge_round, dim3(blocks_for((int64_t)ne)), dim3(TPB), 0, s, e, ne, S->etab.as<Slot>(), S->eslot.as<uint32_t>(), S->elabel.as<uint32_t>(), flag);
            uint32_t changed = 0;
            RB_HIP(hipMemcpyAsync(&changed, flag, 4, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
            if (!changed) break;
        }
        const uint32_t nck = S->n_conf;
        if (!nck) return;
        // this rank's conflicting runs -> the rank that owns their component
        S->cdesc.reserve((size_t)nck * sizeof(ConfRun)); S->cpos.reserve((size_t)nck * 4);
        S->cnops.reserve(((size_t)nck + 1) * 4); S->cnoff.reserve(((size_t)nck + 1) * 4);
        hipLaunchKernelGGL(k_conf_desc, dim3(blocks_for(nck)), dim3(TPB), 0, s, g->uniq().as<uint64_t>(), S->conf_list.as<uint32_t>(), g->status.as<uint32_t>(),
                           g->nops.as<uint32_t>(), g->cvals.as<uint64_t>(), S->elabel.as<uint32_t>(), nck, (uint32_t)S->G, (uint32_t)g->shard_rank,
                           S->cdesc.as<ConfRun>());
        RB_HIP(hipMemsetAsync(S->cnops.as<uint32_t>() + nck, 0, 4, s));
        RouteRuns fr{S->cdesc.as<ConfRun>(), (uint32_t)S->G, nullptr, S->cpos.as<uint32_t>(), S->cnops.as<uint32_t>()};
        route(g, fr, nck, run_counts, [&](RouteRuns &ff, size_t kept) { ff.out = (ConfRun *)slot_reserve(S, RB_SLOT_CONF_RUNS, kept * sizeof(ConfRun)); });
        g->temp.reserve(scan_temp_bytes((size_t)nck + 1));
        exclusive_scan_u32(g->temp.p, g->temp.cap, S->cnops.as<uint32_t>(), S->cnoff.as<uint32_t>(), (size_t)nck + 1, s);
        // ops per destination = offsets at the run-bucket boundaries
        std::vector<uint64_t> rb(S->G + 1), ob(S->G + 1);
        rb[0] = 0;
        for (int r = 0; r < S->G; ++r) rb[r + 1] = rb[r] + (uint64_t)run_counts[r];
        S->bounds.reserve(2 * (S->G + 2) * 8);
        RB_HIP(hipMemcpyAsync(S->bounds.p, rb.data(), (S->G + 1) * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pick_u32, dim3(1), dim3(128), 0, s, S->cnoff.as<uint32_t>(), S->bounds.as<uint64_t>(), (uint32_t)(S->G + 1),
                           S->bounds.as<uint64_t>() + (S->G + 2));
        RB_HIP(hipMemcpyAsync(ob.data(), S->bounds.as<uint64_t>() + (S->G + 2), (S->G + 1) * 8, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        for (int r = 0; r < S->G; ++r) op_counts[r] = (int64_t)(ob[r + 1] - ob[r]);
        const size_t nco = (size_t)ob[S->G];
        uint32_t *oo = (uint32_t *)slot_reserve(S, RB_SLOT_CONF_OPS, nco * 4);
        hipLaunchKernelGGL(k_conf_ops_out, dim3(blocks_for((int64_t)nck * 64)), dim3(TPB), 0, s, S->conf_list.as<uint32_t>(), g->counts().as<uint32_t>(),
                           g->starts().as<uint32_t>(), g->vals1().as<uint32_t>(), g->nops.as<uint32_t>(), S->cpos.as<uint32_t>(),
                           S->cnoff.as<uint32_t>(), nck, oo);
        if (stats) stats->conflict_ops += (int64_t)nco;
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

int rb_shard_conflict_replay(rb_graph *g, const void *runs_dev, int64_t n_runs, const void *ops_dev, int64_t n_ops, int64_t *w_counts) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && n_runs >= 0 && n_ops >= 0 && w_counts, "rb_shard_conflict_replay: bad argument");
        ShardState *S = g->shard;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        for (int r = 0; r < S->G; ++r) w_counts[r] = 0;
        S->slot_bytes[RB_SLOT_CW_IDX] = S->slot_bytes[RB_SLOT_CW_VAL] = 0;
        if (n_runs == 0) { RB_REQUIRE(n_ops == 0, "rb_shard_conflict_replay: ops without runs"); return; }
        RB_REQUIRE(n_runs < (1ll << 30) && n_ops < (1ll << 32), "rb_shard_conflict_replay: too many runs/ops");
        FilterView fv = g->view(S->ordinal0, S->pos_bits);
        const ConfRun *runs = (const ConfRun *)runs_dev;
        const uint32_t R = (uint32_t)n_runs, O = (uint32_t)n_ops;
        const int h = fv.cbf_h;
        S->cnops.reserve(((size_t)R + 1) * 4); S->cnoff.reserve(((size_t)R + 1) * 4);
        S->rk0.reserve((size_t)R * 8); S->rk1.reserve((size_t)R * 8); S->rbig.reserve(((size_t)R + 1) * 8);
        S->ok0.reserve((size_t)O * 8 + 16); S->ok1.reserve((size_t)O * 8 + 16); S->ov0.reserve((size_t)O * 4 + 16); S->ov1.reserve((size_t)O * 4 + 16);
        hipLaunchKernelGGL(k_run_nops, dim3(blocks_for(R + 1)), dim3(TPB), 0, s, runs, R, S->cnops.as<uint32_t>(), S->rk0.as<uint64_t>());
        g->temp.reserve(std::max({scan_temp_bytes((size_t)R + 1), sort_pairs_temp_bytes((size_t)O + 1), sort_keys_temp_bytes(R), select_temp_bytes(R)}));
        exclusive_scan_u32(g->temp.p, g->temp.cap, S->cnops.as<uint32_t>(), S->cnoff.as<uint32_t>(), (size_t)R + 1, s);
        uint32_t total = 0;
        RB_HIP(hipMemcpyAsync(&total, S->cnoff.as<uint32_t>() + R, 4, hipMemcpyDeviceToHost, s));
        // the component's counters: private table, pre-batch values
        const uint32_t log2cap = log2_ceil(2ull * (uint64_t)R * (uint64_t)h + 2);
        S->rtab.reserve(sizeof(Slot) << log2cap); S->rslot.reserve((size_t)R * h * 4);
        RB_HIP(hipMemsetAsync(S->rtab.p, 0xFF, sizeof(Slot) << log2cap, s));
        hipLaunchKernelGGL(k_run_slots, dim3(blocks_for(R)), dim3(TPB), 0, s, fv, runs, R, S->rtab.as<Slot>(), log2cap, S->rslot.as<uint32_t>());
        RB_HIP(hipStreamSynchronize(s));
        RB_REQUIRE(total == O, "rb_shard_conflict_replay: runs announce %u ops, got %u", total, O);
        // run keys are (component label << 32) | arrival number and arrive in arrival order: a stable sort on the label bits is the sort on all 64
        const int label_end = 32 + (int)std::min(32u, S->gid_bits);
        sort_keys_u64(g->temp.p, g->temp.cap, S->rk0.as<uint64_t>(), S->rk1.as<uint64_t>(), R, 32, label_end, s);
        if (O) {
            hipLaunchKernelGGL(k_run_expand, dim3(blocks_for((int64_t)R * 64)), dim3(TPB), 0, s, runs, S->cnoff.as<uint32_t>(), (const uint32_t *)ops_dev, R,
                               S->ok0.as<uint64_t>(), S->ov0.as<uint32_t>());
            // op keys are (component label << 32) | occurrence id: the occurrence bits of this sub-batch, then the label bits
            sort_pairs_u64_u32_2r(g->temp.p, g->temp.cap, S->ok0.as<uint64_t>(), S->ok1.as<uint64_t>(), S->ov0.as<uint32_t>(), S->ov1.as<uint32_t>(), O,
                                  0, S->sub_reads_bits ? (int)std::min(32u, S->pos_bits + S->sub_reads_bits) : 32, 32, label_end, s);
            uint32_t *big_flag = S->rbig.as<uint32_t>(), *big_list = big_flag + R;
            g->devctr.reserve(DEVCTR_BYTES);
            uint32_t *ctr = g->devctr.as<uint32_t>();
            hipLaunchKernelGGL(k_replay_small, dim3(blocks_for(R)), dim3(TPB), 0, s, fv, S->rtab.as<Slot>(), S->rslot.as<uint32_t>(), S->rk1.as<uint64_t>(), R,
                               S->ok1.as<uint64_t>(), S->ov1.as<uint32_t>(), O, big_flag, g->small_ops);
            select_flagged(g->temp.p, g->temp.cap, big_flag, 1u, R, big_list, ctr + 4, s);
            hipLaunchKernelGGL(k_replay_big, dim3(std::min<uint32_t>(R, 65536u)), dim3(64), 0, s, fv, S->rtab.as<Slot>(), S->rslot.as<uint32_t>(), runs,
                               S->rk1.as<uint64_t>(), R, S->ok1.as<uint64_t>(), S->ov1.as<uint32_t>(), O, big_list, ctr + 4);
        }
        // final bytes of every touched counter go back to the counter's owner
        const size_t cap = (size_t)1 << log2cap;
        S->stage0.reserve(cap * 8); S->stage2.reserve(2 * cap + 32);
        uint8_t *w_val = S->stage2.as<uint8_t>(), *w_drop = w_val + cap;
        hipLaunchKernelGGL(k_tab_emit, dim3(blocks_for((int64_t)cap)), dim3(TPB), 0, s, S->rtab.as<Slot>(), cap, S->stage0.as<uint64_t>(), w_val, w_drop);
        RouteIdx fw{S->stage0.as<uint64_t>(), w_drop, (uint64_t)S->span[RB_CBF], nullptr, w_val, nullptr, nullptr, nullptr, nullptr};
        route_any_order(g, fw, cap, w_counts, [&](RouteIdx &ff, size_t kept) {
            ff.out_idx = (uint64_t *)slot_reserve(S, RB_SLOT_CW_IDX, kept * 8);
            ff.out8 = (uint8_t *)slot_reserve(S, RB_SLOT_CW_VAL, kept);
        });
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}

// ---- queries against the sharded filters (any rank may ask about any hash) ----
int rb_shard_query_make(rb_graph *g, int what, int which_bits, const uint64_t *h0_host, size_t n, int64_t *bit_counts, int64_t *ctr_counts) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && bit_counts && ctr_counts && (n == 0 || h0_host), "rb_shard_query_make: bad argument");
        RB_REQUIRE(what >= 0 && what <= 2, "rb_shard_query_make: what must be 0 (lookup), 1 (cbf count) or 2 (graph count)");
        RB_HIP(hipSetDevice(g->p.device));
        if (n) RB_HIP(hipMemcpyAsync(rb::shard_query_h0(g, n), h0_host, n * 8, hipMemcpyHostToDevice, g->stream));
        rb::shard_query_make_dev(g, what, which_bits, n, bit_counts, ctr_counts);
    });
}
int rb_shard_query_serve(rb_graph *g, int which_bits, const void *bidx_dev, int64_t nb, const void *cidx_dev, int64_t nc,
                         void *breply_dev, void *creply_dev) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && nb >= 0 && nc >= 0, "rb_shard_query_serve: bad argument");
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        if (nb) {
            BitFilter *bf = which_bits == RB_DBGBF ? &g->dbg : which_bits == RB_RPKBF ? &g->rpk : nullptr;
            RB_REQUIRE(bf && bf->bits, "rb_shard_query_serve: bit filter %d is not part of this sharded graph", which_bits);
            hipLaunchKernelGGL(k_query_bits, dim3(blocks_for(nb)), dim3(TPB), 0, s, bf->bits, (uint64_t)bf->lo, (const uint64_t *)bidx_dev, (size_t)nb, (uint8_t *)breply_dev);
        }
        if (nc) hipLaunchKernelGGL(k_query_ctrs, dim3(blocks_for(nc)), dim3(TPB), 0, s, g->cbf, (uint64_t)g->cbf_lo, (const uint64_t *)cidx_dev, (size_t)nc, (uint8_t *)creply_dev);
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
    });
}
int rb_shard_query_finish(rb_graph *g, int which_bits, const void *breply_dev, const void *creply_dev, uint8_t *out8_host, float *outf_host) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard, "rb_shard_query_finish: bad argument");
        ShardState *S = g->shard;
        const size_t n = S->q_n;
        if (!n) return;
        const int what = S->q_what;
        RB_REQUIRE(what == 0 ? out8_host != nullptr : outf_host != nullptr, "rb_shard_query_finish: missing output array");
        const void *res = rb::shard_query_combine_dev(g, which_bits, breply_dev, creply_dev);
        hipStream_t s = g->stream;
        if (what == 0) RB_HIP(hipMemcpyAsync(out8_host, res, n, hipMemcpyDeviceToHost, s));
        else RB_HIP(hipMemcpyAsync(outf_host, res, n * 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
    });
}

}  // extern "C"

namespace rb {
// The query protocol with the hashes already on the device (rb_shard_query_make uploads them; the traversal kernels of a sharded
// graph write them there themselves, rb_shard_trav_*): indices, bucketed by owner, into slots Q_BIDX / Q_CIDX.
uint64_t *shard_query_h0(rb_graph *g, size_t n) {
    ShardState *S = g->shard;
    S->q_h0.reserve(std::max<size_t>(n, 1) * 8);
    return S->q_h0.as<uint64_t>();
}
void shard_query_make_dev(rb_graph *g, int what, int which_bits, size_t n, int64_t *bit_counts, int64_t *ctr_counts) {
    ShardState *S = g->shard;
    RB_HIP(hipSetDevice(g->p.device));
    hipStream_t s = g->stream;
    BitFilter *bf = which_bits == RB_DBGBF ? &g->dbg : which_bits == RB_RPKBF ? &g->rpk : nullptr;
    if (what != 1) RB_REQUIRE(bf && bf->bits, "rb_shard_query_make: bit filter %d is not part of this sharded graph", which_bits);
    for (int r = 0; r < S->G; ++r) bit_counts[r] = ctr_counts[r] = 0;
    S->slot_bytes[RB_SLOT_Q_BIDX] = S->slot_bytes[RB_SLOT_Q_CIDX] = 0;
    S->q_n = n; S->q_what = what;
    if (!n) return;
    const int bh = what != 1 ? bf->num_hash : 0, ch = what != 0 ? g->cbf_h : 0;
    S->stage0.reserve(n * bh * 8 + 16); S->stage3.reserve(n * ch * 8 + 16);
    S->q_bpos.reserve(n * bh * 4 + 16); S->q_cpos.reserve(n * ch * 4 + 16);
    hipLaunchKernelGGL(k_query_idx, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, s, kmul_of(g->k), bf ? bf->mod : g->cbf_mod, bh, g->cbf_mod, ch,
                       S->q_h0.as<uint64_t>(), n, S->stage0.as<uint64_t>(), S->stage3.as<uint64_t>());
    if (bh) {
        RouteIdx fb{S->stage0.as<uint64_t>(), nullptr, (uint64_t)S->span[which_bits], nullptr, nullptr, nullptr, nullptr, nullptr, S->q_bpos.as<uint32_t>()};
        route(g, fb, n * bh, bit_counts, [&](RouteIdx &ff, size_t kept) { ff.out_idx = (uint64_t *)slot_reserve(S, RB_SLOT_Q_BIDX, kept * 8); });
    }
    if (ch) {
        RouteIdx fc{S->stage3.as<uint64_t>(), nullptr, (uint64_t)S->span[RB_CBF], nullptr, nullptr, nullptr, nullptr, nullptr, S->q_cpos.as<uint32_t>()};
        route(g, fc, n * ch, ctr_counts, [&](RouteIdx &ff, size_t kept) { ff.out_idx = (uint64_t *)slot_reserve(S, RB_SLOT_Q_CIDX, kept * 8); });
    }
    RB_HIP(hipGetLastError());
    RB_HIP(hipStreamSynchronize(s));
}
// replies of the owners -> one answer per queried hash, left on the device (u8 for what == 0, else float); enqueued on g->stream
const void *shard_query_combine_dev(rb_graph *g, int which_bits, const void *breply_dev, const void *creply_dev) {
    ShardState *S = g->shard;
    const size_t n = S->q_n;
    const int what = S->q_what;
    RB_HIP(hipSetDevice(g->p.device));
    BitFilter *bf = which_bits == RB_DBGBF ? &g->dbg : which_bits == RB_RPKBF ? &g->rpk : nullptr;
    const int bh = what != 1 ? bf->num_hash : 0, ch = what != 0 ? g->cbf_h : 0;
    S->q_out.reserve(n * 4 + 16);
    if (n) hipLaunchKernelGGL(k_query_combine, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, g->stream, what, bh, ch, S->q_bpos.as<uint32_t>(), (const uint8_t *)breply_dev,
                              S->q_cpos.as<uint32_t>(), (const uint8_t *)creply_dev, n, S->q_out.as<uint8_t>(), S->q_out.as<float>());
    RB_HIP(hipGetLastError());
    return S->q_out.p;
}
}  // namespace rb

namespace rb {
void shard_clear_pairs_acc(rb_graph *g) {       // rb_graph_clear of the pair filter: bits waiting in the accumulation copy go too
    ShardState *S = g->shard;
    if (!S || !S->rpk_acc.bits) return;
    RB_HIP(hipMemsetAsync(S->rpk_acc.bits, 0, S->rpk_acc.alloc, g->stream));
    seen_reset(S->rpk_acc, g->stream);
    S->acc_dirty = false;
}
void shard_free(rb_graph *g) {
    ShardState *S = g->shard;
    if (!S) return;
    for (auto &b : S->slot) b.release();
    DevBuf *bufs[] = {&S->dreq_pos, &S->creq_pos, &S->creq_dup, &S->cfinal, &S->conf_list, &S->oinfo, &S->ord_pos, &S->conf2, &S->deferred, &S->heavy2, &S->stage0, &S->stage1, &S->stage2, &S->stage3,
                      &S->rhist, &S->roffs, &S->bounds, &S->rcnt, &S->own_f, &S->own_cs, &S->esz, &S->eoff, &S->etab, &S->eslot, &S->elabel, &S->cdesc,
                      &S->cpos, &S->cnops, &S->cnoff, &S->rk0, &S->rk1, &S->ok0, &S->ok1, &S->ov0, &S->ov1, &S->rtab, &S->rslot, &S->rbig, &S->q_h0, &S->q_bpos, &S->q_cpos, &S->q_out, &S->cache_upd, &S->lmask, &S->lcoll, &S->lcv, &S->lctr};
    for (auto *b : bufs) b->release();
    free_bits(S->rpk_acc);
    if (S->pinned) (void)hipHostFree(S->pinned);
    delete S;
    g->shard = nullptr;
}
}  // namespace rb

namespace rb {
bool shard_is_split(const rb_graph *g) { return g && g->shard && g->shard->replicate_cache; }
}  // namespace rb
