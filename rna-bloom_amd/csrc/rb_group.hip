// rb_group.hip — grouping of the (h0, occurrence) records of a sub-batch, written for gfx950.
//
// What the insert pipeline needs from this stage is not a sorted array but
// GROUPS: equal base hashes next to each other, the occurrences of one hash in sequential order
// (BloomFilterDeBruijnGraph.add is applied per k-mer occurrence in input order,
// R/graph/BloomFilterDeBruijnGraph.java:405-412), the draw strength of every occurrence, and the list of
// runs (hash, count, start).  A hash may appear as several runs ("split runs"), so
// the stage may group on as few hash bits as it likes.
//
// That freedom is what this file uses instead of a full 4-pass radix sort:
//   1. MSD partition of the records on T = ceil(log2(N / 3072)) hash bits into 2^T "fine buckets" of ~2-3 K
//      records, in one or two STABLE passes over HBM (k_part_count -> exclusive scan -> k_part_scatter; a tile
//      of TPB x ITEMS records is ranked through LDS match masks and staged through LDS so that every bucket
//      receives one contiguous piece per tile; consecutive tiles are given to the same XCD so that the pieces
//      of neighbouring tiles meet in one L2).  The second pass is segmented: its tiles never straddle a
//      first-pass bucket, so its scanned histogram IS the table of fine-bucket bounds.
//   2. k_group_buckets: one workgroup per fine bucket sorts it in LDS on the next 16 hash bits (two stable
//      8-bit counting passes), computes the strengths, finds the run heads and writes (hash, count, start)
//      for its runs into slots taken with one atomicAdd per bucket (nothing depends on the order of the runs;
//      RB_GROUP_ORDERED=1: bucket order, through a chained scan over the buckets — ticket + look-back).
// Records are read 3 times and written 2.x times (12-byte records) instead of 5 + 4 times, and the separate
// strength and run-length passes are gone.
#include <stdlib.h>

#include <algorithm>

#include "rb_pipeline.hpp"

namespace rb {

// ---- geometry --------------------------------------------------------------------------------------
constexpr uint32_t GR_KEY_TOP = 60;          // grouping uses hash bits below this one (the top bits of a canonical
                                             // hash are skewed: it is a signed minimum of two hashes)
constexpr uint32_t GR_MAX_GROUP_BITS = 36;
constexpr uint32_t GR_TILE = 4096;           // records per partition tile and LDS capacity of the bucket kernel
constexpr uint32_t GR_BUCKET_TARGET = 3072;  // T is chosen so that the average fine bucket is in (TARGET/2, TARGET]
constexpr uint32_t GR_LOCAL_BITS = 8;        // per local pass
constexpr uint32_t GR_PART_MAX_BITS = 10;    // per partition pass

// A record the emit pass cancelled after its slot was assigned (rb_batch.hip: the occurrence turned out to be a no-op against the
// stores of the sub-batch before): key = all ones AND occurrence = all ones.  The first partition pass leaves such records
// out (GR_FLAG_DEAD); every later stage sees a dense array of the live ones.  (An all-ones key alone proves nothing — it is a
// possible hash — so the occurrence id is checked too; add_range never hands out the all-ones occurrence id.)
constexpr uint64_t GR_DEAD_KEY = ~0ull;
constexpr uint32_t GR_DEAD_VAL = ~0u;
__device__ __forceinline__ uint32_t gr_digit(uint64_t key, uint32_t shift, uint32_t bits) {
    return (uint32_t)(key >> shift) & ((1u << bits) - 1u);
}
// partition digits from the first filter index (GrIdx): d = floor((idx_0 - lo) * 2^T / span) — one mulhi with mul = floor(2^(64+T) / span) —
// is the number of the k-mer's FINE bucket, the T = t_hi + t_lo partition bits taken as 2^T equal index ranges; the first pass takes its
// top t_hi bits (shift = t_lo), the second pass the rest (shift = 0)
struct GrIdxDev { Mod mod; uint64_t lo, mul; uint32_t top, shift; };          // mul == 0: off (digits come from the hash bits); top = 2^T - 1
__device__ __forceinline__ uint32_t gr_idx_digit(uint64_t key, const GrIdxDev &ix, uint32_t bits) {
    const uint64_t i = index_of(key, ix.mod) - ix.lo;    // (a key outside [lo, lo + span) — none exist on a shard — lands in the last bucket)
    const uint32_t d = min((uint32_t)min(__umul64hi(i, ix.mul), (uint64_t)0xFFFFFFFFull), ix.top);
    return (d >> ix.shift) & ((1u << bits) - 1u);
}
__device__ __forceinline__ uint32_t gr_lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Stable ranks inside one wavefront's ITEMS rows of 64 records: rank[r] = number of earlier records of this
// wavefront (rows before r, lanes below in row r) with the same digit.  wc = this wavefront's counters
// (zeroed, 2^bits entries); on return wc[d] = number of records of digit d in the wavefront.
// dig[r] == ~0u marks an absent record; rows from `rows` on are absent altogether (wavefront-uniform).
template <int ITEMS, typename CT>
__device__ __forceinline__ void gr_wave_rank(const uint32_t (&dig)[ITEMS], uint32_t (&rank)[ITEMS], CT *wc, uint32_t bits, uint32_t rows) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        if ((uint32_t)r < rows) {
            const bool valid = dig[r] != ~0u;
            uint64_t m = __ballot(valid);
            for (uint32_t b = 0; b < bits; ++b) {
                const bool bit = (dig[r] >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            uint32_t prior = 0;
            if (valid) prior = (uint32_t)wc[dig[r]];
            rank[r] = prior + gr_lanes_below(m);
            if (valid && (m >> lane) == 1ull) wc[dig[r]] = (CT)(prior + (uint32_t)__popcll(m));   // highest lane of the group
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// The same ranks with the match masks taken from LDS instead of `bits` ballots per row: every lane ORs its lane bit
// into wm[digit] (ds_or_b64; the DS operations of a wavefront execute in issue order), reads the word back — that IS
// the mask of the lanes of this row with the same digit — and the highest lane of each group clears it again.
// wm = this wavefront's mask table (2^bits words, all zero on entry and on return).  ~6 DS operations per row
// instead of ~12 VALU/SALU instructions per digit bit: the bucket kernel below is bound by instruction issue.
template <int ITEMS, typename CT>
__device__ __forceinline__ void gr_wave_rank_lds(const uint32_t (&dig)[ITEMS], uint32_t (&rank)[ITEMS], CT *wc, unsigned long long *wm, uint32_t rows) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        if ((uint32_t)r < rows) {
            const bool valid = dig[r] != ~0u;
            uint64_t m = 0;
            uint32_t prior = 0;
            if (valid) {
                __hip_atomic_fetch_or(&wm[dig[r]], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                m = __hip_atomic_load(&wm[dig[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                prior = (uint32_t)wc[dig[r]];
            }
            rank[r] = prior + gr_lanes_below(m);
            if (valid && (m >> lane) == 1ull) {                                   // highest lane of the group
                __hip_atomic_store(&wm[dig[r]], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                wc[dig[r]] = (CT)(prior + (uint32_t)__popcll(m));
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Digits of 9 or 10 bits with a 256-word mask table: the LDS mask gives the lanes that agree in the low 8 bits, one ballot per
// remaining bit narrows them down to the lanes with the same digit (a 1024-word table per wavefront would be 64 KB of LDS).
// The highest lane of the LOW-bits group clears the word; the highest lane of the digit's group keeps the counter.
template <int ITEMS, typename CT>
__device__ __forceinline__ void gr_wave_rank_lds_wide(const uint32_t (&dig)[ITEMS], uint32_t (&rank)[ITEMS], CT *wc, unsigned long long *wm, uint32_t bits,
                                                      uint32_t rows) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        if ((uint32_t)r < rows) {
            const bool valid = dig[r] != ~0u;
            uint64_t ml = 0;
            uint32_t prior = 0;
            if (valid) {
                __hip_atomic_fetch_or(&wm[dig[r] & 255u], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ml = __hip_atomic_load(&wm[dig[r] & 255u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                prior = (uint32_t)wc[dig[r]];
            }
            uint64_t m = ml;
            for (uint32_t b = 8; b < bits; ++b) {
                const bool bit = valid && ((dig[r] >> b) & 1u);
                const uint64_t bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            rank[r] = prior + gr_lanes_below(m);
            if (valid && (ml >> lane) == 1ull) __hip_atomic_store(&wm[dig[r] & 255u], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (valid && (m >> lane) == 1ull) wc[dig[r]] = (CT)(prior + (uint32_t)__popcll(m));
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// exclusive scan of one value per thread over the block
template <int TPB>
__device__ __forceinline__ uint32_t gr_block_excl_scan(uint32_t v, uint32_t *s_wsum) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += t;
    }
    if (lane == 63u) s_wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < TPB / 64; ++i) {
        const uint32_t t = s_wsum[i];
        if ((uint32_t)i < w) base += t;
    }
    __syncthreads();
    return base + inc - v;
}

// per-digit wavefront counters -> exclusive prefix over the wavefronts (in place) and, in dstart[], the exclusive
// prefix over the digits of the per-digit totals.  Called by all threads, between barriers of the caller.
template <int TPB, typename CT, typename DT>
__device__ __forceinline__ void gr_digit_offsets(CT *wcnt /* [TPB/64][nb] */, DT *dstart /* [nb] */, uint32_t nb, uint32_t *s_wsum) {
    constexpr int NW = TPB / 64;
    const uint32_t per = (nb + TPB - 1) / TPB;       // nb <= 1024, TPB >= 256: at most 4 consecutive digits per thread
    uint32_t tot[4] = {0, 0, 0, 0}, sum = 0;
    for (uint32_t q = 0; q < per; ++q) {
        const uint32_t d = threadIdx.x * per + q;
        if (d < nb) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const uint32_t t = (uint32_t)wcnt[(uint32_t)w * nb + d]; wcnt[(uint32_t)w * nb + d] = (CT)run; run += t; }
            tot[q] = run; sum += run;
        }
    }
    uint32_t base = gr_block_excl_scan<TPB>(sum, s_wsum);
    for (uint32_t q = 0; q < per; ++q) {
        const uint32_t d = threadIdx.x * per + q;
        if (d < nb) { dstart[d] = (DT)base; base += tot[q]; }
    }
}

// blocks b, b+8, b+16, ... run on the same XCD (observed, not promised: used for speed only): give every XCD a
// contiguous range of tiles, so that the bucket pieces of consecutive tiles are written through the same L2
__device__ __forceinline__ uint32_t gr_tile_of_block(uint32_t b, uint32_t grid_tiles, int xcd_map) {
    if (!xcd_map) return b;
    return (b & 7u) * (grid_tiles >> 3) + (b >> 3);
}
inline uint32_t gr_grid_for_tiles(uint32_t ntiles) { return ((ntiles + 7u) / 8u) * 8u; }

// A partition pass works on tiles.  First pass: tile t = records [t*TILE, ...), histogram entry (d, t) at d*ntiles + t.
// Second pass: the tiles of first-pass bucket b are tile_base[b] .. tile_base[b+1]-1 (GrTile descriptors written by
// k_seg_tiles), histogram of bucket b = [d][local tile] at tile_base[b]*nb, so that ONE exclusive scan over all
// entries yields final positions, bucket after bucket, digit after digit.
struct GrTile { uint32_t start, count, hist_base, hist_stride; };
struct GrTiling {
    const GrTile *desc;          // nullptr: uniform tiling of [0, n)
    const uint32_t *ntiles_dev;  // with desc: number of tiles actually in use (device)
    uint32_t n, ntiles, grid_tiles;
    int xcd_map;
};
__device__ __forceinline__ bool gr_get_tile(const GrTiling &tl, GrTile &t) {
    const uint32_t i = gr_tile_of_block(blockIdx.x, tl.grid_tiles, tl.xcd_map);
    if (!tl.desc) {
        if (i >= tl.ntiles) return false;
        t.start = i * GR_TILE; t.count = min(GR_TILE, tl.n - t.start); t.hist_base = i; t.hist_stride = tl.ntiles;
        return true;
    }
    if (i >= *tl.ntiles_dev) return false;
    t = tl.desc[i];
    return true;
}

// ---- partition pass: histogram per (digit, tile) -----------------------------------------------------
// DERIVE (the swept stage's first binning pass): the records are not in memory yet — key = the second hash of the run's hash keys[j], value = j
template <int TPB, bool DERIVE = false>
__global__ void __launch_bounds__(TPB) k_part_count(const uint64_t *__restrict__ keys, GrTiling tl, uint32_t shift, uint32_t bits,
                                                    uint32_t *__restrict__ hist, const uint32_t *__restrict__ dead_vals = nullptr,
                                                    GrIdxDev ix = GrIdxDev{Mod{1, 0, 0}, 0, 0, 0, 0}, uint64_t kmul = 0) {
    constexpr int ITEMS = GR_TILE / TPB;
    __shared__ uint32_t s_h[1u << GR_PART_MAX_BITS];
    GrTile t;
    if (!gr_get_tile(tl, t)) return;
    const uint32_t nb = 1u << bits;
    for (uint32_t d = threadIdx.x; d < nb; d += TPB) s_h[d] = 0;
    __syncthreads();
    uint64_t k[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t j = (uint32_t)i * TPB + threadIdx.x;
        k[i] = j < t.count ? keys[t.start + j] : 0ull;
        if (DERIVE) k[i] = multi_hash(k[i], 1u, kmul);
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t j = (uint32_t)i * TPB + threadIdx.x;
        if (j < t.count && !(dead_vals && k[i] == GR_DEAD_KEY && dead_vals[t.start + j] == GR_DEAD_VAL))
            atomicAdd(&s_h[ix.mul ? gr_idx_digit(k[i], ix, bits) : gr_digit(k[i], shift, bits)], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nb; d += TPB) hist[(size_t)t.hist_base + (size_t)d * t.hist_stride] = s_h[d];
}

// ---- partition pass: stable scatter ------------------------------------------------------------------
template <int TPB, int MAXBITS, bool DERIVE = false>
__global__ void __launch_bounds__(TPB) k_part_scatter(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, GrTiling tl,
                                                      uint32_t shift, uint32_t bits, const uint32_t *__restrict__ goffs /* exclusive scan of hist */,
                                                      uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t wide_lds,
                                                      uint32_t skip_dead = 0u, GrIdxDev ix = GrIdxDev{Mod{1, 0, 0}, 0, 0, 0, 0}, uint64_t kmul = 0) {
    constexpr uint32_t ITEMS = GR_TILE / TPB, NW = TPB / 64, SEG = 64 * ITEMS, MAXNB = 1u << MAXBITS;
    __shared__ uint64_t s_keys[GR_TILE];
    __shared__ uint32_t s_vals[GR_TILE];
    __shared__ uint32_t s_cnt[NW * MAXNB / 2 > MAXNB ? NW * MAXNB / 2 : MAXNB];   // u16 per (wavefront, digit); later: u32 global bases per digit
    __shared__ uint16_t s_dstart[MAXNB];
    __shared__ uint32_t s_wsum[NW];
    // match masks from LDS (gr_wave_rank_lds: the pass is bound by instruction issue, not by HBM): a table of 2^bits words per
    // wavefront for digits of up to 8 bits; wider digits use 256 words per wavefront plus a ballot per extra bit.  The table
    // lives in s_vals — free until the ranked records are staged there, a barrier later — which keeps the 8-bit kernel at
    // 53 KB of LDS: three workgroups per CU instead of two.
    constexpr bool NARROW = MAXBITS <= 8;
    static_assert(sizeof(s_vals) >= NW * 256u * sizeof(unsigned long long), "the mask table must fit the staging array");
    unsigned long long *s_wmask = reinterpret_cast<unsigned long long *>(s_vals);
    uint16_t *s_wcnt = reinterpret_cast<uint16_t *>(s_cnt);
    GrTile t;
    if (!gr_get_tile(tl, t)) return;
    const uint32_t nb = 1u << bits;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t d = threadIdx.x; d < NW * nb; d += TPB) s_wcnt[d] = 0;
    for (uint32_t d = threadIdx.x; d < NW * (NARROW ? nb : 256u); d += TPB) s_wmask[d] = 0ull;
    // rows of this wavefront: records w*SEG + r*64 + lane of the tile
    uint64_t k[ITEMS];
    uint32_t v[ITEMS], dig[ITEMS], rank[ITEMS];
#pragma unroll
    for (uint32_t r = 0; r < ITEMS; ++r) {
        const uint32_t j = w * SEG + r * 64u + lane;
        bool ok = j < t.count;
        k[r] = ok ? keys_in[t.start + j] : 0ull;
        v[r] = (ok && vals_in) ? vals_in[t.start + j] : 0u;
        if (DERIVE) { k[r] = multi_hash(k[r], 1u, kmul); v[r] = t.start + j; }
        if (skip_dead && k[r] == GR_DEAD_KEY && v[r] == GR_DEAD_VAL) ok = false;          // cancelled by the emit pass: not scattered
        dig[r] = ok ? (ix.mul ? gr_idx_digit(k[r], ix, bits) : gr_digit(k[r], shift, bits)) : ~0u;
    }
    const uint32_t rows = t.count > w * SEG ? min(ITEMS, (t.count - w * SEG + 63u) / 64u) : 0u;
    __syncthreads();
    if (NARROW) gr_wave_rank_lds<ITEMS>(dig, rank, s_wcnt + w * nb, s_wmask + w * nb, rows);
    else if (wide_lds) gr_wave_rank_lds_wide<ITEMS>(dig, rank, s_wcnt + w * nb, s_wmask + w * 256u, bits, rows);
    else gr_wave_rank<ITEMS>(dig, rank, s_wcnt + w * nb, bits, rows);
    __syncthreads();
    gr_digit_offsets<TPB>(s_wcnt, s_dstart, nb, s_wsum);
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < ITEMS; ++r)
        if (dig[r] != ~0u) {
            const uint32_t p = (uint32_t)s_dstart[dig[r]] + (uint32_t)s_wcnt[w * nb + dig[r]] + rank[r];
            s_keys[p] = k[r]; s_vals[p] = v[r];
        }
    __syncthreads();
    uint32_t live = 0;                               // records staged = sum of the digit totals (gr_digit_offsets left the per-wavefront sums in s_wsum)
#pragma unroll
    for (uint32_t i = 0; i < NW; ++i) live += s_wsum[i];
    for (uint32_t d = threadIdx.x; d < nb; d += TPB) s_cnt[d] = goffs[(size_t)t.hist_base + (size_t)d * t.hist_stride] - (uint32_t)s_dstart[d];
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < live; j += TPB) {
        const uint64_t key = s_keys[j];
        const uint32_t g = s_cnt[ix.mul ? gr_idx_digit(key, ix, bits) : gr_digit(key, shift, bits)] + j;
        keys_out[g] = key;
        if (vals_out) vals_out[g] = s_vals[j];              // (uniform: key-only sorts pass no values)
    }
}

// first-pass bucket bounds -> tiles of the second pass.  One block; nb1 <= 1024 segments.
__global__ void __launch_bounds__(1024) k_seg_tiles(const uint32_t *__restrict__ goffs1, uint32_t ntiles1, uint32_t nb1, uint32_t n, uint32_t nb2,
                                                    GrTile *__restrict__ desc, uint32_t *__restrict__ seg_tile_base /* [nb1 + 1] */,
                                                    uint32_t *__restrict__ seg_start /* [nb1 + 1] */, uint32_t *__restrict__ ntiles2_dev,
                                                    const uint32_t *__restrict__ n_dev = nullptr /* live records after a pass that skipped dead ones */) {
    __shared__ uint32_t s_wsum[16];
    const uint32_t b = threadIdx.x;
    if (n_dev) n = *n_dev;
    uint32_t s = 0, e = 0;
    if (b < nb1) { s = goffs1[(size_t)b * ntiles1]; e = b + 1u < nb1 ? goffs1[(size_t)(b + 1u) * ntiles1] : n; }
    const uint32_t nt = (e - s + GR_TILE - 1u) / GR_TILE;
    const uint32_t tb = gr_block_excl_scan<1024>(nt, s_wsum);
    if (b < nb1) {
        seg_tile_base[b] = tb; seg_start[b] = s;
        for (uint32_t i = 0; i < nt; ++i) {
            GrTile t; t.start = s + i * GR_TILE; t.count = min(GR_TILE, e - t.start); t.hist_base = tb * nb2 + i; t.hist_stride = nt;
            desc[tb + i] = t;
        }
        if (b == nb1 - 1u) { seg_tile_base[nb1] = tb + nt; seg_start[nb1] = n; *ntiles2_dev = tb + nt; }
    }
}
// fine-bucket bounds bstart[2^T + 1]
__global__ void k_bucket_bounds(const uint32_t *__restrict__ goffs, uint32_t ntiles1, const uint32_t *__restrict__ seg_tile_base,
                                const uint32_t *__restrict__ seg_start, uint32_t t_hi, uint32_t t_lo, uint32_t n, uint32_t *__restrict__ bstart,
                                const uint32_t *__restrict__ n_dev = nullptr) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, nbk = 1u << (t_hi + t_lo);
    if (n_dev) n = *n_dev;
    if (p > nbk) return;
    if (p == nbk) { bstart[p] = n; return; }
    if (t_hi == 0) { bstart[p] = 0; return; }                                // no partition pass: one bucket
    if (t_lo == 0) { bstart[p] = goffs[(size_t)p * ntiles1]; return; }      // single pass: goffs = first-pass offsets
    const uint32_t b = p >> t_lo, lo = p & ((1u << t_lo) - 1u);
    const uint32_t tb = seg_tile_base[b], nt = seg_tile_base[b + 1u] - tb;
    bstart[p] = nt ? goffs[((size_t)tb << t_lo) + (size_t)lo * nt] : seg_start[b];
}

__global__ void k_copy_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = *src; }

// ---- bucket kernel: LDS sort on 2 x 8 bits, strengths, runs --------------------------------------------
// GR_ABL (compile-time, tools/microbench/group_bench.hip only): parts of the bucket kernel compiled out to price them —
// 1 look-back (ordered mode), 2 repair, 4 second counting pass, 8 output stores, 16 both counting passes.  Results are wrong then.
#ifndef GR_ABL
#define GR_ABL 0
#endif
struct GroupRng { uint64_t seed, ordinal0; uint32_t pos_bits; };
constexpr unsigned long long GR_ST_AGG = 1ull << 62, GR_ST_PREFIX = 2ull << 62;

// wavefront-wide look-back over the status words of the buckets before c: sum of their run counts
__device__ __forceinline__ uint32_t gr_look_back(const unsigned long long *status, uint32_t c) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t sum = 0;
    int64_t p = (int64_t)c - 1;
    for (;;) {
        const int64_t mine = p - (int64_t)lane;
        unsigned long long st = GR_ST_PREFIX;                        // below bucket 0: an empty prefix
        if (mine >= 0) st = __hip_atomic_load(&status[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t ready = __ballot((st >> 62) != 0ull), pref = __ballot((st >> 62) == 2ull);
        const uint32_t n_ready = ready == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~ready);   // lanes 0 .. n_ready-1 are published
        const uint32_t first_pref = pref ? (uint32_t)__builtin_ctzll(pref) : 64u;
        const uint32_t take = first_pref < n_ready ? first_pref + 1u : n_ready;
        uint32_t val = lane < take ? (uint32_t)(st & 0xFFFFFFFFull) : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o, 64);
        sum += val;
        if (first_pref < n_ready) return sum;
        p -= (int64_t)take;
        if (take == 0) __builtin_amdgcn_s_sleep(2);
    }
}

// (two workgroups of 512 threads per CU = 4 wavefronts per SIMD: at most 128 VGPRs — the second launch bound keeps the compiler there; a
// few registers more and only ONE workgroup fits a CU: 25 -> 39 ms, measured when the run ordering below first went in)
template <int TPB, bool PREFETCH>
__global__ void __launch_bounds__(TPB, (TPB == 512 ? 4 : 2)) k_group_buckets(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                       const uint32_t *__restrict__ bstart, uint32_t nbuckets,
                                                       uint32_t shift_lo, uint32_t bits_lo, uint32_t shift_hi, uint32_t bits_hi,
                                                       GroupRng rng, uint32_t do_fix, uint32_t *__restrict__ ticket, unsigned long long *__restrict__ status,
                                                       uint32_t *__restrict__ big_list, uint32_t *__restrict__ n_big,
                                                       uint32_t *__restrict__ vals_out, uint8_t *__restrict__ tz_out,
                                                       uint64_t *__restrict__ uniq, uint32_t *__restrict__ counts, uint32_t *__restrict__ starts,
                                                       uint32_t *__restrict__ n_runs_out, uint32_t by_class,
                                                       uint32_t *__restrict__ brun /* null, or per bucket: first run slot, */, uint32_t *__restrict__ bnr /* number of runs */) {
    constexpr uint32_t ITEMS = GR_TILE / TPB, NW = TPB / 64, SEG = 64 * ITEMS, NB = 1u << GR_LOCAL_BITS;
    static_assert(ITEMS * NW == 64, "the segment scan below is one wavefront wide");
    __shared__ uint64_t s_keys[GR_TILE];
    __shared__ uint32_t s_vals[GR_TILE];          // later: head positions (u16)
    __shared__ uint16_t s_wcnt[2][NW * NB];       // one table per pass: each is cleaned right after its use, two barriers before the next
    __shared__ unsigned long long s_wmask[NW * NB];
    __shared__ uint16_t s_dstart[NB];
    __shared__ uint32_t s_wsum[NW], s_seg[ITEMS * NW], s_misc[3], s_fixn, s_redo;
    __shared__ uint16_t s_fixlist[64];
    __shared__ uint32_t s_cls[16];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t d = threadIdx.x; d < NW * NB; d += TPB) { s_wmask[d] = 0ull; s_wcnt[0][d] = 0; s_wcnt[1][d] = 0; }
    uint32_t c = 0, b0 = 0, b1 = 0;
    uint64_t k[ITEMS];
    uint32_t v[ITEMS], dig[ITEMS], rank[ITEMS];
    auto load_bucket = [&](uint32_t cc, uint32_t &bb0, uint32_t &bb1, uint64_t (&kk)[ITEMS], uint32_t (&vv)[ITEMS]) {
        bb0 = bstart[cc]; bb1 = bstart[cc + 1u];
        const uint32_t n = bb1 - bb0 > GR_TILE ? 0u : bb1 - bb0;
#pragma unroll
        for (uint32_t r = 0; r < ITEMS; ++r) {
            const uint32_t j = w * SEG + r * 64u + lane;
            const bool ok = j < n;
            kk[r] = ok ? keys_in[bb0 + j] : 0ull;
            vv[r] = ok ? vals_in[bb0 + j] : 0u;
        }
    };
    // persistent workgroups: buckets are taken in ticket order.  PREFETCH (only without the chained scan, which must never
    // wait for a bucket whose workgroup is still busy with another one): the ticket of the NEXT bucket is taken when this one
    // starts and its records are loaded (into the registers this bucket's records came in) once this one is grouped in LDS,
    // so the ticket -> bounds -> records chain of global latencies runs behind the run extraction and the output of this bucket.
    if (PREFETCH) {
        if (threadIdx.x == 0) s_misc[0] = atomicAdd(ticket, 1u);
        __syncthreads();
        c = s_misc[0];
        if (c >= nbuckets) return;
        load_bucket(c, b0, b1, k, v);
    }
    for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) { s_misc[PREFETCH ? 2 : 0] = atomicAdd(ticket, 1u); s_fixn = 0; s_redo = 0; }
    __syncthreads();
    if (!PREFETCH) {
        c = s_misc[0];
        if (c >= nbuckets) return;
        load_bucket(c, b0, b1, k, v);
    }
    const uint32_t c_next = PREFETCH ? s_misc[2] : 0u;
    const bool big = b1 - b0 > GR_TILE;              // does not fit LDS: left to k_group_big (its runs are appended after all of these)
    const uint32_t cn = big ? 0u : b1 - b0;
    const uint32_t rows = cn > w * SEG ? min(ITEMS, (cn - w * SEG + 63u) / 64u) : 0u;
    if (big && threadIdx.x == 0) big_list[atomicAdd(n_big, 1u)] = c;
    bool in_lds = false;
    // one stable counting pass over the bucket (from the registers the records were loaded into, later from LDS)
    auto sort_pass = [&](uint32_t shift, uint32_t bits, uint16_t *wcnt) {
        if (bits == 0 || cn == 0) return;
        const uint32_t nb = 1u << bits;
        if (in_lds) {
#pragma unroll
            for (uint32_t r = 0; r < ITEMS; ++r) {
                const uint32_t j = w * SEG + r * 64u + lane;
                if (j < cn) { k[r] = s_keys[j]; v[r] = s_vals[j]; }
            }
        }
#pragma unroll
        for (uint32_t r = 0; r < ITEMS; ++r) {
            const uint32_t j = w * SEG + r * 64u + lane;
            dig[r] = j < cn ? gr_digit(k[r], shift, bits) : ~0u;
        }
        gr_wave_rank_lds<ITEMS>(dig, rank, wcnt + w * nb, s_wmask + w * NB, rows);
        __syncthreads();
        gr_digit_offsets<TPB>(wcnt, s_dstart, nb, s_wsum);
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < ITEMS; ++r)
            if (dig[r] != ~0u) {
                const uint32_t p = (uint32_t)s_dstart[dig[r]] + (uint32_t)wcnt[w * nb + dig[r]] + rank[r];
                s_keys[p] = k[r]; s_vals[p] = v[r];
            }
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < NW * nb; d += TPB) wcnt[d] = 0;
        in_lds = true;
    };
    // Two different hashes that agree in every sorted bit (bucket bits + local digits) sit interleaved in occurrence order and
    // would come out as several runs each.  Such groups are few (hashes^2 / 2^17 per bucket): every change of hash inside a
    // group is listed, and the wavefronts take the listed positions in turn — 64 lanes look at the 64 records around the
    // position, and if it is the FIRST change of its group and the group has at most 32 records, every record of the group
    // computes its rank under (hash, position) from the others' hashes (readlane loop) and moves there.  A longer group or a
    // full list is reported; with do_fix = 2 (sharded engine: 4x longer sub-batches, longer runs, costlier conflicts) the
    // bucket is then sorted again on 8 more bits (three passes), which leaves only short groups.  On one GPU that costs more
    // than the conflicts it saves (group_buckets 39 -> 56 ms for 57 M -> 54 M conflicting ops), so long groups stay split there.
    auto fix_groups = [&](uint32_t gshift, uint32_t gbits) {
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; ++i) {
            const uint32_t j = i * TPB + threadIdx.x;
            if (j == 0u || j >= cn) continue;
            const uint64_t k1 = s_keys[j], k0 = s_keys[j - 1u];
            if (k1 != k0 && gr_digit(k0, gshift, gbits) == gr_digit(k1, gshift, gbits)) {
                const uint32_t slot = atomicAdd(&s_fixn, 1u);
                if (slot < 64u) s_fixlist[slot] = (uint16_t)j;
            }
        }
        __syncthreads();
        const uint32_t nfix = min(s_fixn, 64u);
        if (s_fixn > 64u && threadIdx.x == 0) s_redo = 1u;
        for (uint32_t m0 = 0; m0 < nfix; m0 += NW) {             // NW positions per round: all look, then all move
            const uint32_t m = m0 + w;
            uint64_t key = 0;
            uint32_t val = 0, dst = ~0u;
            if (m < nfix) {
                const uint32_t j = s_fixlist[m];                 // lane 32 sits on j
                const int pos = (int)j - 32 + (int)lane;
                const bool valid = pos >= 0 && pos < (int)cn;
                key = valid ? s_keys[pos] : 0ull;
                val = valid ? s_vals[pos] : 0u;
                const uint64_t kj = __shfl(key, 32, 64), k0 = __shfl(key, 31, 64);
                const unsigned long long same = __ballot(valid && gr_digit(key, gshift, gbits) == gr_digit(kj, gshift, gbits));
                const uint32_t nl = ~(uint32_t)same, nr = ~(uint32_t)(same >> 32);
                const uint32_t lrun = nl ? (uint32_t)__builtin_clz(nl) : 32u;     // records of the group right before j (>= 1)
                const uint32_t rrun = nr ? (uint32_t)__builtin_ctz(nr) : 32u;     // j and the records after it
                const uint32_t a0 = 32u - lrun, e0 = 32u + rrun;
                const unsigned long long left = ((1ull << 32) - 1ull) & ~((1ull << a0) - 1ull);    // lanes [a0, 32)
                const unsigned long long eq0 = __ballot(key == k0);
                if (lrun + rrun > 32u) {                         // may reach beyond what the window shows
                    if (lane == 0) s_redo = 1u;
                } else if ((eq0 & left) == left) {               // no earlier change of hash in the group: ours to sort
                    uint32_t rk = 0;
                    for (uint32_t q = a0; q < e0; ++q) {
                        const uint64_t kq = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), (int)q) << 32) |
                                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, (int)q);
                        rk += (kq < key || (kq == key && q < lane)) ? 1u : 0u;
                    }
                    if (lane >= a0 && lane < e0) dst = (uint32_t)((int)j - 32 + (int)a0) + rk;
                }
            }
            __syncthreads();
            if (dst != ~0u) { s_keys[dst] = key; s_vals[dst] = val; }
            __syncthreads();
        }
    };
    if (!(GR_ABL & 16)) sort_pass(shift_lo, bits_lo, s_wcnt[0]);
    if (!(GR_ABL & 20)) sort_pass(shift_hi, bits_hi, s_wcnt[1]);
    if (!in_lds) {      // no local digit at all: keep the order
#pragma unroll
        for (uint32_t r = 0; r < ITEMS; ++r) {
            const uint32_t j = w * SEG + r * 64u + lane;
            if (j < cn) { s_keys[j] = k[r]; s_vals[j] = v[r]; }
        }
        __syncthreads();
    }
    if (do_fix && cn && !(GR_ABL & 2)) {
        fix_groups(shift_lo, bits_lo + bits_hi);
        if (do_fix > 1u && s_redo) {                             // (uniform: read behind the barriers of fix_groups)
            __syncthreads();
            if (threadIdx.x == 0) { s_fixn = 0; s_redo = 0; }
            sort_pass(shift_lo - GR_LOCAL_BITS, GR_LOCAL_BITS, s_wcnt[0]);
            sort_pass(shift_lo, bits_lo, s_wcnt[1]);
            sort_pass(shift_hi, bits_hi, s_wcnt[0]);
            fix_groups(shift_lo - GR_LOCAL_BITS, bits_lo + bits_hi + GR_LOCAL_BITS);
        }
    }
    // the bucket is grouped and lives in LDS: the registers that held its records take the next bucket's
    uint32_t nb0 = 0, nb1 = 0;
    if (PREFETCH && c_next < nbuckets) load_bucket(c_next, nb0, nb1, k, v);
    // thread t takes records t, t + TPB, ... : run heads first (their number is what the buckets after
    // this one wait for), then — while wavefront 0 looks back — sorted occurrences and strengths
    unsigned long long hb[ITEMS];
    uint32_t occ[ITEMS];
    bool head[ITEMS];
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        const uint32_t j = i * TPB + threadIdx.x;
        head[i] = false; occ[i] = 0;
        if (j < cn) {
            const uint64_t key = s_keys[j];
            head[i] = j == 0u || s_keys[j - 1u] != key;
            occ[i] = s_vals[j];
        }
        hb[i] = __ballot(head[i]);
        if (lane == 0) s_seg[i * NW + w] = (uint32_t)__popcll(hb[i]);
    }
    __syncthreads();
    if (w == 0) {                          // exclusive scan of the 64 (row, wavefront) counts; the total is published at once
        const uint32_t mine = s_seg[lane];
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((int)lane >= o) inc += t; }
        s_seg[lane] = inc - mine;
        if (lane == 63u) {
            s_misc[1] = inc;
            if (status) __hip_atomic_store(&status[c], (c == 0 ? GR_ST_PREFIX : GR_ST_AGG) | (unsigned long long)inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else s_misc[0] = atomicAdd(n_runs_out, inc);        // run slots in the order the buckets finish
        }
    }
    __syncthreads();                       // everybody holds its occurrences in registers: s_vals becomes the head positions
    const uint32_t nruns = s_misc[1];
    uint16_t *hpos = reinterpret_cast<uint16_t *>(s_vals);
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i)
        if (head[i]) hpos[s_seg[i * NW + w] + gr_lanes_below(hb[i])] = (uint16_t)(i * TPB + threadIdx.x);
    if (w == 0 && status) {                // runs in bucket order (RB_GROUP_ORDERED=1): chained scan over the buckets
        uint32_t rb_ = 0;
        if (c != 0 && !(GR_ABL & 1)) rb_ = gr_look_back(status, c);
        if (lane == 0) {
            s_misc[0] = rb_;
            if (c != 0) __hip_atomic_store(&status[c], GR_ST_PREFIX | (unsigned long long)(rb_ + nruns), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c == nbuckets - 1u) *n_runs_out = rb_ + nruns;
        }
    }
    const uint32_t pmask = (1u << rng.pos_bits) - 1u;
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        const uint32_t j = i * TPB + threadIdx.x;
        if (j < cn && !(GR_ABL & 8)) {
            vals_out[b0 + j] = occ[i];
            const uint32_t rr = rng31(rng.seed, rng.ordinal0 + (uint64_t)(occ[i] >> rng.pos_bits), occ[i] & pmask) | 0x8000u;
            tz_out[b0 + j] = (uint8_t)(__ffs((int)rr) - 1);
        }
    }
    __syncthreads();
    const uint32_t run_base = s_misc[0];
    if (brun && threadIdx.x == 0) { brun[c] = run_base; bnr[c] = nruns; }     // (an oversized bucket: no runs here, k_group_big appends them)
    if (by_class && nruns > 64u && nruns != cn) {          // (nruns == cn: every run is one occurrence long — the all-new-k-mers regime)
        // Stage B walks a run's occurrences in a per-lane loop, so a wavefront takes as long as its longest run: the bucket's runs go out
        // ordered by length class (1, 2, 3-4, 5-8, ... 65+; long ones first), which makes the 64 runs of a wavefront alike.  Nothing
        // downstream depends on the order of the runs (ties anywhere are broken by occurrence ids).
        if (threadIdx.x < 16u) s_cls[threadIdx.x] = 0u;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nruns; i += TPB) {
            const uint32_t j = hpos[i], e = i + 1u < nruns ? (uint32_t)hpos[i + 1u] : cn;
            atomicAdd(&s_cls[min(7u, 32u - (uint32_t)__clz((int)(e - j - 1u)))], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t run = 0; for (int q = 7; q >= 0; --q) { const uint32_t t = s_cls[q]; s_cls[q] = run; run += t; } }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nruns; i += TPB) {
            const uint32_t j = hpos[i], e = i + 1u < nruns ? (uint32_t)hpos[i + 1u] : cn;
            const uint32_t q = min(7u, 32u - (uint32_t)__clz((int)(e - j - 1u)));
            const uint32_t slot = run_base + s_cls[q] + atomicAdd(&s_cls[8u + q], 1u);
            uniq[slot] = s_keys[j];
            starts[slot] = b0 + j;
            counts[slot] = e - j;
        }
    } else
    for (uint32_t i = threadIdx.x; i < nruns; i += TPB) {
        const uint32_t j = hpos[i], e = i + 1u < nruns ? (uint32_t)hpos[i + 1u] : cn;
        uniq[run_base + i] = s_keys[j];
        starts[run_base + i] = b0 + j;
        counts[run_base + i] = e - j;
    }
    if (PREFETCH) {
        if (c_next >= nbuckets) return;
        c = c_next; b0 = nb0; b1 = nb1;
    }
    }   // next bucket
}

// Buckets that do not fit LDS (a k-mer with thousands of surviving occurrences in the sub-batch, cold prefilter
// cache): one workgroup per bucket sorts it on the same local digits with the same stable counting passes, but
// through global memory (ping-pong between the record buffer the bucket lives in and the other one), piece by
// piece, and then streams over the result: a run may span any number of pieces, so a hot k-mer stays ONE run
// (k_cbf_heavy's case) instead of being cut.  Run slots are taken from the run counter the main kernel left
// (atomicAdd per piece: the order of these few runs among themselves is arbitrary, which nothing depends on).
template <int TPB>
__global__ void __launch_bounds__(TPB) k_group_big(uint64_t *keys_a, uint32_t *vals_a, uint64_t *keys_b, uint32_t *vals_b,
                                                   const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ big_list,
                                                   const uint32_t *__restrict__ n_big_dev,
                                                   uint32_t shift_lo, uint32_t bits_lo, uint32_t shift_hi, uint32_t bits_hi, GroupRng rng,
                                                   uint32_t *__restrict__ vals_out, uint8_t *__restrict__ tz_out,
                                                   uint64_t *__restrict__ uniq, uint32_t *__restrict__ counts, uint32_t *__restrict__ starts,
                                                   uint32_t *__restrict__ run_cursor) {
    constexpr uint32_t ITEMS = GR_TILE / TPB, NW = TPB / 64, SEG = 64 * ITEMS, NB = 1u << GR_LOCAL_BITS;
    __shared__ uint64_t s_keys[GR_TILE];
    __shared__ uint16_t s_hpos[GR_TILE];
    __shared__ uint16_t s_wcnt[NW * NB];
    __shared__ unsigned long long s_wmask[NW * NB];
    __shared__ uint32_t s_hist[NB], s_dbase[NB], s_tot[NB];
    __shared__ uint32_t s_wsum[NW], s_seg[ITEMS * NW], s_misc[4];
    __shared__ uint64_t s_carry_key;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t d = threadIdx.x; d < NW * NB; d += TPB) s_wmask[d] = 0ull;
    const uint32_t n_big = *n_big_dev;
    const uint32_t pmask = (1u << rng.pos_bits) - 1u;
    for (uint32_t bi = blockIdx.x; bi < n_big; bi += gridDim.x) {
        const uint32_t c = big_list[bi], b0 = bstart[c], b1 = bstart[c + 1u];
        uint64_t *ksrc = keys_a, *kdst = keys_b;
        uint32_t *vsrc = vals_a, *vdst = vals_b;
        for (int pass = 0; pass < 2; ++pass) {
            const uint32_t shift = pass ? shift_hi : shift_lo, bits = pass ? bits_hi : bits_lo;
            if (bits == 0) continue;
            const uint32_t nb = 1u << bits;
            __syncthreads();
            for (uint32_t d = threadIdx.x; d < nb; d += TPB) s_hist[d] = 0;
            __syncthreads();
            for (uint32_t x = b0 + threadIdx.x; x < b1; x += TPB) atomicAdd(&s_hist[gr_digit(ksrc[x], shift, bits)], 1u);
            __syncthreads();
            {   // exclusive scan of the histogram -> running bases per digit
                const uint32_t per = (nb + TPB - 1) / TPB;       // 1 (nb <= 256 <= TPB)
                uint32_t vsum = 0;
                if (threadIdx.x * per < nb) vsum = s_hist[threadIdx.x];
                const uint32_t ex = gr_block_excl_scan<TPB>(vsum, s_wsum);
                if (threadIdx.x < nb) s_dbase[threadIdx.x] = ex;
            }
            __syncthreads();
            for (uint32_t p0 = b0; p0 < b1; p0 += GR_TILE) {
                const uint32_t cn = min(GR_TILE, b1 - p0);
                const uint32_t rows = cn > w * SEG ? min(ITEMS, (cn - w * SEG + 63u) / 64u) : 0u;
                uint64_t k[ITEMS];
                uint32_t v[ITEMS], dig[ITEMS], rank[ITEMS];
#pragma unroll
                for (uint32_t r = 0; r < ITEMS; ++r) {
                    const uint32_t j = w * SEG + r * 64u + lane;
                    const bool ok = j < cn;
                    k[r] = ok ? ksrc[p0 + j] : 0ull;
                    v[r] = ok ? vsrc[p0 + j] : 0u;
                    dig[r] = ok ? gr_digit(k[r], shift, bits) : ~0u;
                }
                for (uint32_t d = threadIdx.x; d < NW * nb; d += TPB) s_wcnt[d] = 0;
                __syncthreads();
                gr_wave_rank_lds<ITEMS>(dig, rank, s_wcnt + w * nb, s_wmask + w * NB, rows);
                __syncthreads();
                if (threadIdx.x < nb) {      // exclusive prefix over the wavefronts, total of the piece
                    uint32_t run = 0;
#pragma unroll
                    for (uint32_t ww = 0; ww < NW; ++ww) { const uint32_t t = s_wcnt[ww * nb + threadIdx.x]; s_wcnt[ww * nb + threadIdx.x] = (uint16_t)run; run += t; }
                    s_tot[threadIdx.x] = run;
                }
                __syncthreads();
#pragma unroll
                for (uint32_t r = 0; r < ITEMS; ++r)
                    if (dig[r] != ~0u) {
                        const uint32_t g = b0 + s_dbase[dig[r]] + (uint32_t)s_wcnt[w * nb + dig[r]] + rank[r];
                        kdst[g] = k[r]; vdst[g] = v[r];
                    }
                __syncthreads();
                if (threadIdx.x < nb) s_dbase[threadIdx.x] += s_tot[threadIdx.x];
                __syncthreads();
            }
            { uint64_t *t = ksrc; ksrc = kdst; kdst = t; }
            { uint32_t *t = vsrc; vsrc = vdst; vdst = t; }
            __threadfence();
            __syncthreads();
        }
        // stream over the grouped bucket: strengths, occurrences, runs (the open run is carried from piece to piece)
        uint32_t carry_start = b0;               // block-uniform copies; the key lives in s_carry_key
        for (uint32_t p0 = b0; p0 < b1; p0 += GR_TILE) {
            const uint32_t cn = min(GR_TILE, b1 - p0);
            const bool first = p0 == b0;
            __syncthreads();
            for (uint32_t j = threadIdx.x; j < cn; j += TPB) s_keys[j] = ksrc[p0 + j];
            __syncthreads();
            unsigned long long hb[ITEMS];
            bool head[ITEMS];
            const uint64_t ckey = first ? 0ull : s_carry_key;
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; ++i) {
                const uint32_t j = i * TPB + threadIdx.x;
                head[i] = false;
                if (j < cn) {
                    const uint64_t key = s_keys[j];
                    head[i] = j == 0u ? (first || key != ckey) : s_keys[j - 1u] != key;
                    const uint32_t occ = vsrc[p0 + j];
                    vals_out[p0 + j] = occ;
                    const uint32_t rr = rng31(rng.seed, rng.ordinal0 + (uint64_t)(occ >> rng.pos_bits), occ & pmask) | 0x8000u;
                    tz_out[p0 + j] = (uint8_t)(__ffs((int)rr) - 1);
                }
                hb[i] = __ballot(head[i]);
                if (lane == 0) s_seg[i * NW + w] = (uint32_t)__popcll(hb[i]);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t run = 0;
                for (uint32_t q = 0; q < ITEMS * NW; ++q) { const uint32_t t = s_seg[q]; s_seg[q] = run; run += t; }
                s_misc[1] = run;                                   // heads in this piece
                const uint32_t closed = run - (first ? 1u : 0u);   // every head closes the run before it, except the very first
                s_misc[0] = closed ? atomicAdd(run_cursor, closed) : 0u;
            }
            __syncthreads();
            const uint32_t nheads = s_misc[1], base = s_misc[0];
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; ++i)
                if (head[i]) s_hpos[s_seg[i * NW + w] + gr_lanes_below(hb[i])] = (uint16_t)(i * TPB + threadIdx.x);
            __syncthreads();
            // head i closes: the carried run (i == 0, not the first piece) or the run that began at head i-1
            for (uint32_t i = threadIdx.x; i < nheads; i += TPB) {
                const uint32_t j = s_hpos[i];
                if (i == 0u) {
                    if (!first) { uniq[base] = ckey; starts[base] = carry_start; counts[base] = p0 + j - carry_start; }
                } else {
                    const uint32_t jp = s_hpos[i - 1u], slot = base + i - (first ? 1u : 0u);
                    uniq[slot] = s_keys[jp]; starts[slot] = p0 + jp; counts[slot] = j - jp;
                }
            }
            __syncthreads();
            if (nheads) {                           // the last head of the piece opens the run that is carried on
                const uint32_t jl = s_hpos[nheads - 1u];
                carry_start = p0 + jl;
                if (threadIdx.x == 0) s_carry_key = s_keys[jl];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {                     // the run still open at the end of the bucket
            const uint32_t slot = atomicAdd(run_cursor, 1u);
            uniq[slot] = s_carry_key; starts[slot] = carry_start; counts[slot] = b1 - carry_start;
        }
        __syncthreads();
    }
}

// ---- host side ---------------------------------------------------------------------------------------
struct GroupPlan {
    uint32_t n = 0, T = 0;
    uint32_t t_lo = 0, t_hi = 0;             // partition passes (bits; 0 = no pass), MSD order: hi first
    uint32_t shift_lo = 0, shift_hi = 0;
    uint32_t l_lo = 0, l_hi = 0, lshift_lo = 0, lshift_hi = 0;   // local passes (LSD: lo first)
    uint32_t ntiles = 0, ntiles2_max = 0, nbuckets = 1;
    bool dead = false;                       // the input may hold records the emit pass cancelled (GR_DEAD_KEY / GR_DEAD_VAL): the first partition pass drops them
    uint32_t tpb = 256;
    uint32_t fix_cap = 0;                    // 1: groups of up to 32 records that agree in all sorted bits are sorted on the full hash; 2: buckets with longer ones are redone
    int xcd_map = 1;
    size_t hist_entries = 0;
    // layout of the temporary storage
    size_t off_status = 0, off_ticket = 0, off_bstart = 0, off_big = 0, off_hist = 0, off_goffs = 0, off_scan = 0, off_desc = 0, off_segtb = 0, off_segst = 0,
           scan_bytes = 0, total = 0;
};
static size_t gr_align(size_t x) { return (x + 255) / 256 * 256; }

static GroupPlan group_plan(size_t N, int group_bits, int bucket_target = 0, int flags = 0, int force_T = -1) {
    GroupPlan P;
    P.n = (uint32_t)N;
    P.dead = (flags & GR_FLAG_DEAD) != 0;
    const uint32_t gb = (uint32_t)std::max(1, std::min<int>(group_bits, (int)GR_MAX_GROUP_BITS));
    uint32_t T = 0;
    // smaller buckets = fewer distinct hashes per bucket = fewer hashes that agree in the 16 locally sorted bits (1 % of the hashes
    // at 3072, 0.25 % at 768) at the price of more workgroup rounds in the bucket kernel; since the bucket kernel repairs such
    // groups itself, the target is 3072 everywhere.  A caller that names a target (the sharded engine: 4x longer sub-batches, so
    // longer runs, and every conflicting op is routed between ranks) also gets the thorough repair (fix level 2).  Measured with
    // 8 virtual ranks, per pass: no repair, 3072: 179 M conflicting ops; no repair, 768: 101 M (grouping 203 ms, conflict
    // routing 100 ms); repair level 2, 3072: 68 M (grouping 172 ms, routing 39 ms).
    const size_t target = getenv("RB_GROUP_TARGET") ? (size_t)std::max(64, std::min(atoi(getenv("RB_GROUP_TARGET")), (int)GR_TILE))
                        : bucket_target > 0 ? (size_t)bucket_target : (size_t)GR_BUCKET_TARGET;
    while ((target << T) < N) ++T;
    if (const char *e = getenv("RB_GROUP_T")) T = (uint32_t)std::max(0, atoi(e));
    if (force_T >= 0) T = (uint32_t)force_T;   // (binning by index range into the fine buckets of another grouping: rb_sweep below)
    T = std::min({T, gb, 2u * GR_PART_MAX_BITS});
    if (P.dead && T == 0) T = 1;             // cancelled records leave in a partition pass: there has to be one
    P.T = T;
    if (T <= GR_PART_MAX_BITS) { P.t_hi = T; P.t_lo = 0; }
    else { P.t_hi = (T + 1u) / 2u; P.t_lo = T - P.t_hi; }
    P.shift_hi = GR_KEY_TOP - P.t_hi;
    P.shift_lo = GR_KEY_TOP - T;
    const uint32_t L = std::min(gb - T, 2u * GR_LOCAL_BITS);
    P.l_hi = (L + 1u) / 2u;
    P.l_lo = L - P.l_hi;
    P.lshift_hi = GR_KEY_TOP - T - P.l_hi;
    P.lshift_lo = GR_KEY_TOP - T - L;
    P.tpb = getenv("RB_GROUP_TPB") ? (uint32_t)atoi(getenv("RB_GROUP_TPB")) : 512u;
    if (P.tpb != 256u) P.tpb = 512u;
    P.xcd_map = getenv("RB_GROUP_XCD") ? atoi(getenv("RB_GROUP_XCD")) : 1;
    // callers that ask for fewer grouping bits than we have want split runs (tests of the arbitration); everybody else gets them repaired
    P.fix_cap = L == 2u * GR_LOCAL_BITS ? (bucket_target > 0 ? 2u : 1u) : 0u;
    if (const char *e = getenv("RB_GROUP_FIX")) P.fix_cap = (uint32_t)std::max(0, std::min(atoi(e), 2));
    P.ntiles = (uint32_t)((N + GR_TILE - 1) / GR_TILE);
    P.ntiles2_max = P.t_lo ? P.ntiles + (1u << P.t_hi) : 0;
    P.nbuckets = 1u << T;
    if (T) P.hist_entries = std::max((size_t)P.ntiles << P.t_hi, (size_t)P.ntiles2_max << P.t_lo) + 2;
    size_t o = 0;
    P.off_status = o; o += gr_align((size_t)P.nbuckets * 8);
    P.off_ticket = o; o += 256;
    P.off_bstart = o; o += gr_align(((size_t)P.nbuckets + 1) * 4);
    P.off_big = o; o += gr_align((size_t)P.nbuckets * 4);
    if (T) {
        P.off_hist = o; o += gr_align(P.hist_entries * 4);
        P.off_goffs = o; o += gr_align(P.hist_entries * 4);
        P.scan_bytes = gr_align(scan_temp_bytes(P.hist_entries));
        P.off_scan = o; o += P.scan_bytes;
        if (P.t_lo) {
            P.off_desc = o; o += gr_align((size_t)P.ntiles2_max * sizeof(GrTile));
            P.off_segtb = o; o += gr_align(((size_t)(1u << P.t_hi) + 1) * 4);
            P.off_segst = o; o += gr_align(((size_t)(1u << P.t_hi) + 1) * 4);
        }
    }
    P.total = o;
    return P;
}

// debugging aid (RB_DEBUG): the buckets that did not fit LDS in the last grouping of N records that used `temp` (call with the stream idle)
void group_debug_big(const void *temp, size_t N, int group_bits, int bucket_target, uint32_t *n_big_out, uint64_t *records_out, uint32_t *largest_out, int flags) {
    const GroupPlan P = group_plan(N, group_bits, bucket_target, flags);
    const char *tp = static_cast<const char *>(temp);
    uint32_t tick[4] = {0, 0, 0, 0};
    RB_HIP(hipMemcpy(tick, tp + P.off_ticket, 16, hipMemcpyDeviceToHost));
    const uint32_t nb = std::min(tick[2], P.nbuckets);
    std::vector<uint32_t> big(nb), bs((size_t)P.nbuckets + 1);
    if (nb) RB_HIP(hipMemcpy(big.data(), tp + P.off_big, (size_t)nb * 4, hipMemcpyDeviceToHost));
    RB_HIP(hipMemcpy(bs.data(), tp + P.off_bstart, ((size_t)P.nbuckets + 1) * 4, hipMemcpyDeviceToHost));
    uint64_t rec = 0; uint32_t mx = 0;
    for (uint32_t i = 0; i < nb; ++i) { const uint32_t c = bs[big[i] + 1] - bs[big[i]]; rec += c; mx = std::max(mx, c); }
    *n_big_out = nb; *records_out = rec; *largest_out = mx;
}
size_t group_temp_bytes(size_t N, int group_bits, int bucket_target, int flags) { return group_plan(N, group_bits, bucket_target, flags).total; }
// where the grouping of N records with GR_FLAG_DEAD leaves the number of live records (device address inside `temp`)
const uint32_t *group_live_count(const void *temp, size_t N, int group_bits, int bucket_target, int flags) {
    const GroupPlan P = group_plan(N, group_bits, bucket_target, flags);
    return reinterpret_cast<const uint32_t *>(static_cast<const char *>(temp) + P.off_ticket) + 3;
}

template <int TPB>
static void part_pass(const GrTiling &tl, size_t entries, uint32_t shift, uint32_t bits, const uint64_t *kin, const uint32_t *vin, uint64_t *kout,
                      uint32_t *vout, uint32_t *hist, uint32_t *goffs, void *scan_tmp, size_t scan_bytes, hipStream_t st, rb_graph *prof,
                      uint32_t *n_live_dev = nullptr /* non-null: the pass drops cancelled records and leaves the number of live ones here */,
                      GrIdxDev ix = GrIdxDev{Mod{1, 0, 0}, 0, 0, 0, 0}, uint64_t derive_kmul = 0 /* != 0: kin holds run hashes, the records are (second hash, position) */) {
    const dim3 grid(tl.grid_tiles), blk(TPB);
    if (prof) prof->prof_begin(st);
    if (n_live_dev) RB_HIP(hipMemsetAsync(hist + entries, 0, 4, st));        // one entry more: its scanned value is the total
    if (derive_kmul) hipLaunchKernelGGL((k_part_count<TPB, true>), grid, blk, 0, st, kin, tl, shift, bits, hist, (const uint32_t *)nullptr, ix, derive_kmul);
    else hipLaunchKernelGGL(k_part_count<TPB>, grid, blk, 0, st, kin, tl, shift, bits, hist, n_live_dev ? vin : (const uint32_t *)nullptr, ix);
    if (prof) { prof->prof_end("group_part_count", st); prof->prof_begin(st); }
    exclusive_scan_u32(scan_tmp, scan_bytes, hist, goffs, entries + (n_live_dev ? 1 : 0), st);
    if (n_live_dev) hipLaunchKernelGGL(k_copy_u32, dim3(1), dim3(64), 0, st, n_live_dev, goffs + entries);
    if (prof) { prof->prof_end("group_scan", st); prof->prof_begin(st); }
    const uint32_t wide = !(getenv("RB_GROUP_WIDE_LDS") && atoi(getenv("RB_GROUP_WIDE_LDS")) == 0);
    if (derive_kmul) {
        if (bits <= 8u) hipLaunchKernelGGL((k_part_scatter<TPB, 8, true>), grid, blk, 0, st, kin, (const uint32_t *)nullptr, tl, shift, bits, goffs, kout, vout, wide, 0u, ix, derive_kmul);
        else hipLaunchKernelGGL((k_part_scatter<TPB, GR_PART_MAX_BITS, true>), grid, blk, 0, st, kin, (const uint32_t *)nullptr, tl, shift, bits, goffs, kout, vout, wide, 0u, ix, derive_kmul);
    } else if (bits <= 8u) hipLaunchKernelGGL((k_part_scatter<TPB, 8>), grid, blk, 0, st, kin, vin, tl, shift, bits, goffs, kout, vout, wide, n_live_dev ? 1u : 0u, ix);
    else hipLaunchKernelGGL((k_part_scatter<TPB, GR_PART_MAX_BITS>), grid, blk, 0, st, kin, vin, tl, shift, bits, goffs, kout, vout, wide, n_live_dev ? 1u : 0u, ix);
    if (prof) prof->prof_end("group_part_scatter", st);
}

// ---- LSD radix sort out of the same stable partition passes --------------------------------------------------
// The conflict path (rb_graph.hip, rb_shard.hip) and the sketch sets (rb_sketch.hip) need a few full sorts of small arrays
// (10^5 ... 10^7 keys).  A stable partition on one digit, repeated from the lowest digit up, IS an LSD radix sort, and the
// stable partition exists above: k_part_count -> scan -> k_part_scatter on a uniform tiling.  Digits of up to 10 bits; the bit
// ranges a caller knows to be constant are skipped; the passes ping-pong between the input and output arrays (the INPUT arrays
// are scratch: every caller is done with them) and an even pass count ends with one copy.  Replaces rocPRIM's onesweep (rounds 1-3).
namespace {
struct LsdPlan {
    uint32_t ntiles = 0, n_pass = 0, shift[16], bits[16];
    size_t entries = 0, off_hist = 0, off_goffs = 0, off_scan = 0, scan_bytes = 0, off_idx = 0, total = 0;
};
LsdPlan lsd_plan(size_t n, const int (*ranges)[2], int n_ranges, bool idx_vals) {
    LsdPlan P;
    P.ntiles = (uint32_t)((n + GR_TILE - 1) / GR_TILE);
    uint32_t maxbits = 0;
    for (int r = 0; r < n_ranges; ++r) {
        const uint32_t B = (uint32_t)std::max(0, ranges[r][1] - ranges[r][0]);
        // digits of at most 10 bits, spread evenly; 8 bits for small arrays: the histogram is tiles x 2^bits entries, and below ~1 M
        // records a 10-bit histogram is a quarter of the records themselves and its scan the longest kernel of the pass
        const uint32_t maxb = n <= ((size_t)1 << 20) ? 8u : GR_PART_MAX_BITS;
        const uint32_t np = (B + maxb - 1) / maxb;
        uint32_t lo = (uint32_t)ranges[r][0], left = B;
        for (uint32_t q = 0; q < np; ++q) {
            const uint32_t b = (left + (np - q) - 1) / (np - q);
            P.shift[P.n_pass] = lo; P.bits[P.n_pass] = b; ++P.n_pass;
            lo += b; left -= b;
            maxbits = std::max(maxbits, b);
        }
    }
    P.entries = ((size_t)P.ntiles << maxbits) + 2;
    size_t o = 0;
    P.off_hist = o; o += gr_align(P.entries * 4);
    P.off_goffs = o; o += gr_align(P.entries * 4);
    P.scan_bytes = gr_align(scan_temp_bytes(P.entries));
    P.off_scan = o; o += P.scan_bytes;
    if (idx_vals) { P.off_idx = o; o += 2 * gr_align(n * 4); }
    P.total = o;
    return P;
}
void lsd_sort(const LsdPlan &P, char *tp, uint64_t *k_in, uint64_t *k_out, uint32_t *v_in, uint32_t *v_out, size_t n, hipStream_t st) {
    uint32_t *hist = reinterpret_cast<uint32_t *>(tp + P.off_hist), *goffs = reinterpret_cast<uint32_t *>(tp + P.off_goffs);
    const GrTiling tl{nullptr, nullptr, (uint32_t)n, P.ntiles, gr_grid_for_tiles(P.ntiles), 1};
    uint64_t *ka = k_in, *kb = k_out;
    uint32_t *va = v_in, *vb = v_out;
    for (uint32_t q = 0; q < P.n_pass; ++q) {
        part_pass<512>(tl, (size_t)P.ntiles << P.bits[q], P.shift[q], P.bits[q], ka, va, kb, vb, hist, goffs, tp + P.off_scan, P.scan_bytes, st, nullptr);
        std::swap(ka, kb); std::swap(va, vb);
    }
    if ((P.n_pass & 1u) == 0) {                         // an even number of passes (or none) leaves the result in the input arrays
        RB_HIP(hipMemcpyAsync(k_out, k_in, n * 8, hipMemcpyDeviceToDevice, st));
        if (v_in && v_out) RB_HIP(hipMemcpyAsync(v_out, v_in, n * 4, hipMemcpyDeviceToDevice, st));
    }
}
__global__ void k_lsd_iota(uint32_t *__restrict__ v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
__global__ void k_lsd_gather64(const uint32_t *__restrict__ idx, const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
}  // namespace

size_t sort_pairs_temp_bytes(size_t n) { const int r[1][2] = {{0, 64}}; return lsd_plan(n, r, 1, false).total; }
size_t sort_keys_temp_bytes(size_t n) { return sort_pairs_temp_bytes(n); }
size_t sort_pairs32_temp_bytes(size_t n) { const int r[1][2] = {{0, 64}}; return lsd_plan(n, r, 1, true).total; }
// stable sort of (key, value) pairs on key bits [begin_bit, end_bit); keys_in / vals_in are clobbered
void sort_pairs_u64_u32(void *temp, size_t temp_bytes, uint64_t *keys_in, uint64_t *keys_out, uint32_t *vals_in, uint32_t *vals_out, size_t n,
                        int begin_bit, int end_bit, hipStream_t s) {
    if (n == 0) return;
    RB_REQUIRE(n < (1ull << 32) - 2 * GR_TILE, "sort_pairs_u64_u32: too many records");
    const int r[1][2] = {{begin_bit, end_bit}};
    const LsdPlan P = lsd_plan(n, r, 1, false);
    RB_REQUIRE(temp_bytes >= P.total, "sort_pairs_u64_u32: temp too small");
    lsd_sort(P, static_cast<char *>(temp), keys_in, keys_out, vals_in, vals_out, n, s);
}
// the same on two bit ranges, the lower one first (the bits between them are known to be equal in all keys)
void sort_pairs_u64_u32_2r(void *temp, size_t temp_bytes, uint64_t *keys_in, uint64_t *keys_out, uint32_t *vals_in, uint32_t *vals_out, size_t n,
                           int lo_begin, int lo_end, int hi_begin, int hi_end, hipStream_t s) {
    if (n == 0) return;
    RB_REQUIRE(n < (1ull << 32) - 2 * GR_TILE, "sort_pairs_u64_u32_2r: too many records");
    const int r[2][2] = {{lo_begin, lo_end}, {hi_begin, hi_end}};
    const LsdPlan P = lsd_plan(n, r, 2, false);
    RB_REQUIRE(temp_bytes >= P.total, "sort_pairs_u64_u32_2r: temp too small");
    lsd_sort(P, static_cast<char *>(temp), keys_in, keys_out, vals_in, vals_out, n, s);
}
void sort_keys_u64(void *temp, size_t temp_bytes, uint64_t *keys_in, uint64_t *keys_out, size_t n, int begin_bit, int end_bit, hipStream_t s) {
    sort_pairs_u64_u32(temp, temp_bytes, keys_in, keys_out, nullptr, nullptr, n, begin_bit, end_bit, s);
}
// 64-bit values: the pairs are sorted as (key, index) and the values gathered through the sorted indices
void sort_pairs_u64_u64(void *temp, size_t temp_bytes, uint64_t *keys_in, uint64_t *keys_out, uint64_t *vals_in, uint64_t *vals_out, size_t n,
                        int begin_bit, int end_bit, hipStream_t s) {
    if (n == 0) return;
    RB_REQUIRE(n < (1ull << 32) - 2 * GR_TILE, "sort_pairs_u64_u64: too many records");
    const int r[1][2] = {{begin_bit, end_bit}};
    const LsdPlan P = lsd_plan(n, r, 1, true);
    RB_REQUIRE(temp_bytes >= P.total, "sort_pairs_u64_u64: temp too small");
    char *tp = static_cast<char *>(temp);
    uint32_t *i0 = reinterpret_cast<uint32_t *>(tp + P.off_idx), *i1 = reinterpret_cast<uint32_t *>(tp + P.off_idx + gr_align(n * 4));
    hipLaunchKernelGGL(k_lsd_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, i0, n);
    lsd_sort(P, tp, keys_in, keys_out, i0, i1, n, s);
    hipLaunchKernelGGL(k_lsd_gather64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, i1, vals_in, vals_out, n);
}

// The MSD partition of the records into the plan's 2^T fine buckets (one or two stable passes); on return (*kin, *vin) is where the
// partitioned records sit — keys0 / vals0 or the scratch pair — and bstart[2^T + 1] (inside tp) holds the bucket bounds.
static GrIdxDev gr_idx_dev(const GroupPlan &P, const GrIdx &idx) {
    if (idx.span > (1ull << P.T))              // (mul must fit 64 bits: more indices than fine buckets — anything but a toy filter)
        return GrIdxDev{idx.mod, idx.lo, (uint64_t)((((unsigned __int128)1 << 64) << P.T) / idx.span), (1u << P.T) - 1u, P.t_lo};
    return GrIdxDev{Mod{1, 0, 0}, 0, 0, 0, 0};
}
template <int TPB>
static uint32_t *partition_records(const GroupPlan &P, uint64_t *keys0, uint32_t *vals0, uint64_t *keys_tmp, uint32_t *vals_tmp, char *tp, hipStream_t st,
                                   rb_graph *prof, GrIdx idx, const uint64_t **kin_out, const uint32_t **vin_out,
                                   const uint64_t *derive_src = nullptr, uint64_t derive_kmul = 0 /* both given: the first pass makes its records from the
                                   run hashes derive_src[0 .. n) — (second hash, position) — and keys0 / vals0 are scratch like the other pair */) {
    unsigned long long *status = reinterpret_cast<unsigned long long *>(tp + P.off_status);
    uint32_t *ticket = reinterpret_cast<uint32_t *>(tp + P.off_ticket);
    uint32_t *bstart = reinterpret_cast<uint32_t *>(tp + P.off_bstart);
    RB_HIP(hipMemsetAsync(status, 0, P.off_bstart, st));        // status words + ticket
    const uint64_t *kin = derive_src ? derive_src : keys0;
    const uint32_t *vin = derive_src ? nullptr : vals0;
    if (P.T) {
        uint32_t *hist = reinterpret_cast<uint32_t *>(tp + P.off_hist), *goffs = reinterpret_cast<uint32_t *>(tp + P.off_goffs);
        void *scan_tmp = tp + P.off_scan;
        GrTiling t1{nullptr, nullptr, P.n, P.ntiles, gr_grid_for_tiles(P.ntiles), P.xcd_map};
        uint32_t *n_live = P.dead ? ticket + 3 : nullptr;      // (the status / ticket block was zeroed above)
        GrIdxDev ix = gr_idx_dev(P, idx);
        part_pass<TPB>(t1, (size_t)P.ntiles << P.t_hi, P.shift_hi, P.t_hi, kin, vin, keys_tmp, vals_tmp, hist, goffs, scan_tmp, P.scan_bytes, st, prof, n_live, ix, derive_src ? derive_kmul : 0);
        kin = keys_tmp; vin = vals_tmp;
        uint32_t *segtb = nullptr, *segst = nullptr;
        if (P.t_lo) {
            GrTile *desc = reinterpret_cast<GrTile *>(tp + P.off_desc);
            segtb = reinterpret_cast<uint32_t *>(tp + P.off_segtb); segst = reinterpret_cast<uint32_t *>(tp + P.off_segst);
            uint32_t *nt2 = ticket + 1;
            hipLaunchKernelGGL(k_seg_tiles, dim3(1), dim3(1024), 0, st, goffs, P.ntiles, 1u << P.t_hi, P.n, 1u << P.t_lo, desc, segtb, segst, nt2, (const uint32_t *)n_live);
            RB_HIP(hipMemsetAsync(hist, 0, ((size_t)P.ntiles2_max << P.t_lo) * 4, st));
            GrTiling t2{desc, nt2, P.n, P.ntiles2_max, gr_grid_for_tiles(P.ntiles2_max), P.xcd_map};
            ix.shift = 0;                          // (index-keyed: the low t_lo bits of the fine-bucket number)
            part_pass<TPB>(t2, (size_t)P.ntiles2_max << P.t_lo, P.shift_lo, P.t_lo, kin, vin, keys0, vals0, hist, goffs, scan_tmp, P.scan_bytes, st, prof, nullptr, ix);
            kin = keys0; vin = vals0;
        }
        hipLaunchKernelGGL(k_bucket_bounds, dim3((P.nbuckets + 256u) / 256u), dim3(256), 0, st, goffs, P.ntiles, segtb, segst, P.t_hi, P.t_lo, P.n, bstart, (const uint32_t *)n_live);
    } else
        hipLaunchKernelGGL(k_bucket_bounds, dim3(1), dim3(64), 0, st, (const uint32_t *)nullptr, 0u, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, 0u, P.n, bstart);
    *kin_out = kin; *vin_out = vin;
    return bstart;
}

// Groups the N records (keys0, vals0) — both arrays are clobbered; (keys_tmp, vals_tmp) is scratch of the same size.
// Outputs: vals_out[N] occurrences in grouped order, tz_out[N] their strengths, runs (uniq, counts, starts) and
// *n_runs_dev.  Everything is enqueued on `st`; nothing is synchronised.
template <int TPB>
static void group_records_impl(const GroupPlan &P, uint64_t *keys0, uint32_t *vals0, uint64_t *keys_tmp, uint32_t *vals_tmp, GroupRng rng, char *tp,
                               uint32_t *vals_out, uint8_t *tz_out, uint64_t *uniq, uint32_t *counts, uint32_t *starts, uint32_t *n_runs_dev,
                               hipStream_t st, rb_graph *prof, GrIdx idx, GroupExport ex) {
    unsigned long long *status = reinterpret_cast<unsigned long long *>(tp + P.off_status);
    uint32_t *ticket = reinterpret_cast<uint32_t *>(tp + P.off_ticket);
    const uint64_t *kin = nullptr;
    const uint32_t *vin = nullptr;
    uint32_t *bstart = partition_records<TPB>(P, keys0, vals0, keys_tmp, vals_tmp, tp, st, prof, idx, &kin, &vin);
    uint32_t *big_list = reinterpret_cast<uint32_t *>(tp + P.off_big), *n_big = ticket + 2;
    if (prof) prof->prof_begin(st);
    const uint32_t bucket_grid = std::min(P.nbuckets, (uint32_t)(getenv("RB_GROUP_GRID") ? atoi(getenv("RB_GROUP_GRID")) : 768));
    // Run slots: nothing downstream depends on the order of the runs (the runs of oversized buckets were always appended in
    // arbitrary order), so a bucket takes its slots with one atomicAdd when its run count is known.  RB_GROUP_ORDERED=1 keeps
    // the runs in bucket order through a chained scan over the buckets (a quarter of the kernel's time: a bucket's look-back
    // walks the ~500 buckets in flight before it).
    const bool ordered = getenv("RB_GROUP_ORDERED") && atoi(getenv("RB_GROUP_ORDERED")) != 0;
    if (!ordered) RB_HIP(hipMemsetAsync(n_runs_dev, 0, 4, st));
    const bool prefetch = !(getenv("RB_GROUP_PREFETCH") && atoi(getenv("RB_GROUP_PREFETCH")) == 0);
    const uint32_t by_class = getenv("RB_GROUP_CLASSES") ? (uint32_t)(atoi(getenv("RB_GROUP_CLASSES")) != 0) : 1u;    // a bucket's runs ordered by length class (stage B's wavefronts alike)
    if (ordered || !prefetch)
        hipLaunchKernelGGL((k_group_buckets<TPB, false>), dim3(bucket_grid), dim3(TPB), 0, st, kin, vin, bstart, P.nbuckets, P.lshift_lo, P.l_lo, P.lshift_hi, P.l_hi,
                           rng, P.fix_cap, ticket, ordered ? status : nullptr, big_list, n_big, vals_out, tz_out, uniq, counts, starts, n_runs_dev, 0u, ex.brun, ex.bnr);
    else
        hipLaunchKernelGGL((k_group_buckets<TPB, true>), dim3(bucket_grid), dim3(TPB), 0, st, kin, vin, bstart, P.nbuckets, P.lshift_lo, P.l_lo, P.lshift_hi, P.l_hi,
                           rng, P.fix_cap, ticket, nullptr, big_list, n_big, vals_out, tz_out, uniq, counts, starts, n_runs_dev, by_class, ex.brun, ex.bnr);
    if (ex.n_main) hipLaunchKernelGGL(k_copy_u32, dim3(1), dim3(64), 0, st, ex.n_main, n_runs_dev);       // the runs from here on are the oversized buckets'
    if (prof) { prof->prof_end("group_buckets", st); prof->prof_begin(st); }
    // the buckets that do not fit LDS (none in a warm steady state): sorted through the record buffer that is free now
    uint64_t *ka = const_cast<uint64_t *>(kin), *kb = kin == keys0 ? keys_tmp : keys0;
    uint32_t *va = const_cast<uint32_t *>(vin), *vb = vin == vals0 ? vals_tmp : vals0;
    hipLaunchKernelGGL(k_group_big<512>, dim3(std::min(P.nbuckets, 2048u)), dim3(512), 0, st, ka, va, kb, vb, bstart, big_list, n_big,
                       P.lshift_lo, P.l_lo, P.lshift_hi, P.l_hi, rng, vals_out, tz_out, uniq, counts, starts, n_runs_dev);
    if (prof) prof->prof_end("group_big_buckets", st);
    RB_HIP(hipGetLastError());
}

void group_records_device(uint64_t *keys0, uint32_t *vals0, uint64_t *keys_tmp, uint32_t *vals_tmp, size_t N, int group_bits,
                          uint64_t seed, uint64_t ordinal0, uint32_t pos_bits, void *temp, size_t temp_bytes,
                          uint32_t *vals_out, uint8_t *tz_out, uint64_t *uniq, uint32_t *counts, uint32_t *starts, uint32_t *n_runs_dev,
                          hipStream_t st, rb_graph *prof, int bucket_target, int flags, GrIdx idx, GroupExport ex) {
    RB_REQUIRE(N > 0 && N < (1ull << 32) - 2 * GR_TILE, "group_records_device: bad record count");
    const GroupPlan P = group_plan(N, group_bits, bucket_target, flags);
    RB_REQUIRE(temp_bytes >= P.total, "group_records_device: temp too small");
    const GroupRng rng{seed, ordinal0, pos_bits};
    if (P.tpb == 512u) group_records_impl<512>(P, keys0, vals0, keys_tmp, vals_tmp, rng, static_cast<char *>(temp), vals_out, tz_out, uniq, counts, starts, n_runs_dev, st, prof, idx, ex);
    else group_records_impl<256>(P, keys0, vals0, keys_tmp, vals_tmp, rng, static_cast<char *>(temp), vals_out, tz_out, uniq, counts, starts, n_runs_dev, st, prof, idx, ex);
}
uint32_t group_index_buckets(size_t N, int group_bits, int bucket_target, int flags, GrIdx idx) {
    const GroupPlan P = group_plan(N, group_bits, bucket_target, flags);
    return (P.T && idx.span > (1ull << P.T)) ? P.T : 0u;
}

// ---- swept Bloom-bit stage -------------------------------------------------------------------------------------------------
// Stage A tests and sets two Bloom bits per run.  With the grouping keyed by the first index, probe 0 of the runs of fine bucket c falls
// into c's index range, but probe 1 goes anywhere: a random word load and a returning device-scope atomicOr per run, which execute at the
// memory side at 18-27 G/s however the words are laid out — 0.13 ms per million runs, three quarters of an insert where most k-mers are
// new (long reads: profiles/r04_group_idx.txt).  Here the second probes are BINNED by the same 2^T index ranges (the grouping's own
// stable partition passes over (h1, run) records), and one workgroup per range takes the range's words into LDS, applies the probes 0 of
// its runs and the probes 1 binned to it with LDS atomics, and writes the words back: the filter is read and written once per sub-batch,
// sequentially, and no probe leaves the CU.  What a probe reports is what k_probe_h2 + k_set_bits report: 1 = the bit was set before the
// sub-batch, 2 = it was clear and another probe of the sub-batch set it first (resolved by the collision table as before), 0 = this probe
// set it.  A probe that finds its bit set in LDS looks at the word in HBM — still the state before the sub-batch, the range is written
// back after its probes — to tell 1 from 2.  Neighbouring ranges may share a word: words that are not wholly inside the range are
// written back with atomicOr (bits are only ever set), the others with plain 16-byte stores.
constexpr uint32_t SW_WORDS = 16384;          // 64 KB of filter per round (+ 2 x 4 KB of marks: met, touched): two workgroups per CU (160 KB of LDS)
constexpr uint32_t SW_MARKS = 1024;           // words of collision marks, addressed by the low bits of the bit index
__device__ __forceinline__ uint32_t sw_digit(uint64_t x, const GrIdxDev &ix) {
    return min((uint32_t)min(__umul64hi(x, ix.mul), (uint64_t)0xFFFFFFFFull), ix.top);
}
// first index (relative to ix.lo) of fine bucket c
__device__ __forceinline__ uint64_t sw_first(uint32_t c, const GrIdxDev &ix, uint64_t span, uint32_t T) {
    if (c == 0u) return 0ull;
    if (c > ix.top) return span;
    uint64_t e = ((uint64_t)c * span) >> T;
    while (e > 0ull && sw_digit(e - 1ull, ix) >= c) --e;
    while (e < span && sw_digit(e, ix) < c) ++e;
    return e;
}
// (round 5) 4 = set before the sub-batch AND another probe of the sub-batch asked (or may have asked: marks go by the low bits of the index)
// for the same bit — irrelevant to the Bloom filter, but the counting filter has the same index: a run none of whose probes met anybody has its
// counters to itself and reads them with plain loads instead of claiming them (k_probe_h2).
// what a probe reports (st0 / st1): 0 it set the bit and nobody else asked for it; 1 set before the sub-batch; 2 clear before, set by another
// probe of the sub-batch that got there first; 3 it set the bit and another probe MAY have met it there (marks are per low bits of the index:
// a probe of another bit that shares them is told so too, and finds no entry in the collision table) — the probes with 2 or 3 are the ones
// the first-setter arbitration has to look at, everybody else is done
template <int TPB>
__global__ void __launch_bounds__(TPB, 4) k_sweep_bits(uint32_t *words, GrIdxDev ix, uint64_t span, uint32_t T, const uint64_t *__restrict__ uniq,
                                                    const uint32_t *__restrict__ brun, const uint32_t *__restrict__ bnr,
                                                    const uint64_t *__restrict__ k1, const uint32_t *__restrict__ v1, const uint32_t *__restrict__ bstart1,
                                                    uint8_t *st0, uint8_t *st1, uint32_t SWW /* words of filter per round: dynamic LDS = (SWW + SW_MARKS) words */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sw_lds[];
    uint32_t *s_w = sw_lds, *s_c = sw_lds + SWW, *s_t = sw_lds + SWW + SW_MARKS;      // filter words of the round; "met" marks; "touched" marks of bits that were set before
    __shared__ uint32_t s_any;
    const uint32_t c = blockIdx.x;
    const uint32_t r0 = brun[c], nr = bnr[c], p0 = bstart1[c], p1 = bstart1[c + 1u];
    if (nr == 0u && p0 == p1) return;
    const uint64_t x_lo = sw_first(c, ix, span, T), x_hi = sw_first(c + 1u, ix, span, T);
    const uint64_t w_lo = (x_lo >> 5) & ~3ull, w_hi = (x_hi + 31ull) >> 5;          // words [w_lo, w_hi), from a 16-byte boundary
    const uint64_t in_lo = (x_lo + 31ull) >> 5, in_hi = x_hi >> 5;                  // words [in_lo, in_hi) hold bits of this range only
    // the range's probes: bit offsets from the range's first loaded word, into registers once and on their way while the first round's words are
    // loaded (a bucket has at most GR_TILE runs: every probe 0, and the first KEEP x TPB probes 1; a bin far above the average walks the rest from memory)
    constexpr uint32_t KEEP = GR_TILE / TPB;
    constexpr uint32_t NONE = ~0u;
    static_assert(KEEP <= 16, "one mask bit per remembered probe");
    const uint64_t bit0 = w_lo << 5;
    uint32_t i0[KEEP], i1[KEEP];
    {
        uint64_t h[KEEP];
#pragma unroll
        for (uint32_t it = 0; it < KEEP; ++it) { const uint32_t r = it * TPB + threadIdx.x; h[it] = r < nr ? uniq[r0 + r] : 0ull; }
#pragma unroll
        for (uint32_t it = 0; it < KEEP; ++it) { const uint32_t r = it * TPB + threadIdx.x; i0[it] = r < nr ? (uint32_t)(index_of(h[it], ix.mod) - ix.lo - bit0) : NONE; }
#pragma unroll
        for (uint32_t it = 0; it < KEEP; ++it) { const uint32_t j = p0 + it * TPB + threadIdx.x; h[it] = j < p1 ? k1[j] : 0ull; }
#pragma unroll
        for (uint32_t it = 0; it < KEEP; ++it) { const uint32_t j = p0 + it * TPB + threadIdx.x; i1[it] = j < p1 ? (uint32_t)(index_of(h[it], ix.mod) - ix.lo - bit0) : NONE; }
    }
    for (uint64_t wb = w_lo; wb < w_hi; wb += SWW) {
        const uint32_t cnt = (uint32_t)min((uint64_t)SWW, w_hi - wb);
        const uint32_t wrel = (uint32_t)(wb - w_lo);           // this round's first word, counted from the range's
        for (uint32_t i = threadIdx.x * 4u; i < cnt; i += TPB * 4u) {
            if (i + 4u <= cnt) *reinterpret_cast<uint4 *>(s_w + i) = *reinterpret_cast<const uint4 *>(words + wb + i);
            else for (uint32_t q = i; q < cnt; ++q) s_w[q] = words[wb + q];
        }
        for (uint32_t i = threadIdx.x; i < 2u * SW_MARKS; i += TPB) s_c[i] = 0u;          // (s_t lies behind s_c)
        if (threadIdx.x == 0) s_any = 0u;
        __syncthreads();
        // one probe (rel: its bit, counted from the range's first loaded word): test-and-set in LDS; a bit found set is looked up in HBM (still the
        // state before the sub-batch).  Returns the report: -1 another round's, 0 set by this probe — it may hear of a collision after the barrier.
        auto probe = [&](uint32_t rel) -> int {
            const uint32_t w = (rel >> 5) - wrel;
            if (rel == NONE || w >= cnt) return -1;
            const uint32_t m = 1u << (rel & 31u);
            if (!(atomicOr(&s_w[w], m) & m)) return 0;
            if (__hip_atomic_load(&words[w_lo + (rel >> 5)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & m) {
                // set before the sub-batch: nothing to arbitrate — but whether ANOTHER probe of the sub-batch asks for the same bit decides whether the
                // run may read its counter without claiming it (same index in the counting filter, k_probe_h2): the first such probe leaves a touch mark,
                // a later one finds it, marks the bit as met and reports 4; the first one hears of it behind the barrier
                if (atomicOr(&s_t[w & (SW_MARKS - 1u)], m) & m) { atomicOr(&s_c[w & (SW_MARKS - 1u)], m); s_any = 1u; return 4; }
                return 1;
            }
            atomicOr(&s_c[w & (SW_MARKS - 1u)], m); s_any = 1u;
            return 2;
        };
        uint32_t setmask = 0, premask = 0;                     // bit it: probe 0 of iteration it set its bit in this round (premask: found it set before the sub-batch); bit 16 + it: probe 1
#pragma unroll
        for (uint32_t it = 0; it < KEEP; ++it) {
            const int s = probe(i0[it]);
            if (s >= 0) st0[r0 + it * TPB + threadIdx.x] = (uint8_t)s;
            if (s == 0) setmask |= 1u << it;
            if (s == 1) premask |= 1u << it;
        }
#pragma unroll
        for (uint32_t it = 0; it < KEEP; ++it) {
            const int s = probe(i1[it]);
            if (s > 0) st1[v1[p0 + it * TPB + threadIdx.x]] = (uint8_t)s;
            if (s == 0) setmask |= 1u << (16u + it);
            if (s == 1) premask |= 1u << (16u + it);
        }
        for (uint32_t j = p0 + KEEP * TPB + threadIdx.x; j < p1; j += TPB) {       // (a bin far above the average)
            const int s = probe((uint32_t)(index_of(k1[j], ix.mod) - ix.lo - bit0));
            if (s > 0) st1[v1[j]] = (uint8_t)s;
        }
        __syncthreads();
        if (s_any) {                         // the probes that set a bit somebody else then met them on are told so
            auto marked = [&](uint32_t rel) { return ((s_c[((rel >> 5) - wrel) & (SW_MARKS - 1u)] >> (rel & 31u)) & 1u) != 0u; };
#pragma unroll
            for (uint32_t it = 0; it < KEEP; ++it) {
                if (((setmask >> it) & 1u) && marked(i0[it])) st0[r0 + it * TPB + threadIdx.x] = 3;
                if (((premask >> it) & 1u) && marked(i0[it])) st0[r0 + it * TPB + threadIdx.x] = 4;
                if (((setmask >> (16u + it)) & 1u) && marked(i1[it])) st1[v1[p0 + it * TPB + threadIdx.x]] = 3;
                if (((premask >> (16u + it)) & 1u) && marked(i1[it])) st1[v1[p0 + it * TPB + threadIdx.x]] = 4;
            }
            for (uint32_t j = p0 + KEEP * TPB + threadIdx.x; j < p1; j += TPB) {
                const uint32_t rel = (uint32_t)(index_of(k1[j], ix.mod) - ix.lo - bit0);
                if ((rel >> 5) - wrel < cnt && marked(rel)) {
                    const uint32_t d = v1[j];
                    if (st1[d] == 0) st1[d] = 3; else if (st1[d] == 1) st1[d] = 4;
                }
            }
        }
        for (uint32_t i = threadIdx.x * 4u; i < cnt; i += TPB * 4u) {
            const uint64_t gw = wb + i;
            if (gw >= in_lo && gw + 4ull <= in_hi) *reinterpret_cast<uint4 *>(words + gw) = *reinterpret_cast<const uint4 *>(s_w + i);
            else
                for (uint32_t q = 0; q < 4u && i + q < cnt; ++q) {
                    const uint64_t g = gw + q;
                    if (g >= in_lo && g < in_hi) words[g] = s_w[i + q];
                    else if ((g << 5) < x_hi && ((g + 1ull) << 5) > x_lo) atomicOr(&words[g], s_w[i + q]);     // shared with a neighbour: what was loaded was set, what is new is ours
                }
        }
        __syncthreads();
    }
}
// the runs of oversized buckets (appended by k_group_big; none in a warm steady state) are not contiguous per bucket: their probe 0 looks
// at HBM before the sweep and sets its bit after it
__global__ void k_sw_big_pre(const uint32_t *__restrict__ words, Mod mod, uint64_t lo, const uint64_t *__restrict__ uniq, uint32_t d0, uint32_t D, uint8_t *__restrict__ st0) {
    const uint32_t d = d0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const uint64_t i0 = index_of(uniq[d], mod) - lo;
    st0[d] = (words[i0 >> 5] >> (uint32_t)(i0 & 31ull)) & 1u;
}
__global__ void k_sw_big_set(uint32_t *__restrict__ words, Mod mod, uint64_t lo, const uint64_t *__restrict__ uniq, uint32_t d0, uint32_t D, uint8_t *__restrict__ st0) {
    const uint32_t d = d0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D || st0[d]) return;
    const uint64_t i0 = index_of(uniq[d], mod) - lo;
    const uint32_t m = 1u << (uint32_t)(i0 & 31ull);
    if (atomicOr(&words[i0 >> 5], m) & m) st0[d] = 2;
}
// INVARIANT of the swept stage: from the launch of k_sweep_bits to its end NOTHING else writes `words` — no other stream, no producer of
// the next sub-batch.  A round loads a range's words into LDS, decides "set before the sub-batch" against the words still in HBM, and writes
// the interior words back with plain 16-byte stores: a bit another writer set in between would be lost, and the 1-versus-2 report would be
// wrong.  The callers (run_core, rb_graph.hip: the consumer stream owns dbgbf, the producer touches scratch and rpkbf only; the sharded
// engine does not sweep) hold it; a future overlap that writes dbgbf beside the sweep has to switch the write-back to atomicOr of
// (s_w & ~loaded) first.
size_t sweep_temp_bytes(size_t D, uint32_t T) { return group_plan(D, 64, 0, 0, (int)T).total; }
void sweep_bits_device(uint32_t *words, GrIdx idx, uint32_t T, uint64_t kmul, const uint64_t *uniq, uint32_t D, uint32_t n_main, const uint32_t *brun,
                       const uint32_t *bnr, uint64_t *keys_a, uint32_t *vals_a, uint64_t *keys_b, uint32_t *vals_b, void *temp, size_t temp_bytes,
                       uint8_t *st0, uint8_t *st1, hipStream_t st) {
    RB_REQUIRE(D > 0 && T > 0 && idx.span > (1ull << T) && (idx.span >> T) < (1ull << 31), "sweep_bits_device: nothing to sweep by / ranges of 2^31 bits and more");
    const GroupPlan P = group_plan(D, 64, 0, 0, (int)T);
    RB_REQUIRE(P.T == T && temp_bytes >= P.total, "sweep_bits_device: plan / temp mismatch");
    RB_HIP(hipMemsetAsync(st1, 0, D, st));
    RB_REQUIRE(kmul != 0, "sweep_bits_device: kmul");
    const uint64_t *k1 = nullptr;
    const uint32_t *v1 = nullptr;            // the (h1, run) records are made by the first binning pass itself, from the runs' hashes
    const uint32_t *bstart1 = P.tpb == 512u ? partition_records<512>(P, keys_a, vals_a, keys_b, vals_b, static_cast<char *>(temp), st, nullptr, idx, &k1, &v1, uniq, kmul)
                                            : partition_records<256>(P, keys_a, vals_a, keys_b, vals_b, static_cast<char *>(temp), st, nullptr, idx, &k1, &v1, uniq, kmul);
    GrIdxDev ix = gr_idx_dev(P, idx);
    const uint32_t n_big = D - std::min(n_main, D);
    if (n_big) hipLaunchKernelGGL(k_sw_big_pre, dim3((n_big + 255u) / 256u), dim3(256), 0, st, (const uint32_t *)words, idx.mod, idx.lo, uniq, n_main, D, st0);
    // a round holds a whole range where it fits 64 KB; LDS is sized to the range (two workgroups per CU either way: 128 VGPRs)
    const uint64_t range_words = (idx.span >> T) / 32 + 16;
    const uint32_t sww = (uint32_t)std::min<uint64_t>(SW_WORDS, (range_words + 1023) / 1024 * 1024);
    RB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep_bits<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((SW_WORDS + 2 * SW_MARKS) * 4)));
    hipLaunchKernelGGL(k_sweep_bits<512>, dim3(P.nbuckets), dim3(512), (sww + 2 * SW_MARKS) * 4, st, words, ix, idx.span, T, uniq, brun, bnr, k1, v1, bstart1, st0, st1, sww);
    if (n_big) hipLaunchKernelGGL(k_sw_big_set, dim3((n_big + 255u) / 256u), dim3(256), 0, st, words, idx.mod, idx.lo, uniq, n_main, D, st0);
    RB_HIP(hipGetLastError());
}

}  // namespace rb
