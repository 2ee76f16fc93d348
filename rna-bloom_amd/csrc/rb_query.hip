// rb_query.hip — everything that only READS the graph (plus the order-independent bit-set calls): lookups and counts of base hashes,
// getKmers, the count profile of resident reads, neighbours, and the three traversals (maximum-coverage walk, greedy extension with
// lookahead, naive extension) on one GPU and — with replayed answers — on a sharded graph.  Kernels first, then their C-ABI entry points.
// (Split out of rb_graph.hip in round 5; the insert pipeline is rb_graph.hip, life cycle / insert entry points / filter files rb_capi.hip.)
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <new>
#include <vector>

#include <functional>
#include <string>
#include <thread>

#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include "rb_pipeline.hpp"

using namespace rb;

// ---- direct (order-independent) bit-filter ops and queries on arrays of base hashes ----
__global__ void rb::k_bits_add(uint32_t *bits, Mod mod, int num_hash, uint64_t kmul, const uint64_t *__restrict__ h0, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int j = 0; j < num_hash; ++j) bit_set(bits, index_of(multi_hash(h0[i], (uint32_t)j, kmul), mod));
}

namespace {

// CountingBloomFilter.getCount(long[]) :235-251 (zero check inside the h>=1 loop)
__device__ __forceinline__ float cbf_get_count(const uint8_t *cbf, const Mod &mod, int num_hash, uint64_t kmul, uint64_t h0) {
    uint32_t mn = cbf[index_of(h0, mod)];
    for (int j = 1; j < num_hash; ++j) {
        uint32_t c = cbf[index_of(multi_hash(h0, (uint32_t)j, kmul), mod)];
        if (c < mn) mn = c;
        if (mn == 0u) return 0.0f;
    }
    return minifloat_to_float(mn);
}
__device__ __forceinline__ float graph_count(const FilterView &fv, uint64_t h0) {   // BloomFilterDeBruijnGraph.java:562-570
    if (!bits_lookup(fv.dbg, fv.dbg_mod, fv.dbg_h, fv.kmul, h0)) return 0.0f;
    return cbf_get_count(fv.cbf, fv.cbf_mod, fv.cbf_h, fv.kmul, h0) + 1.0f;
}
__global__ void k_bits_lookup(const uint32_t *bits, Mod mod, int num_hash, uint64_t kmul, const uint64_t *__restrict__ h0, size_t n, uint8_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bits_lookup(bits, mod, num_hash, kmul, h0[i]) ? 1 : 0;
}
__global__ void k_graph_count(FilterView fv, const uint64_t *__restrict__ h0, size_t n, float *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = graph_count(fv, h0[i]);
}
__global__ void k_cbf_count(FilterView fv, const uint64_t *__restrict__ h0, size_t n, float *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cbf_get_count(fv.cbf, fv.cbf_mod, fv.cbf_h, fv.kmul, h0[i]);
}

// getKmers: hash EVERY window of a read (unusable bases hash as seed 0, exactly like seedTab's
// zero rows, R/bloom/hash/NTHash.java:133-166), count = 0 for windows containing an unusable base
// (R/bloom/hash/CanonicalHashFunction.java:46-78).  One thread per 32-window chunk.
// HASH_ONLY (a shard of a distributed graph: the counts come from a query exchange): out_c = 1 where the window is usable
template <bool HASH_ONLY>
__global__ void k_get_kmers(FilterView fv, int stranded, const uint64_t *__restrict__ codes,
                            const uint32_t *__restrict__ valid, const uint32_t *__restrict__ rnz /* reverse-strand seed non-zero (NTHash.java:30: ch & 7), or null */,
                            const uint32_t *__restrict__ word_read,
                            const uint32_t *__restrict__ woff, const uint32_t *__restrict__ len,
                            int64_t n_words, int k, const int64_t *__restrict__ koff,
                            uint64_t *__restrict__ out_f, uint64_t *__restrict__ out_r, float *__restrict__ out_c) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const uint32_t r = word_read[w], wr = woff[r], L = len[r];
    const uint32_t b0 = (uint32_t)(w - wr) * 32u, uk = (uint32_t)k;
    if ((uint64_t)b0 + uk > L) return;
    const uint64_t bend64 = (uint64_t)b0 + 32u + uk - 1u;
    const uint32_t bend = bend64 < L ? (uint32_t)bend64 : L;
    const uint64_t *cw = codes + wr;
    const uint32_t *vw = valid + wr;
    const uint32_t *zw = rnz ? rnz + wr : vw;      // a base outside ACGTU may still carry a reverse-strand seed (K M S W Y I E ...)
    uint64_t f = 0, rv = 0;
    uint32_t filled = 0, run = 0;
    for (uint32_t b = b0; b < bend; ++b) {
        const bool ok = (vw[b >> 5] >> (b & 31u)) & 1u, rok = (zw[b >> 5] >> (b & 31u)) & 1u;
        const uint64_t s_in = ok ? seed_of((uint32_t)(cw[b >> 5] >> (2u * (b & 31u))) & 3u) : 0ull;
        const uint64_t sc_in = rok ? seed_of(3u - ((uint32_t)(cw[b >> 5] >> (2u * (b & 31u))) & 3u)) : 0ull;
        run = ok ? run + 1u : 0u;
        if (filled < uk) { f = rotl(f, 1) ^ s_in; rv ^= rotl(sc_in, filled); ++filled; }
        else {
            const uint32_t bo = b - uk;
            const bool oko = (vw[bo >> 5] >> (bo & 31u)) & 1u, roko = (zw[bo >> 5] >> (bo & 31u)) & 1u;
            const uint32_t oc = (uint32_t)(cw[bo >> 5] >> (2u * (bo & 31u))) & 3u;
            const uint64_t s_out = oko ? seed_of(oc) : 0ull, sc_out = roko ? seed_of(3u - oc) : 0ull;
            f = rotl(f, 1) ^ rotl(s_out, uk) ^ s_in;
            rv = rotr(rv, 1) ^ rotr(sc_out, 1) ^ rotl(sc_in, uk - 1u);
        }
        if (filled >= uk) {
            const uint32_t p = b - uk + 1u;
            const int64_t o = koff[r] + p;
            const uint64_t base = stranded ? f : canonical(f, rv);
            out_f[o] = f;
            out_r[o] = stranded ? 0ull : rv;
            out_c[o] = run >= uk ? (HASH_ONLY ? 1.0f : graph_count(fv, base)) : 0.0f;
        }
    }
}

// getKmers' counts for reads that already sit in HBM (a resident rb_batch): count[row(r) + p] = graph.getCount of window p of read r,
// 0 where the window holds an unusable base (R/bloom/hash/CanonicalHashFunction.java:46-78) — what stage 2 reads first of every
// read (R/RNABloom.java:1984, 2097-2114).  One thread per 32-window word as in k_get_kmers; the hashes are rolled, nothing but the
// counts is written.  Written for memory-level parallelism: a lane collects four usable windows, computes all their filter
// indices, issues the 8 Bloom-bit loads, then the 8 counter loads, and only then combines them (graph_count per window would be
// four dependent round trips each).  koff == nullptr: rows of `stride` counts (uniform reads).
__global__ void __launch_bounds__(256) k_batch_counts(FilterView fv, int stranded, const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                                                      const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ woff,
                                                      const uint32_t *__restrict__ len, int64_t w_first, int64_t n_words, uint32_t r_first, int k,
                                                      const int64_t *__restrict__ koff, int64_t stride, int64_t row_base, float *__restrict__ out_c) {
    const int64_t w = w_first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= w_first + n_words) return;
    const uint32_t r = word_read[w], wr = woff[r], L = len[r];
    const uint32_t b0 = (uint32_t)(w - wr) * 32u, uk = (uint32_t)k;
    if ((uint64_t)b0 + uk > L) return;
    const uint64_t bend64 = (uint64_t)b0 + 32u + uk - 1u;
    const uint32_t bend = bend64 < L ? (uint32_t)bend64 : L;
    const uint64_t *cw = codes + wr;
    const uint32_t *vw = valid + wr;
    const int64_t row = (koff ? koff[r - r_first] : (int64_t)(r - r_first) * stride) - row_base;
    const bool h2 = fv.dbg_h == 2 && fv.cbf_h == 2;
    uint64_t f = 0, rv = 0, pend_h[4];
    uint32_t filled = 0, run = 0, pend_p[4], n_pend = 0;
    auto flush = [&]() {
        uint64_t bi[4][2], ci[4][2];
        uint32_t bw[4][2], cb[4][2];
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            const uint64_t h0 = q < n_pend ? pend_h[q] : pend_h[0], h1 = multi_hash(h0, 1u, fv.kmul);
            bi[q][0] = index_of(h0, fv.dbg_mod); bi[q][1] = index_of(h1, fv.dbg_mod);
            ci[q][0] = index_of(h0, fv.cbf_mod); ci[q][1] = index_of(h1, fv.cbf_mod);
        }
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) { bw[q][0] = fv.dbg[bi[q][0] >> 5]; bw[q][1] = fv.dbg[bi[q][1] >> 5]; }
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) { cb[q][0] = fv.cbf[ci[q][0]]; cb[q][1] = fv.cbf[ci[q][1]]; }
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            if (q >= n_pend) break;
            const bool in = ((bw[q][0] >> (uint32_t)(bi[q][0] & 31u)) & (bw[q][1] >> (uint32_t)(bi[q][1] & 31u)) & 1u) != 0u;
            const uint32_t mn = cb[q][0] < cb[q][1] ? cb[q][0] : cb[q][1];
            out_c[row + pend_p[q]] = in ? minifloat_to_float(mn) + 1.0f : 0.0f;
        }
        n_pend = 0;
    };
    for (uint32_t b = b0; b < bend; ++b) {
        const bool ok = (vw[b >> 5] >> (b & 31u)) & 1u;
        const uint32_t ic = (uint32_t)(cw[b >> 5] >> (2u * (b & 31u))) & 3u;
        const uint64_t s_in = ok ? seed_of(ic) : 0ull, sc_in = ok ? seed_of(3u - ic) : 0ull;
        run = ok ? run + 1u : 0u;
        if (filled < uk) { f = rotl(f, 1) ^ s_in; rv ^= rotl(sc_in, filled); ++filled; }
        else {
            const uint32_t bo = b - uk;
            const bool oko = (vw[bo >> 5] >> (bo & 31u)) & 1u;
            const uint32_t oc = (uint32_t)(cw[bo >> 5] >> (2u * (bo & 31u))) & 3u;
            const uint64_t s_out = oko ? seed_of(oc) : 0ull, sc_out = oko ? seed_of(3u - oc) : 0ull;
            f = rotl(f, 1) ^ rotl(s_out, uk) ^ s_in;
            rv = rotr(rv, 1) ^ rotr(sc_out, 1) ^ rotl(sc_in, uk - 1u);
        }
        if (filled >= uk) {
            const uint32_t p = b - uk + 1u;
            const uint64_t base = stranded ? f : canonical(f, rv);
            if (run < uk) out_c[row + p] = 0.0f;
            else if (!h2) out_c[row + p] = graph_count(fv, base);
            else {
                pend_h[n_pend] = base; pend_p[n_pend] = p;
                if (++n_pend == 4u) flush();
            }
        }
    }
    if (n_pend) flush();
}

// Kmer.getSuccessors / getPredecessors: the four neighbours' hashes and counts
// (R/bloom/hash/{,Canonical}{Successors,Predecessors}NTHashIterator.java; R/graph/Kmer.java:210-255)
__device__ __forceinline__ uint32_t code_of_char(uint32_t ch) {
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                  case 'T': case 't': case 'U': case 'u': return 3; default: return 4; }
}
template <bool HASH_ONLY>
__global__ void k_neighbors(FilterView fv, int stranded, int k, int direction, const uint64_t *__restrict__ f,
                            const uint64_t *__restrict__ r, const uint8_t *__restrict__ ch, size_t n,
                            uint64_t *__restrict__ f4, uint64_t *__restrict__ r4, float *__restrict__ c4) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 4) return;
    const size_t i = t >> 2;
    const uint32_t in = (uint32_t)(t & 3u), uk = (uint32_t)k;
    const uint32_t oc = code_of_char(ch[i]);
    const uint64_t s_out = oc < 4 ? seed_of(oc) : 0ull, sc_out = oc < 4 ? seed_of(3u - oc) : 0ull;
    uint64_t nf, nr = 0;
    if (direction == 0) {
        nf = rotl(f[i], 1) ^ rotl(s_out, uk) ^ seed_of(in);
        if (!stranded) nr = rotr(r[i], 1) ^ rotr(sc_out, 1) ^ rotl(seed_of(3u - in), uk - 1u);
    } else if (direction == 1) {
        nf = rotr(f[i], 1) ^ rotr(s_out, 1) ^ rotl(seed_of(in), uk - 1u);
        if (!stranded) nr = rotl(r[i], 1) ^ rotl(sc_out, uk) ^ seed_of(3u - in);
    } else if (direction == 2) {   // left variants: replace the FIRST base ({,Canonical}LeftVariantsNTHashIterator.java)
        nf = f[i] ^ rotl(s_out, uk - 1u) ^ rotl(seed_of(in), uk - 1u);
        if (!stranded) nr = r[i] ^ sc_out ^ seed_of(3u - in);
    } else {                       // right variants: replace the LAST base ({,Canonical}RightVariantsNTHashIterator.java)
        nf = f[i] ^ s_out ^ seed_of(in);
        if (!stranded) nr = r[i] ^ rotl(sc_out, uk - 1u) ^ rotl(seed_of(3u - in), uk - 1u);
    }
    f4[t] = nf;
    if (r4) r4[t] = nr;
    c4[t] = HASH_ONLY ? 0.0f : graph_count(fv, stranded ? nf : smin(nf, nr));
}

// ---- where a traversal kernel takes graph.getCount from ----
// DirectCounts: the graph's own filters (one GPU holds them).  ReplayCounts: a SHARDED graph (rb_shard_trav_*): the counts live
// on other ranks, so a walk runs until it needs a count it has not been told, files the request (the four neighbours of a k-mer
// go out together) and suspends at the START of its current step; after the exchange round (rb_shard_query_* protocol) the
// answers are in the walk's cache and the same kernel replays the step from its start — every count it asked before is now a
// cache hit, so it gets exactly as far as the next unknown neighbourhood.  One body per traversal, two count sources: the
// sharded walks cannot drift from the single-GPU ones.  A finished step empties the cache.
struct TravArrays {
    uint64_t *f, *r;            // per walk: hashes of the k-mer it stands on when suspended
    int32_t *len;               // appended k-mers so far
    uint8_t *phase;             // 0 fresh, 1 suspended, 2 finished
    uint64_t *ckey; float *cval; uint32_t *cn; uint32_t ccap;       // answers [walk][ccap]
    uint8_t *over;              // the answers of one step did not fit ccap
    uint64_t *req; uint32_t *req_walk; uint32_t *ctr; uint32_t req_cap;   // this round's requests; ctr[0] = requests, ctr[1] = walks suspended
};
struct DirectCounts {
    static constexpr bool kReplay = false;
    FilterView fv;
    struct Walk {
        const FilterView *fv;
        __device__ __forceinline__ bool get(uint64_t h0, float &c) const { c = graph_count(*fv, h0); return true; }
        __device__ __forceinline__ void step_done() const {}
    };
    __device__ __forceinline__ Walk walk(size_t) const { return Walk{&fv}; }
    __device__ __forceinline__ TravArrays arrays() const { return TravArrays{}; }
};
struct ReplayCounts {
    static constexpr bool kReplay = true;
    TravArrays t;
    struct Walk {
        const uint64_t *key; const float *val; uint32_t *cn; uint32_t n; uint64_t *req; uint32_t *req_walk; uint32_t *ctr; uint32_t req_cap, id;
        __device__ __forceinline__ bool get(uint64_t h0, float &c) const {
            for (uint32_t q = 0; q < n; ++q) if (key[q] == h0) { c = val[q]; return true; }
            const uint32_t p = atomicAdd(&ctr[0], 1u);
            if (p < req_cap) { req[p] = h0; req_walk[p] = id; }
            c = 0.0f;
            return false;
        }
        __device__ __forceinline__ void step_done() { *cn = 0u; n = 0u; }
    };
    __device__ __forceinline__ Walk walk(size_t i) const {
        return Walk{t.ckey + i * t.ccap, t.cval + i * t.ccap, t.cn + i, min(t.cn[i], t.ccap), t.req, t.req_walk, t.ctr, t.req_cap, (uint32_t)i};
    }
    __device__ __forceinline__ TravArrays arrays() const { return t; }
};
constexpr uint8_t WALK_REASON_TOO_WIDE = 8;   // sharded graphs only: one step asked for more counts than the walk's answer cache holds
// the answers of one exchange round go into the caches of the walks that asked
// (gate: the extra BloomFilter of the `bf` variants, looked up by its owners in the same round — a k-mer that fails it counts 0:
// Kmer.getSuccessors(k, numHash, graph, bf) skips it before its count is read, and every caller's threshold is >= 1)
__global__ void k_trav_absorb(TravArrays t, const float *__restrict__ ans, const uint8_t *__restrict__ gate, uint32_t n_req) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_req) return;
    const uint32_t i = t.req_walk[j];
    const uint32_t p = atomicAdd(&t.cn[i], 1u);
    if (p < t.ccap) { t.ckey[(size_t)i * t.ccap + p] = t.req[j]; t.cval[(size_t)i * t.ccap + p] = (gate && !gate[j]) ? 0.0f : ans[j]; }
    else t.over[i] = 1;
}

// ---- greedy maximum-coverage walk: the loop around Kmer.getMaxCovSuccessor / getMaxCovPredecessor ----
// One lane per walk (R/util/GraphUtils.java:1591-1675 getMaxCoveragePath runs two of them; :1906-1990 the
// lookahead-free part of greedyExtend*).  Per step: the 4 neighbours in order A,C,G,T
// ({,Canonical}{Successors,Predecessors}NTHashIterator), graph.getCount of each, the FIRST strict maximum
// among those with count >= min_cov (R/graph/Kmer.java:301-355).  The walk ends when there is none (reason 0),
// when the best neighbour IS the target k-mer (1; not appended), when it is a k-mer the walk already appended
// (2; not appended; Kmer.equals compares bytes: hashes are compared first, then the bases), or after `bound`
// appended k-mers (3); a seed with a base outside ACGTU ends at once (4).
// seq[i]: for a right walk the seed's bases followed by the appended ones; for a left walk the seed's bases
// REVERSED followed by the prepended ones — either way k-mer number j of the walk (0-based) is seq[j+1 .. j+k].
template <class SRC>
__global__ void k_walk_max_cov(SRC src, int stranded, int k, int direction, const uint8_t *__restrict__ seeds,
                               const uint8_t *__restrict__ targets, size_t n, int bound, float min_cov,
                               uint8_t *__restrict__ seq, uint8_t *__restrict__ out_b, uint64_t *__restrict__ out_f, uint64_t *__restrict__ out_r,
                               float *__restrict__ out_c, int32_t *__restrict__ out_len, uint8_t *__restrict__ out_reason) {
    __shared__ uint32_t s_seen[32][64];                   // [word][lane]: bitmap of the hashes the lane's walk appended
    for (int q = 0; q < 32; ++q) s_seen[q][threadIdx.x] = 0u;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TravArrays ta = src.arrays();
    if (SRC::kReplay && ta.phase[i] == 2) return;
    if (SRC::kReplay && ta.over[i]) { out_len[i] = ta.len[i]; out_reason[i] = WALK_REASON_TOO_WIDE; ta.phase[i] = 2; return; }
    const bool resume = SRC::kReplay && ta.phase[i] == 1;
    auto w = src.walk(i);
    const uint32_t uk = (uint32_t)k;
    const size_t stride = (size_t)k + (size_t)bound;
    uint8_t *sq = seq + i * stride;
    uint64_t *pf = out_f + i * (size_t)bound, *pr = out_r + i * (size_t)bound;
    float *pc = out_c + i * (size_t)bound;
    // hashes of the seed (and of the target) from scratch: NTHash.java:332-337, 367-373
    auto hash_kmer = [&](const uint8_t *b, uint64_t &f, uint64_t &r) -> bool {
        f = 0; r = 0;
        for (uint32_t q = 0; q < uk; ++q) {
            const uint32_t c = code_of_char(b[q]);
            if (c > 3u) return false;
            f = rotl(f, 1) ^ seed_of(c);
            r ^= rotl(seed_of(3u - c), q);
        }
        return true;
    };
    uint64_t f, r, tf = 0, tr = 0;
    const uint8_t *sb = seeds + i * (size_t)k;
    if (!hash_kmer(sb, f, r)) { out_len[i] = 0; out_reason[i] = 4; if (SRC::kReplay) ta.phase[i] = 2; return; }
    const uint8_t *tb = targets ? targets + i * (size_t)k : nullptr;
    bool has_target = tb && hash_kmer(tb, tf, tr);
    const uint8_t acgt[4] = {'A', 'C', 'G', 'T'};
    int len = 0;
    uint8_t reason = 3;
    if (resume) {                                             // a suspended walk: where it stood, and the bitmap of what it appended
        f = ta.f[i]; r = ta.r[i]; len = ta.len[i];
        for (int j = 0; j < len; ++j) { const uint32_t hb = (uint32_t)((pf[j] * 0x9E3779B97F4A7C15ull) >> 54); s_seen[hb >> 5][threadIdx.x] |= 1u << (hb & 31u); }
    } else
        for (uint32_t q = 0; q < uk; ++q) sq[q] = (direction == 0) ? sb[q] : sb[uk - 1u - q];
    while (len < bound) {
        const uint32_t oc = code_of_char(sq[len]);            // base leaving: first base (right walk) / last base (left walk)
        const uint64_t s_out = seed_of(oc), sc_out = seed_of(3u - oc);
        float best_c = -1.0f;
        uint64_t best_f = 0, best_r = 0;
        uint32_t best_in = 0;
        bool miss = false;
        for (uint32_t in = 0; in < 4u; ++in) {
            uint64_t nf, nr = 0;
            if (direction == 0) {
                nf = rotl(f, 1) ^ rotl(s_out, uk) ^ seed_of(in);
                if (!stranded) nr = rotr(r, 1) ^ rotr(sc_out, 1) ^ rotl(seed_of(3u - in), uk - 1u);
            } else {
                nf = rotr(f, 1) ^ rotr(s_out, 1) ^ rotl(seed_of(in), uk - 1u);
                if (!stranded) nr = rotl(r, 1) ^ rotl(sc_out, uk) ^ seed_of(3u - in);
            }
            float c;
            if (!w.get(stranded ? nf : smin(nf, nr), c)) { miss = true; continue; }
            if (c >= min_cov && c > best_c) { best_c = c; best_f = nf; best_r = nr; best_in = in; }
        }
        if (SRC::kReplay && miss) { ta.f[i] = f; ta.r[i] = r; ta.len[i] = len; ta.phase[i] = 1; atomicAdd(&ta.ctr[1], 1u); return; }
        if (best_c < 0.0f) { reason = 0; break; }
        const uint8_t nb = acgt[best_in];
        // bases of the candidate: seq[len+1 .. len+k-1] + nb
        auto same_as = [&](const uint8_t *other_fwd_or_rev, bool other_is_seq) -> bool {   // other: k bases in walk orientation
            for (uint32_t q = 0; q + 1u < uk; ++q) if (code_of_char(sq[(size_t)len + 1u + q]) != code_of_char(other_fwd_or_rev[q])) return false;
            (void)other_is_seq;
            return code_of_char(other_fwd_or_rev[uk - 1u]) == best_in;
        };
        if (has_target && best_f == tf) {
            bool eq = true;                                    // target bases are given left to right
            for (uint32_t q = 0; q < uk && eq; ++q) {
                const uint8_t cb = (q + 1u < uk) ? sq[(size_t)len + 1u + q] : nb;          // candidate in walk orientation
                const uint8_t tq = (direction == 0) ? tb[q] : tb[uk - 1u - q];
                eq = code_of_char(cb) == code_of_char(tq);
            }
            if (eq) { reason = 1; break; }
        }
        // has the walk appended this k-mer before?  A per-lane bitmap of the appended hashes (1024 bits in LDS) says "no"
        // for almost every step; only a set bit sends the lane through the list of its hashes (then bases)
        const uint32_t hb = (uint32_t)((best_f * 0x9E3779B97F4A7C15ull) >> 54);       // 10 bits
        bool seen = false;
        if ((s_seen[hb >> 5][threadIdx.x] >> (hb & 31u)) & 1u)
            for (int j = 0; j < len && !seen; ++j)
                if (pf[j] == best_f && same_as(sq + (size_t)j + 1u, true)) seen = true;
        if (seen) { reason = 2; break; }
        s_seen[hb >> 5][threadIdx.x] |= 1u << (hb & 31u);
        sq[(size_t)uk + (size_t)len] = nb;
        out_b[i * (size_t)bound + (size_t)len] = nb;
        pf[len] = best_f; pr[len] = best_r; pc[len] = best_c;
        f = best_f; r = best_r;
        ++len;
        w.step_done();
    }
    out_len[i] = len;
    out_reason[i] = reason;
    if (SRC::kReplay) ta.phase[i] = 2;
}

// ---- GraphUtils.naiveExtendRight / naiveExtendLeft (R/util/GraphUtils.java:6780-7112): extension through unbranched
// stretches.  One lane per walk.  Per step, with `best` = the k-mer the walk stands on (the seed at first):
//   * back-branch test (not in the NoBackChecks forms): any left (right walk) / right (left walk) variant of `best` —
//     the base about to leave replaced — with count >= 1 ends the walk (:6794-6799; Kmer.hasDepthLeft / hasDepthRight
//     never consult the graph and always answer true, R/graph/Kmer.java:407-486, so the variant's existence decides);
//   * neighbours with count >= minKmerCov: none ends the walk, exactly one is taken, two or more end it ("too many good
//     branches", :6805-6819 — again hasDepth* is always true);
//   * mode 0 (terminators, :6780-6833 / :6959-7012): a candidate that is one of the walk's terminator k-mers (every
//     k-mer of a per-walk sequence, Kmer.equals = same bases) or that the walk added before ends it, not added;
//     mode 1 (bounded, :6835-6886 / :7014-7065): added, then the walk ends once ++length > bound (bound + 1 k-mers);
//     mode 2 (NoBackChecks, :6888-6933 / :7067-7112): ends, not added, when the candidate equals the seed or the k-mer
//     added last; else as mode 1.
// reason: 0 no neighbour, 1 back branch, 2 several neighbours, 3 bound, 4 invalid seed, 5 terminator / used k-mer,
// 6 output capacity reached (mode 0 has no bound of its own), 7 the candidate repeats the seed / the last k-mer.
template <class SRC>
__global__ void k_naive_extend(SRC src, int stranded, int k, int direction, int mode, const uint8_t *__restrict__ seeds, size_t n,
                               int bound, int cap, float min_cov, const uint8_t *__restrict__ term_seq, const int64_t *__restrict__ term_off,
                               const uint64_t *__restrict__ term_f, const int64_t *__restrict__ term_koff,
                               uint8_t *__restrict__ seq, uint8_t *__restrict__ out_b, uint64_t *__restrict__ wf,
                               int32_t *__restrict__ out_len, uint8_t *__restrict__ out_reason) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TravArrays ta = src.arrays();
    if (SRC::kReplay && ta.phase[i] == 2) return;
    if (SRC::kReplay && ta.over[i]) { out_len[i] = ta.len[i]; out_reason[i] = WALK_REASON_TOO_WIDE; ta.phase[i] = 2; return; }
    const bool resume = SRC::kReplay && ta.phase[i] == 1;
    auto w = src.walk(i);
    const uint32_t uk = (uint32_t)k;
    const size_t stride = (size_t)k + (size_t)cap;
    uint8_t *sq = seq + i * stride;                       // walk orientation: seed (reversed for a left walk), then the added bases
    uint64_t *pf = wf + i * (size_t)cap;                  // forward hashes of the k-mers added (mode 0: the used set)
    const uint8_t *sb = seeds + i * (size_t)k;
    uint64_t f = 0, r = 0;
    for (uint32_t q = 0; q < uk; ++q) {
        const uint32_t c = code_of_char(sb[q]);
        if (c > 3u) { out_len[i] = 0; out_reason[i] = 4; if (SRC::kReplay) ta.phase[i] = 2; return; }
        f = rotl(f, 1) ^ seed_of(c);
        r ^= rotl(seed_of(3u - c), q);
    }
    const uint64_t seed_f = f;
    const uint8_t acgt[4] = {'A', 'C', 'G', 'T'};
    int len = 0;
    uint8_t reason = 3;
    if (resume) { f = ta.f[i]; r = ta.r[i]; len = ta.len[i]; }
    else for (uint32_t q = 0; q < uk; ++q) sq[q] = (direction == 0) ? sb[q] : sb[uk - 1u - q];
    // candidate (walk orientation: sq[len+1 .. len+k-1] + nb) against k bases given left to right
    auto cand_equals = [&](const uint8_t *other, uint32_t best_in) -> bool {
        for (uint32_t q = 0; q < uk; ++q) {
            const uint32_t cq = (q + 1u < uk) ? code_of_char(sq[(size_t)len + 1u + q]) : best_in;      // walk orientation
            const uint32_t oq = code_of_char((direction == 0) ? other[q] : other[uk - 1u - q]);
            if (cq != oq) return false;
        }
        return true;
    };
    for (;;) {
        const uint32_t oc = code_of_char(sq[len]);            // base about to leave: first base (right walk) / last base (left walk)
        const uint64_t s_out = seed_of(oc), sc_out = seed_of(3u - oc);
        uint32_t n_nb = 0, best_in = 0;
        uint64_t best_f = 0, best_r = 0;
        bool miss = false;
        for (uint32_t in = 0; in < 4u; ++in) {
            uint64_t nf, nr = 0;
            if (direction == 0) { nf = rotl(f, 1) ^ rotl(s_out, uk) ^ seed_of(in); if (!stranded) nr = rotr(r, 1) ^ rotr(sc_out, 1) ^ rotl(seed_of(3u - in), uk - 1u); }
            else { nf = rotr(f, 1) ^ rotr(s_out, 1) ^ rotl(seed_of(in), uk - 1u); if (!stranded) nr = rotl(r, 1) ^ rotl(sc_out, uk) ^ seed_of(3u - in); }
            float c;
            if (!w.get(stranded ? nf : smin(nf, nr), c)) { miss = true; continue; }
            if (c >= min_cov) { if (n_nb++ == 0) { best_in = in; best_f = nf; best_r = nr; } }
        }
        if (!miss && n_nb == 0) { reason = 0; break; }        // `while (!neighbors.isEmpty())`: a dead end ends the walk before the back-branch test (:6791)
        if (mode != 2) {                                      // back branches: variants of the current k-mer in that base
            bool back = false;                                // (a sharded graph asks for them in the same round as the neighbours)
            for (uint32_t in = 0; in < 4u && (SRC::kReplay || !back); ++in) {
                if (in == oc) continue;
                uint64_t vf, vr = 0;
                if (direction == 0) { vf = f ^ rotl(s_out, uk - 1u) ^ rotl(seed_of(in), uk - 1u); if (!stranded) vr = r ^ sc_out ^ seed_of(3u - in); }
                else { vf = f ^ s_out ^ seed_of(in); if (!stranded) vr = r ^ rotl(sc_out, uk - 1u) ^ rotl(seed_of(3u - in), uk - 1u); }
                float c;
                if (!w.get(stranded ? vf : smin(vf, vr), c)) { miss = true; continue; }
                back = back || c >= 1.0f;
            }
            if (SRC::kReplay && miss) { ta.f[i] = f; ta.r[i] = r; ta.len[i] = len; ta.phase[i] = 1; atomicAdd(&ta.ctr[1], 1u); return; }
            if (back) { reason = 1; break; }
        } else if (SRC::kReplay && miss) { ta.f[i] = f; ta.r[i] = r; ta.len[i] = len; ta.phase[i] = 1; atomicAdd(&ta.ctr[1], 1u); return; }
        if (n_nb > 1) { reason = 2; break; }
        if (mode == 0) {
            bool hit = false;
            for (int64_t j = term_koff[i]; j < term_koff[i + 1] && !hit; ++j)
                if (term_f[j] == best_f && cand_equals(term_seq + term_off[i] + (j - term_koff[i]), best_in)) hit = true;
            for (int j = 0; j < len && !hit; ++j)
                if (pf[j] == best_f) {                        // a k-mer the walk added: sq[j+1 .. j+k] in walk orientation
                    bool eq = code_of_char(sq[(size_t)j + uk]) == best_in;
                    for (uint32_t q = 0; q + 1u < uk && eq; ++q) eq = code_of_char(sq[(size_t)len + 1u + q]) == code_of_char(sq[(size_t)j + 1u + q]);
                    hit = eq;
                }
            if (hit) { reason = 5; break; }
            if (len >= cap) { reason = 6; break; }
        } else if (mode == 2) {
            bool rep = best_f == seed_f && cand_equals(sb, best_in);
            if (!rep && len > 0 && pf[len - 1] == best_f) {   // equals the k-mer added last: sq[len .. len+k-1]
                rep = code_of_char(sq[(size_t)len + uk - 1u]) == best_in;
                for (uint32_t q = 0; q + 1u < uk && rep; ++q) rep = code_of_char(sq[(size_t)len + 1u + q]) == code_of_char(sq[(size_t)len + q]);
            }
            if (rep) { reason = 7; break; }
        }
        sq[(size_t)uk + (size_t)len] = acgt[best_in];
        out_b[i * (size_t)cap + (size_t)len] = acgt[best_in];
        pf[len] = best_f;
        f = best_f; r = best_r;
        ++len;
        w.step_done();
        if (mode != 0 && len > bound) { reason = 3; break; }
    }
    out_len[i] = len;
    out_reason[i] = reason;
    if (SRC::kReplay) ta.phase[i] = 2;
}

// ---- greedy extension with lookahead: GraphUtils.greedyExtendRight / greedyExtendLeft ----
// (R/util/GraphUtils.java:1961-1976 / :1906-1921 around greedyExtendRightOnce / LeftOnce :501-529, :564-592, which
// score each candidate neighbour with getMaxMedianCoverageRight / Left :248-310, :375-438 — despite the name the
// best MINIMUM k-mer coverage over the depth-first paths of exactly `lookahead` k-mers that start at the candidate.)
// One lane per walk.  Per step: candidates = neighbours with count >= 1 in order A,C,G,T (Kmer.getSuccessors,
// R/graph/Kmer.java:228-255); none -> stop; one -> take it; else the candidate with the largest score, a tie
// going to the larger count (strictly).  No visited set: the reference has none here.  The depth-first search keeps,
// per level, the siblings not yet tried — the reference's `frontier` of neighbour deques.
constexpr int WALK_MAX_LOOKAHEAD = 16;
struct WalkCand { uint64_t f, r; float c; uint32_t in; };
struct WalkGate { const uint32_t *bits; Mod mod; int num_hash; };      // the extra BloomFilter of the `bf` variants, bits == nullptr: none
// (-1: a count is not known yet — sharded graphs; the requests for all four neighbours are filed)
template <class W>
__device__ __forceinline__ int walk_neighbors(W &w, uint64_t kmul, const WalkGate &gate, int stranded, uint32_t uk, int direction,
                                              uint64_t f, uint64_t r, uint32_t oc, float min_cov, WalkCand *out) {
    const uint64_t s_out = seed_of(oc), sc_out = seed_of(3u - oc);
    int n = 0;
    bool miss = false;
    for (uint32_t in = 0; in < 4u; ++in) {
        uint64_t nf, nr = 0;
        if (direction == 0) {
            nf = rotl(f, 1) ^ rotl(s_out, uk) ^ seed_of(in);
            if (!stranded) nr = rotr(r, 1) ^ rotr(sc_out, 1) ^ rotl(seed_of(3u - in), uk - 1u);
        } else {
            nf = rotr(f, 1) ^ rotr(s_out, 1) ^ rotl(seed_of(in), uk - 1u);
            if (!stranded) nr = rotl(r, 1) ^ rotl(sc_out, uk) ^ seed_of(3u - in);
        }
        const uint64_t h0 = stranded ? nf : smin(nf, nr);
        if (gate.bits && !bits_lookup(gate.bits, gate.mod, gate.num_hash, kmul, h0)) continue;   // Kmer.getSuccessors(k, numHash, graph, bf): bf.lookup first
        float c;
        if (!w.get(h0, c)) { miss = true; continue; }
        if (c >= min_cov) { out[n].f = nf; out[n].r = nr; out[n].c = c; out[n].in = in; ++n; }
    }
    return miss ? -1 : n;
}
template <class SRC>
__global__ void k_greedy_extend(SRC src, uint64_t kmul, WalkGate gate, int stranded, int k, int direction, const uint8_t *__restrict__ seeds, size_t n,
                                int lookahead, int bound, uint8_t *__restrict__ seq, uint8_t *__restrict__ out_b,
                                float *__restrict__ out_c, int32_t *__restrict__ out_len, uint8_t *__restrict__ out_reason) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TravArrays ta = src.arrays();
    if (SRC::kReplay && ta.phase[i] == 2) return;
    if (SRC::kReplay && ta.over[i]) { out_len[i] = ta.len[i]; out_reason[i] = WALK_REASON_TOO_WIDE; ta.phase[i] = 2; return; }
    const bool resume = SRC::kReplay && ta.phase[i] == 1;
    auto w = src.walk(i);
    const uint32_t uk = (uint32_t)k;
    const size_t stride = (size_t)k + (size_t)bound + (size_t)WALK_MAX_LOOKAHEAD + 1;
    uint8_t *sq = seq + i * stride;                    // walk orientation, see k_walk_max_cov; the search writes ahead of the walk
    const uint8_t *sb = seeds + i * (size_t)k;
    uint64_t f = 0, r = 0;
    for (uint32_t q = 0; q < uk; ++q) {
        const uint32_t c = code_of_char(sb[q]);
        if (c > 3u) { out_len[i] = 0; out_reason[i] = 4; if (SRC::kReplay) ta.phase[i] = 2; return; }
        f = rotl(f, 1) ^ seed_of(c);
        r ^= rotl(seed_of(3u - c), q);
    }
    const uint8_t acgt[4] = {'A', 'C', 'G', 'T'};
    int len = 0;
    uint8_t reason = 3;
    if (resume) { f = ta.f[i]; r = ta.r[i]; len = ta.len[i]; }
    else for (uint32_t q = 0; q < uk; ++q) sq[q] = (direction == 0) ? sb[q] : sb[uk - 1u - q];
    WalkCand cand[4];
    WalkCand frontier[WALK_MAX_LOOKAHEAD][4];          // siblings not yet tried, per level of the search
    int fr_n[WALK_MAX_LOOKAHEAD], fr_next[WALK_MAX_LOOKAHEAD];
    WalkCand path[WALK_MAX_LOOKAHEAD + 1];             // path[0] = the candidate being scored ("source")
    bool suspended = false;                            // sharded graphs: a neighbourhood whose counts are not known yet
    while (len < bound) {
        const int nc = walk_neighbors(w, kmul, gate, stranded, uk, direction, f, r, code_of_char(sq[len]), 1.0f, cand);
        if (nc < 0) { suspended = true; break; }
        if (nc == 0) { reason = 0; break; }
        int best = 0;
        if (nc > 1) {
            float best_cov = -1.0f;
            for (int ci = 0; ci < nc; ++ci) {
                // getMaxMedianCoverageRight(graph, cand[ci], lookahead): sq[len + k] is the candidate's new base
                sq[(size_t)len + uk] = acgt[cand[ci].in];
                float score;
                path[0] = cand[ci];
                int psize = 1, depth = 0;                // psize = path.size(); depth = frontier.size()
                WalkCand nb[4];
                int nn = walk_neighbors(w, kmul, gate, stranded, uk, direction, cand[ci].f, cand[ci].r, code_of_char(sq[(size_t)len + 1u]), 1.0f, nb);
                // (sharded graphs: an unknown neighbourhood is filed and taken for a dead end, and the search goes on — through the
                // siblings and the other candidates — so that ONE exchange round brings every neighbourhood that can be asked for now:
                // a level of the search per round instead of a neighbourhood per round.  The step is replayed anyway.)
                if (nn < 0) { suspended = true; nn = 0; }
                if (nn == 0) score = (lookahead > 0) ? 0.0f : cand[ci].c;
                else {
                    float best_path = 0.0f;
                    for (int q = 0; q < nn; ++q) frontier[0][q] = nb[q];
                    fr_n[0] = nn; fr_next[0] = 1; depth = 1;
                    path[1] = nb[0]; psize = 2;
                    sq[(size_t)len + uk + 1u] = acgt[nb[0].in];
                    while (depth > 0) {
                        if (psize < lookahead) {
                            const WalkCand &cur = path[psize - 1];
                            // cursor = k-mer number (psize-1) after the candidate: its leaving base is sq[len + 1 + (psize-1)]
                            nn = walk_neighbors(w, kmul, gate, stranded, uk, direction, cur.f, cur.r, code_of_char(sq[(size_t)len + (size_t)psize]), 1.0f, nb);
                            if (nn < 0) { suspended = true; nn = 0; }
                            if (nn > 0) {
                                for (int q = 0; q < nn; ++q) frontier[depth][q] = nb[q];
                                fr_n[depth] = nn; fr_next[depth] = 1; ++depth;
                                path[psize] = nb[0];
                                sq[(size_t)len + uk + (size_t)psize] = acgt[nb[0].in];
                                ++psize;
                                continue;
                            }
                        }
                        if (psize == lookahead) {
                            float mn = path[0].c;
                            for (int q = 1; q < psize; ++q) mn = path[q].c < mn ? path[q].c : mn;
                            if (best_path < mn) best_path = mn;
                        }
                        while (depth > 0) {
                            --psize;                                        // path.removeLast()
                            if (fr_next[depth - 1] >= fr_n[depth - 1]) --depth;  // that level is exhausted
                            else {
                                path[psize] = frontier[depth - 1][fr_next[depth - 1]++];
                                sq[(size_t)len + uk + (size_t)psize] = acgt[path[psize].in];
                                ++psize;
                                break;
                            }
                        }
                    }
                    score = best_path;
                }
                if (score > best_cov) { best = ci; best_cov = score; }
                else if (score == best_cov && cand[ci].c > cand[best].c) best = ci;
            }
            if (suspended) break;
        }
        sq[(size_t)len + uk] = acgt[cand[best].in];
        out_b[i * (size_t)bound + (size_t)len] = acgt[cand[best].in];
        out_c[i * (size_t)bound + (size_t)len] = cand[best].c;
        f = cand[best].f; r = cand[best].r;
        ++len;
        w.step_done();
    }
    if (SRC::kReplay && suspended) { ta.f[i] = f; ta.r[i] = r; ta.len[i] = len; ta.phase[i] = 1; atomicAdd(&ta.ctr[1], 1u); return; }
    out_len[i] = len;
    out_reason[i] = reason;
    if (SRC::kReplay) ta.phase[i] = 2;
}

// ---- BloomFilter.lookupThenAdd over an array, in array order (R/bloom/BloomFilter.java:147-155) ----
// Sequentially, element i finds bit b set iff b was set before the call or an earlier probe of the array set it:
// the probe with the smallest (element, hash number) id among those that want a clear bit is its first setter.
// Pass 1 tests the bits against the state before the call and registers candidates for first setter in a hash
// table (atomicMin of the probe id); pass 2 answers every element from its own bits and the table; then all
// bits are set.  Same arbitration as the dbgbf half of k_probe / k_late_claim.
__global__ void k_lta_probe(const uint32_t *__restrict__ bits, Mod mod, int num_hash, uint64_t kmul, const uint64_t *__restrict__ h0,
                            size_t n, Slot *ftable, uint32_t f_log2, uint8_t *__restrict__ premask) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t pm = 0;
    for (int j = 0; j < num_hash; ++j) {
        const uint64_t idx = index_of(multi_hash(h0[i], (uint32_t)j, kmul), mod);
        if (bit_test(bits, idx)) pm |= 1u << j;
        else {
            Slot *s = table_insert(ftable, f_log2, idx);
            atomicMin(&s->val, ((unsigned long long)i << 4) | (unsigned long long)j);
        }
    }
    premask[i] = (uint8_t)pm;
}
__global__ void k_lta_resolve(Mod mod, int num_hash, uint64_t kmul, const uint64_t *__restrict__ h0, size_t n, const Slot *ftable,
                              uint32_t f_log2, const uint8_t *__restrict__ premask, uint8_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pm = premask[i];
    bool found = true;
    for (int j = 0; j < num_hash; ++j) {
        if ((pm >> j) & 1u) continue;
        const Slot *s = table_find(ftable, f_log2, index_of(multi_hash(h0[i], (uint32_t)j, kmul), mod));
        if (!(s->val < (((unsigned long long)i << 4) | (unsigned long long)j))) found = false;   // no early exit in the reference either
    }
    out[i] = found ? 1 : 0;
}

}  // namespace

using rb::HostPin;
extern "C" {
int rb_filter_lookup(rb_graph *g, int which, const uint64_t *h0, size_t n, uint8_t *out) {
    return guarded([&] {
        RB_REQUIRE(g && (n == 0 || (h0 && out)), "rb_filter_lookup: null argument");
        BitFilter *f = bit_filter(g, which);
        RB_REQUIRE(f, "rb_filter_lookup: filter %d is not a bit filter", which);
        RB_REQUIRE(!g->shard, "rb_filter_lookup: queries are not available on a shard handle");
        if (!f->bits) { set_error("rb_filter_lookup: filter %d not initialised", which); throw HipError{RB_ERR_STATE}; }
        if (!n) return;
        HostPin pin_in(h0, n * 8), pin_out(out, n);
        QueryLease q(g);
        if (!f->bits) { set_error("rb_filter_lookup: filter %d not initialised", which); throw HipError{RB_ERR_STATE}; }
        uint64_t *d = upload_h0(g, q.c->b0, h0, n, q.c->st);
        q.c->b1.reserve(n);
        hipLaunchKernelGGL(k_bits_lookup, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, q.c->st, f->bits, f->mod, f->num_hash,
                           kmul_of(g->k), d, n, q.c->b1.as<uint8_t>());
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpyAsync(out, q.c->b1.p, n, hipMemcpyDeviceToHost, q.c->st));
        RB_HIP(hipStreamSynchronize(q.c->st));
    });
}
int rb_filter_lookup_then_add(rb_graph *g, int which, const uint64_t *h0, size_t n, uint8_t *out) {
    return guarded([&] {
        RB_REQUIRE(g && (n == 0 || (h0 && out)), "rb_filter_lookup_then_add: null argument");
        WriteLock wl(g->rw);
        BitFilter *f = bit_filter(g, which);
        RB_REQUIRE(f, "rb_filter_lookup_then_add: filter %d is not a bit filter", which);
        RB_REQUIRE(!g->shard, "rb_filter_lookup_then_add: not available on a shard handle");
        RB_REQUIRE(n < ((size_t)1 << 59), "rb_filter_lookup_then_add: too many elements");
        if (!f->bits) { set_error("rb_filter_lookup_then_add: filter %d not initialised", which); throw HipError{RB_ERR_STATE}; }
        if (!n) return;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        uint64_t *d = upload_h0(g, g->qbuf0, h0, n);
        g->qbuf1.reserve(n); g->qbuf2.reserve(n);
        const uint32_t f_log2 = log2_ceil(2ull * (uint64_t)n * (uint64_t)f->num_hash + 2);
        g->ftable.reserve(sizeof(Slot) << f_log2);
        RB_HIP(hipMemsetAsync(g->ftable.p, 0xFF, sizeof(Slot) << f_log2, s));
        const uint64_t kmul = kmul_of(g->k);
        hipLaunchKernelGGL(k_lta_probe, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, s, f->bits, f->mod, f->num_hash, kmul, d, n,
                           g->ftable.as<Slot>(), f_log2, g->qbuf2.as<uint8_t>());
        hipLaunchKernelGGL(k_lta_resolve, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, s, f->mod, f->num_hash, kmul, d, n,
                           g->ftable.as<Slot>(), f_log2, g->qbuf2.as<uint8_t>(), g->qbuf1.as<uint8_t>());
        hipLaunchKernelGGL(k_bits_add, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, s, f->bits, f->mod, f->num_hash, kmul, d, n);
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpyAsync(out, g->qbuf1.p, n, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
    });
}
int rb_graph_contains(rb_graph *g, const uint64_t *h0, size_t n, uint8_t *out) { return rb_filter_lookup(g, RB_DBGBF, h0, n, out); }

static int count_common(rb_graph *g, const uint64_t *h0, size_t n, float *out, bool graph_level) {
    return guarded([&] {
        RB_REQUIRE(g && (n == 0 || (h0 && out)), "rb_graph_count: null argument");
        RB_REQUIRE(!g->shard, "rb_graph_count: queries are not available on a shard handle");
        if (!n) return;
        HostPin pin_in(h0, n * 8), pin_out(out, n * 4);
        QueryLease q(g);
        uint64_t *d = upload_h0(g, q.c->b0, h0, n, q.c->st);
        q.c->b1.reserve(n * 4);
        RB_REQUIRE(g->cbf, "rb_filter_get_count: the counting filter has been destroyed");
        FilterView fv = g->view(0, 0, graph_level);
        if (graph_level) hipLaunchKernelGGL(k_graph_count, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, q.c->st, fv, d, n, q.c->b1.as<float>());
        else hipLaunchKernelGGL(k_cbf_count, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, q.c->st, fv, d, n, q.c->b1.as<float>());
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpyAsync(out, q.c->b1.p, n * 4, hipMemcpyDeviceToHost, q.c->st));
        RB_HIP(hipStreamSynchronize(q.c->st));
    });
}
}  // extern "C"
// CountingBloomFilter.getCount(long) for hashes that already sit in device memory (rb_sketch.hip: strobemer / k-mer-pair
// hash -> count without a trip through the host); enqueued on the graph's stream, not synchronised
void rb::cbf_counts_device(rb_graph *g, const uint64_t *d_h0, size_t n, float *d_out) {
    RB_REQUIRE(g && !g->shard && g->cbf, "count lookup: handle without a local counting filter");
    if (!n) return;
    hipLaunchKernelGGL(k_cbf_count, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, g->stream, g->view(0, 0, false), d_h0, n, d_out);
    RB_HIP(hipGetLastError());
}
extern "C" {
int rb_graph_count(rb_graph *g, const uint64_t *h0, size_t n, float *out) { return count_common(g, h0, n, out, true); }
int rb_filter_get_count(rb_graph *g, const uint64_t *h0, size_t n, float *out) { return count_common(g, h0, n, out, false); }

int rb_graph_kmers(rb_graph *g, const char *seq, const int64_t *offsets, int64_t n_reads, int64_t *koffsets,
                   uint64_t *f, uint64_t *r, float *count) {
    return guarded([&] {
        RB_REQUIRE(g && offsets && koffsets && n_reads >= 0, "rb_graph_kmers: null argument");
        koffsets[0] = 0;
        for (int64_t i = 0; i < n_reads; ++i) {
            int64_t l = offsets[i + 1] - offsets[i];
            koffsets[i + 1] = koffsets[i] + (l >= g->k ? l - g->k + 1 : 0);
        }
        const int64_t total = koffsets[n_reads];
        if (!f || !count || total == 0) return;
        RB_HIP(hipSetDevice(g->p.device));
        HostPin pin_seq(seq + offsets[0], (size_t)(offsets[n_reads] - offsets[0])), pin_f(f, (size_t)total * 8), pin_r(r, (size_t)total * 8),
                pin_c(count, (size_t)total * 4);
        QueryLease q(g);
        hipStream_t s = q.c->st;
        // in pieces of <= 16 M k-mers (20 bytes of device scratch each): the scratch stays at 320 MB however many reads are asked for
        const int64_t piece_max = getenv("RB_QUERY_PIECE") ? std::max<int64_t>(1, atoll(getenv("RB_QUERY_PIECE"))) : (int64_t)16 << 20;
        std::vector<int64_t> rel;
        for (int64_t ra = 0; ra < n_reads;) {
            int64_t lo = ra + 1, hi = n_reads;
            while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (koffsets[mid] - koffsets[ra] <= piece_max) lo = mid; else hi = mid - 1; }
            const int64_t rb_ = lo, pn = rb_ - ra, pt = koffsets[rb_] - koffsets[ra];
            if (pt > 0) {
                rb::AsciiUpload up;
                rb_batch *b = nullptr;
                try {
                    rb::ascii_batch_begin(up, g->p.device, seq, nullptr, offsets, ra, pn, 0, s, true);
                    b = rb::ascii_batch_finish(up);
                } catch (...) { rb::ascii_batch_abort(up); throw; }
                struct G { rb_batch *b; ~G() { rb_batch_destroy(b); } } guard{b};
                rel.resize((size_t)pn + 1);
                for (int64_t i = 0; i <= pn; ++i) rel[(size_t)i] = koffsets[ra + i] - koffsets[ra];
                q.c->b0.reserve(((size_t)pn + 1) * 8); q.c->b1.reserve((size_t)pt * 8); q.c->b2.reserve((size_t)pt * 8); q.c->b3.reserve((size_t)pt * 4);
                RB_HIP(hipMemcpyAsync(q.c->b0.p, rel.data(), ((size_t)pn + 1) * 8, hipMemcpyHostToDevice, s));
                // on a shard of a distributed graph only the hashes are local (count = 1 for a usable window): the caller gets the
                // counts with one rb_shard_query_* exchange (rnabloom/sharded.py::ShardRank.getKmers)
                if (g->shard)
                    hipLaunchKernelGGL(k_get_kmers<true>, dim3(blocks_for(b->n_words)), dim3(TPB), 0, s, g->view(0, 0), (int)g->stranded,
                                       b->codes, b->valid, b->rnz, b->word_read, b->woff, b->len, b->n_words, g->k, q.c->b0.as<int64_t>(),
                                       q.c->b1.as<uint64_t>(), q.c->b2.as<uint64_t>(), q.c->b3.as<float>());
                else
                    hipLaunchKernelGGL(k_get_kmers<false>, dim3(blocks_for(b->n_words)), dim3(TPB), 0, s, g->view(0, 0), (int)g->stranded,
                                       b->codes, b->valid, b->rnz, b->word_read, b->woff, b->len, b->n_words, g->k, q.c->b0.as<int64_t>(),
                                       q.c->b1.as<uint64_t>(), q.c->b2.as<uint64_t>(), q.c->b3.as<float>());
                RB_HIP(hipGetLastError());
                const int64_t o = koffsets[ra];
                RB_HIP(hipMemcpyAsync(f + o, q.c->b1.p, (size_t)pt * 8, hipMemcpyDeviceToHost, s));
                if (r) RB_HIP(hipMemcpyAsync(r + o, q.c->b2.p, (size_t)pt * 8, hipMemcpyDeviceToHost, s));
                RB_HIP(hipMemcpyAsync(count + o, q.c->b3.p, (size_t)pt * 4, hipMemcpyDeviceToHost, s));
                RB_HIP(hipStreamSynchronize(s));                  // (rel and the piece's batch are released next)
            }
            ra = rb_;
        }
    });
}

int rb_graph_batch_counts(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, const int64_t *koffsets, float *out, int out_on_device,
                          int64_t *stride_out) {
    return guarded([&] {
        RB_REQUIRE(g && b && (n == 0 || out), "rb_graph_batch_counts: null argument");
        RB_REQUIRE(first >= 0 && n >= 0 && first + n <= b->n_reads, "rb_graph_batch_counts: read range outside the batch");
        RB_REQUIRE(!g->shard, "rb_graph_batch_counts: queries are not available on a shard handle");
        RB_REQUIRE(b->device == g->p.device, "rb_graph_batch_counts: batch and graph live on different devices");
        const int64_t stride = b->max_len >= (uint32_t)g->k ? (int64_t)b->max_len - g->k + 1 : 0;
        if (stride_out) *stride_out = stride;
        const int64_t total = koffsets ? (n ? koffsets[n] - koffsets[0] : 0) : n * stride;
        if (n == 0 || total == 0) return;
        RB_REQUIRE(!koffsets || koffsets[0] == 0, "rb_graph_batch_counts: koffsets[0] must be 0");
        RB_REQUIRE(g->cbf, "rb_graph_batch_counts: the counting filter has been destroyed");
        RB_HIP(hipSetDevice(g->p.device));
        HostPin pin_out(out_on_device ? nullptr : out, (size_t)total * 4);
        QueryLease q(g);
        hipStream_t s = q.c->st;
        const int64_t *dko = nullptr;
        if (koffsets) {
            q.c->b0.reserve(((size_t)n + 1) * 8);
            RB_HIP(hipMemcpyAsync(q.c->b0.p, koffsets, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, s));
            dko = q.c->b0.as<int64_t>();
        }
        auto launch = [&](int64_t ra, int64_t rb_, int64_t row_base, float *dst) {      // reads [first + ra, first + rb_): rows at dst + row - row_base
            const int64_t w0 = b->h_woff[(size_t)(first + ra)], nw = (int64_t)b->h_woff[(size_t)(first + rb_)] - w0;
            if (nw > 0)
                hipLaunchKernelGGL(k_batch_counts, dim3(blocks_for(nw, 256)), dim3(256), 0, s, g->view(0, 0), (int)g->stranded, b->codes, b->valid,
                                   b->word_read, b->woff, b->len, w0, nw, (uint32_t)first, g->k, dko, stride, row_base, dst);
            RB_HIP(hipGetLastError());
        };
        // (stride mode: rows are padded where a read is shorter than the longest one, and reads shorter than k have no thread at all)
        if (out_on_device) {
            if (!koffsets) RB_HIP(hipMemsetAsync(out, 0, (size_t)total * 4, s));
            launch(0, n, 0, out);
            RB_HIP(hipStreamSynchronize(s));
            return;
        }
        // To the host in pieces of <= 64 M counts through two device buffers: the copy of piece c runs on its own stream beside
        // the kernel of piece c + 1, and the scratch stays at 512 MB however many reads are asked for.
        const int64_t piece_max = getenv("RB_QUERY_PIECE") ? std::max<int64_t>(1, atoll(getenv("RB_QUERY_PIECE"))) : (int64_t)64 << 20;
        std::vector<int64_t> cut{0};                              // read boundaries of the pieces (at least one read each)
        auto row_of = [&](int64_t i) { return koffsets ? koffsets[i] : i * stride; };
        int64_t largest = 0;
        while (cut.back() < n) {
            int64_t a = cut.back(), lo = a + 1, hi = n;
            while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (row_of(mid) - row_of(a) <= piece_max) lo = mid; else hi = mid - 1; }
            cut.push_back(lo);
            largest = std::max(largest, row_of(lo) - row_of(a));
        }
        q.c->b3.reserve((size_t)largest * 4 * 2);
        float *buf[2] = {q.c->b3.as<float>(), q.c->b3.as<float>() + largest};
        hipStream_t s2 = nullptr;
        std::vector<hipEvent_t> ev;
        struct Cleanup { hipStream_t &s2; std::vector<hipEvent_t> &ev; ~Cleanup() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); if (s2) (void)hipStreamDestroy(s2); } } cleanup{s2, ev};
        RB_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        std::vector<hipEvent_t> copied;                           // per piece: its copy to the host is done (its buffer is free again)
        for (size_t c = 0; c + 1 < cut.size(); ++c) {
            const int64_t ra = cut[c], rb_ = cut[c + 1], oa = row_of(ra), ob = row_of(rb_);
            float *dst = buf[c & 1];
            if (c >= 2) RB_HIP(hipStreamWaitEvent(s, copied[c - 2], 0));
            if (ob > oa) {
                if (!koffsets) RB_HIP(hipMemsetAsync(dst, 0, (size_t)(ob - oa) * 4, s));
                launch(ra, rb_, oa, dst);
            }
            hipEvent_t e, e2;
            RB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev.push_back(e);
            RB_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming)); ev.push_back(e2);
            RB_HIP(hipEventRecord(e, s));
            RB_HIP(hipStreamWaitEvent(s2, e, 0));
            if (ob > oa) RB_HIP(hipMemcpyAsync(out + oa, dst, (size_t)(ob - oa) * 4, hipMemcpyDeviceToHost, s2));
            RB_HIP(hipEventRecord(e2, s2));
            copied.push_back(e2);
        }
        RB_HIP(hipStreamSynchronize(s));
        RB_HIP(hipStreamSynchronize(s2));
    });
}

int rb_graph_neighbors(rb_graph *g, const uint64_t *f, const uint64_t *r, const uint8_t *char_out, size_t n,
                       int direction, uint64_t *f4, uint64_t *r4, float *count4) {
    return guarded([&] {
        RB_REQUIRE(g && (n == 0 || (f && char_out && f4 && count4)), "rb_graph_neighbors: null argument");
        RB_REQUIRE(g->stranded || n == 0 || r, "rb_graph_neighbors: reverse hashes required for a canonical graph");
        RB_REQUIRE(direction >= 0 && direction <= 3, "rb_graph_neighbors: direction must be 0..3");
        if (!n) return;
        QueryLease q(g);
        hipStream_t s = q.c->st;
        q.c->b0.reserve(n * 8 * 2 + n); q.c->b1.reserve(n * 32); q.c->b2.reserve(n * 32); q.c->b3.reserve(n * 16);
        uint64_t *df = q.c->b0.as<uint64_t>(), *dr = df + n;
        uint8_t *dc = reinterpret_cast<uint8_t *>(dr + n);
        RB_HIP(hipMemcpyAsync(df, f, n * 8, hipMemcpyHostToDevice, s));
        if (r) RB_HIP(hipMemcpyAsync(dr, r, n * 8, hipMemcpyHostToDevice, s));
        RB_HIP(hipMemcpyAsync(dc, char_out, n, hipMemcpyHostToDevice, s));
        if (g->shard)      // hashes only (count4 = 0): the counts of a distributed graph come from a query exchange (ShardRank.neighbors)
            hipLaunchKernelGGL(k_neighbors<true>, dim3(blocks_for((int64_t)n * 4)), dim3(TPB), 0, s, g->view(0, 0), (int)g->stranded, g->k,
                               direction, df, dr, dc, n, q.c->b1.as<uint64_t>(), q.c->b2.as<uint64_t>(), q.c->b3.as<float>());
        else
            hipLaunchKernelGGL(k_neighbors<false>, dim3(blocks_for((int64_t)n * 4)), dim3(TPB), 0, s, g->view(0, 0), (int)g->stranded, g->k,
                               direction, df, dr, dc, n, q.c->b1.as<uint64_t>(), q.c->b2.as<uint64_t>(), q.c->b3.as<float>());
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpyAsync(f4, q.c->b1.p, n * 32, hipMemcpyDeviceToHost, s));
        if (r4) RB_HIP(hipMemcpyAsync(r4, q.c->b2.p, n * 32, hipMemcpyDeviceToHost, s));
        RB_HIP(hipMemcpyAsync(count4, q.c->b3.p, n * 16, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
    });
}

int rb_graph_walk(rb_graph *g, const char *seeds, const char *targets, size_t n, int direction, int bound, float min_cov,
                  char *out_bases, uint64_t *out_f, uint64_t *out_r, float *out_count, int32_t *out_len, uint8_t *out_reason) {
    return guarded([&] {
        RB_REQUIRE(g && (n == 0 || (seeds && out_bases && out_len && out_reason)), "rb_graph_walk: null argument");
        RB_REQUIRE(direction == 0 || direction == 1, "rb_graph_walk: direction must be 0 (right) or 1 (left)");
        RB_REQUIRE(bound >= 1 && bound <= (1 << 20), "rb_graph_walk: bound out of range [1, 2^20]");
        RB_REQUIRE(!g->shard, "rb_graph_walk: queries are not available on a shard handle");
        if (!n) return;
        QueryLease q(g);
        hipStream_t s = q.c->st;
        const size_t k = (size_t)g->k, nb = n * (size_t)bound, stride = k + (size_t)bound;
        q.c->b0.reserve(n * k * 2 + n * stride + nb + 64);     // seeds | targets | seq | appended bases
        q.c->b1.reserve(nb * 8); q.c->b2.reserve(nb * 8); q.c->b3.reserve(nb * 4 + n * 4 + n + 64);
        uint8_t *dseed = q.c->b0.as<uint8_t>(), *dtarget = dseed + n * k, *dseq = dtarget + n * k, *dbases = dseq + n * stride;
        float *dc = q.c->b3.as<float>();
        int32_t *dlen = reinterpret_cast<int32_t *>(dc + nb);
        uint8_t *dreason = reinterpret_cast<uint8_t *>(dlen + n);
        RB_HIP(hipMemcpyAsync(dseed, seeds, n * k, hipMemcpyHostToDevice, s));
        if (targets) RB_HIP(hipMemcpyAsync(dtarget, targets, n * k, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_walk_max_cov<DirectCounts>, dim3(blocks_for((int64_t)n, 64)), dim3(64), 0, s, DirectCounts{g->view(0, 0)}, (int)g->stranded, g->k, direction,
                           dseed, targets ? dtarget : (const uint8_t *)nullptr, n, bound, min_cov, dseq, dbases, q.c->b1.as<uint64_t>(),
                           q.c->b2.as<uint64_t>(), dc, dlen, dreason);
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpyAsync(out_len, dlen, n * 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipMemcpyAsync(out_reason, dreason, n, hipMemcpyDeviceToHost, s));
        if (out_f) RB_HIP(hipMemcpyAsync(out_f, q.c->b1.p, nb * 8, hipMemcpyDeviceToHost, s));
        if (out_r) RB_HIP(hipMemcpyAsync(out_r, q.c->b2.p, nb * 8, hipMemcpyDeviceToHost, s));
        if (out_count) RB_HIP(hipMemcpyAsync(out_count, dc, nb * 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipMemcpyAsync(out_bases, dbases, nb, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
    });
}

int rb_graph_greedy_extend(rb_graph *g, const rb_graph *gate, const char *seeds, size_t n, int direction, int lookahead, int bound,
                           char *out_bases, float *out_count, int32_t *out_len, uint8_t *out_reason) {
    return guarded([&] {
        if (gate) {
            RB_REQUIRE(gate->dbg.bits && !gate->shard && gate->p.device == g->p.device && gate->k == g->k,
                       "rb_graph_greedy_extend: the gate must be a filter on the same device with the same k");
        }
        RB_REQUIRE(g && (n == 0 || (seeds && out_bases && out_len && out_reason)), "rb_graph_greedy_extend: null argument");
        RB_REQUIRE(direction == 0 || direction == 1, "rb_graph_greedy_extend: direction must be 0 (right) or 1 (left)");
        RB_REQUIRE(bound >= 1 && bound <= (1 << 20), "rb_graph_greedy_extend: bound out of range [1, 2^20]");
        RB_REQUIRE(lookahead >= 0 && lookahead <= WALK_MAX_LOOKAHEAD, "rb_graph_greedy_extend: lookahead out of range [0, %d]", WALK_MAX_LOOKAHEAD);
        RB_REQUIRE(!g->shard, "rb_graph_greedy_extend: queries are not available on a shard handle");
        if (!n) return;
        // the gate is read too: shared lock on it for the call (taken before g's when its address is lower: two calls that
        // name each other as graph and gate cannot deadlock against a writer waiting in between)
        rb_graph *gm = const_cast<rb_graph *>(gate);
        std::shared_lock<std::shared_mutex> gate_lk;
        if (gm && gm != g && gm < g) gate_lk = std::shared_lock<std::shared_mutex>(gm->rw);
        QueryLease q(g);
        if (gm && gm != g && gm > g) gate_lk = std::shared_lock<std::shared_mutex>(gm->rw);
        if (gate) RB_REQUIRE(gate->dbg.bits, "rb_graph_greedy_extend: the gate's filter has been destroyed");
        hipStream_t s = q.c->st;
        const size_t k = (size_t)g->k, nb = n * (size_t)bound, stride = k + (size_t)bound + (size_t)WALK_MAX_LOOKAHEAD + 1;
        q.c->b0.reserve(n * k + n * stride + nb + 64);          // seeds | seq | appended bases
        q.c->b3.reserve(nb * 4 + n * 4 + n + 64);
        uint8_t *dseed = q.c->b0.as<uint8_t>(), *dseq = dseed + n * k, *dbases = dseq + n * stride;
        float *dc = q.c->b3.as<float>();
        int32_t *dlen = reinterpret_cast<int32_t *>(dc + nb);
        uint8_t *dreason = reinterpret_cast<uint8_t *>(dlen + n);
        RB_HIP(hipMemcpyAsync(dseed, seeds, n * k, hipMemcpyHostToDevice, s));
        WalkGate wg{gate ? gate->dbg.bits : nullptr, gate ? gate->dbg.mod : g->dbg.mod, gate ? gate->dbg.num_hash : 0};
        hipLaunchKernelGGL(k_greedy_extend<DirectCounts>, dim3(blocks_for((int64_t)n, 64)), dim3(64), 0, s, DirectCounts{g->view(0, 0)}, kmul_of(g->k), wg, (int)g->stranded, g->k, direction,
                           dseed, n, lookahead, bound, dseq, dbases, dc, dlen, dreason);
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpyAsync(out_len, dlen, n * 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipMemcpyAsync(out_reason, dreason, n, hipMemcpyDeviceToHost, s));
        if (out_count) RB_HIP(hipMemcpyAsync(out_count, dc, nb * 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipMemcpyAsync(out_bases, dbases, nb, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
    });
}

// naiveExtend*'s terminators (mode 0): forward hash of every k-mer of every terminator sequence, hashed on the host side of the
// call (they are short — the k-mers of the fragment being extended); offsets relative to the first sequence
static void naive_terminators(size_t k, size_t n, int mode, const char *term_seq, const int64_t *term_off, std::vector<int64_t> &tko,
                              std::vector<int64_t> &rel, std::vector<uint64_t> &tf, size_t &tbytes) {
    tko.assign(n + 1, 0); rel.assign(n + 1, 0);
    tbytes = 0;
    if (mode == 0) {
        for (size_t i = 0; i < n; ++i) { const int64_t l = term_off[i + 1] - term_off[i]; tko[i + 1] = tko[i] + (l >= (int64_t)k ? l - (int64_t)k + 1 : 0); }
        for (size_t i = 0; i <= n; ++i) rel[i] = term_off[i] - term_off[0];
        tbytes = (size_t)(term_off[n] - term_off[0]);
    }
    tf.assign(std::max<size_t>((size_t)tko[n], 1), 0);
    if (mode != 0) return;
    for (size_t i = 0; i < n; ++i) {
        const char *t = term_seq + term_off[i];
        const int64_t l = term_off[i + 1] - term_off[i];
        for (int64_t p = 0; p + (int64_t)k <= l; ++p) {   // NTP64: forward hash from scratch (an N hashes as seed 0 and never equals a candidate)
            uint64_t f = 0;
            for (size_t x = 0; x < k; ++x) {
                uint64_t sd = 0;
                switch (t[p + (int64_t)x]) { case 'A': case 'a': sd = 0x3c8bfbb395c60474ull; break; case 'C': case 'c': sd = 0x3193c18562a02b4cull; break;
                                             case 'G': case 'g': sd = 0x20323ed082572324ull; break; case 'T': case 't': case 'U': case 'u': sd = 0x295549f54be24456ull; break; default: break; }
                f = ((f << 1) | (f >> 63)) ^ sd;
            }
            tf[(size_t)tko[i] + (size_t)p] = f;
        }
    }
}

int rb_graph_naive_extend(rb_graph *g, const char *seeds, size_t n, int direction, int mode, int bound, int cap, float min_cov,
                          const char *term_seq, const int64_t *term_off, char *out_bases, int32_t *out_len, uint8_t *out_reason) {
    return guarded([&] {
        RB_REQUIRE(g && (n == 0 || (seeds && out_bases && out_len && out_reason)), "rb_graph_naive_extend: null argument");
        RB_REQUIRE(direction == 0 || direction == 1, "rb_graph_naive_extend: direction must be 0 (right) or 1 (left)");
        RB_REQUIRE(mode >= 0 && mode <= 2, "rb_graph_naive_extend: mode must be 0 (terminators), 1 (bounded) or 2 (bounded, no back checks)");
        RB_REQUIRE(mode == 0 ? (cap >= 1 && cap <= (1 << 20) && (n == 0 || (term_seq && term_off))) : (bound >= 0 && bound < (1 << 20)),
                   "rb_graph_naive_extend: mode 0 needs a capacity and the terminator sequences, modes 1 / 2 a bound");
        RB_REQUIRE(!g->shard, "rb_graph_naive_extend: queries are not available on a shard handle");
        if (!n) return;
        if (mode != 0) cap = bound + 1;                          // ++extensionLength > bound: up to bound + 1 k-mers
        QueryLease q(g);
        hipStream_t s = q.c->st;
        const size_t k = (size_t)g->k, stride = k + (size_t)cap;
        std::vector<int64_t> tko, rel;
        std::vector<uint64_t> tf;
        size_t tbytes = 0;
        naive_terminators(k, n, mode, term_seq, term_off, tko, rel, tf, tbytes);
        const size_t nt = (size_t)tko[n];
        q.c->b0.reserve(n * k + n * stride + n * (size_t)cap + tbytes + 64);   // seeds | seq | out bases | terminator text
        q.c->b1.reserve(n * (size_t)cap * 8 + nt * 8 + 64);                    // walk hashes | terminator hashes
        q.c->b2.reserve((n + 1) * 16 + 64);                                   // terminator offsets | terminator k-mer offsets
        q.c->b3.reserve(n * 4 + n + 64);
        uint8_t *dseed = q.c->b0.as<uint8_t>(), *dseq = dseed + n * k, *dbases = dseq + n * stride, *dterm = dbases + n * (size_t)cap;
        uint64_t *dwf = q.c->b1.as<uint64_t>(), *dtf = dwf + n * (size_t)cap;
        int64_t *dtoff = q.c->b2.as<int64_t>(), *dtko = dtoff + (n + 1);
        int32_t *dlen = q.c->b3.as<int32_t>();
        uint8_t *dreason = reinterpret_cast<uint8_t *>(dlen + n);
        RB_HIP(hipMemcpyAsync(dseed, seeds, n * k, hipMemcpyHostToDevice, s));
        if (mode == 0) {
            if (tbytes) RB_HIP(hipMemcpyAsync(dterm, term_seq + term_off[0], tbytes, hipMemcpyHostToDevice, s));
            if (nt) RB_HIP(hipMemcpyAsync(dtf, tf.data(), nt * 8, hipMemcpyHostToDevice, s));
        }
        RB_HIP(hipMemcpyAsync(dtoff, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
        RB_HIP(hipMemcpyAsync(dtko, tko.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_naive_extend<DirectCounts>, dim3(blocks_for((int64_t)n, 64)), dim3(64), 0, s, DirectCounts{g->view(0, 0)}, (int)g->stranded, g->k, direction, mode, dseed, n,
                           bound, cap, min_cov, dterm, dtoff, dtf, dtko, dseq, dbases, dwf, dlen, dreason);
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpyAsync(out_len, dlen, n * 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipMemcpyAsync(out_reason, dreason, n, hipMemcpyDeviceToHost, s));
        RB_HIP(hipMemcpyAsync(out_bases, dbases, n * (size_t)cap, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
    });
}

// ---- the same three traversals on a SHARDED graph (rb_shard_trav_*): begin -> { advance -> [all_to_all] -> rb_shard_query_serve
// -> [all_to_all back] -> absorb } until no rank has a walk left -> end.  `advance` runs this rank's walks (the kernels above with
// ReplayCounts) up to their next unanswered neighbourhood and makes the query for what they asked (slots Q_BIDX / Q_CIDX, as
// rb_shard_query_make does); `absorb` turns the owners' replies into counts and files them in the walks' caches.
}  // extern "C"
struct rb_trav {
    int kind = 0, direction = 0, mode = 0, lookahead = 0, bound = 0, cap = 0;
    float min_cov = 1.0f;
    size_t n = 0, per_walk = 0;            // per_walk: entries of out_bases (and of the hash / count rows) per walk
    bool has_targets = false;
    rb::DevBuf text, hashes, misc, state, reqw;
    uint8_t *dseed = nullptr, *dtarget = nullptr, *dseq = nullptr, *dbases = nullptr, *dterm = nullptr, *dreason = nullptr;
    uint64_t *df = nullptr, *dr = nullptr, *dtf = nullptr;
    float *dc = nullptr;
    int32_t *dlen = nullptr;
    int64_t *dtoff = nullptr, *dtko = nullptr;
    TravArrays ta{};
    uint32_t n_req = 0;
    int64_t rounds = 0;
    rb_graph *gate = nullptr;              // greedy extension's gate filter: another shard handle of this rank (its dbgbf)
};
namespace rb {
void trav_free(rb_graph *g) {
    if (!g->trav) return;
    rb_trav *t = g->trav;
    t->text.release(); t->hashes.release(); t->misc.release(); t->state.release(); t->reqw.release();
    delete t;
    g->trav = nullptr;
}
}  // namespace rb
extern "C" {

int rb_shard_trav_begin(rb_graph *g, int kind, const char *seeds, const char *targets, size_t n, int direction, int mode_or_lookahead,
                        int bound, int cap, float min_cov, const char *term_seq, const int64_t *term_off, int answer_cap) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard, "rb_shard_trav_begin: not a shard handle");
        RB_REQUIRE(kind >= 0 && kind <= 2, "rb_shard_trav_begin: kind must be 0 (max-coverage walk), 1 (greedy extension) or 2 (naive extension)");
        RB_REQUIRE(n == 0 || seeds, "rb_shard_trav_begin: null seeds");
        RB_REQUIRE(direction == 0 || direction == 1, "rb_shard_trav_begin: direction must be 0 (right) or 1 (left)");
        RB_REQUIRE(n < ((size_t)1 << 28), "rb_shard_trav_begin: too many walks in one call");
        RB_REQUIRE(g->dbg.bits && g->cbf, "rb_shard_trav_begin: this call needs dbgbf and cbf");
        int mode = 0, lookahead = 0;
        if (kind == 0) RB_REQUIRE(bound >= 1 && bound <= (1 << 20), "rb_shard_trav_begin: bound out of range [1, 2^20]");
        if (kind == 1) {
            lookahead = mode_or_lookahead;
            RB_REQUIRE(bound >= 1 && bound <= (1 << 20), "rb_shard_trav_begin: bound out of range [1, 2^20]");
            RB_REQUIRE(lookahead >= 0 && lookahead <= WALK_MAX_LOOKAHEAD, "rb_shard_trav_begin: lookahead out of range [0, %d]", WALK_MAX_LOOKAHEAD);
        }
        if (kind == 2) {
            mode = mode_or_lookahead;
            RB_REQUIRE(mode >= 0 && mode <= 2, "rb_shard_trav_begin: mode must be 0 (terminators), 1 (bounded) or 2 (bounded, no back checks)");
            RB_REQUIRE(mode == 0 ? (cap >= 1 && cap <= (1 << 20) && (n == 0 || (term_seq && term_off))) : (bound >= 0 && bound < (1 << 20)),
                       "rb_shard_trav_begin: mode 0 needs a capacity and the terminator sequences, modes 1 / 2 a bound");
            if (mode != 0) cap = bound + 1;
        }
        RB_HIP(hipSetDevice(g->p.device));
        rb::trav_free(g);
        rb_trav *t = new rb_trav();
        g->trav = t;
        t->kind = kind; t->direction = direction; t->mode = mode; t->lookahead = lookahead; t->bound = bound; t->cap = cap; t->min_cov = min_cov;
        t->n = n; t->has_targets = targets != nullptr;
        hipStream_t s = g->stream;
        const size_t k = (size_t)g->k;
        t->per_walk = kind == 2 ? (size_t)cap : (size_t)bound;
        const size_t stride = kind == 0 ? k + (size_t)bound : kind == 1 ? k + (size_t)bound + (size_t)WALK_MAX_LOOKAHEAD + 1 : k + (size_t)cap;
        const size_t rows = std::max<size_t>(n, 1) * t->per_walk;
        std::vector<int64_t> tko(1, 0), rel(1, 0);
        std::vector<uint64_t> tf(1, 0);
        size_t tbytes = 0;
        if (kind == 2) naive_terminators(k, n, mode, term_seq, term_off, tko, rel, tf, tbytes);
        const size_t nt = kind == 2 ? (size_t)tko[n] : 0;
        t->text.reserve(n * k * 2 + n * stride + rows + tbytes + 64);          // seeds | targets | seq | appended bases | terminator text
        t->dseed = t->text.as<uint8_t>(); t->dtarget = t->dseed + n * k; t->dseq = t->dtarget + n * k; t->dbases = t->dseq + n * stride; t->dterm = t->dbases + rows;
        t->hashes.reserve(rows * 16 + nt * 8 + 64);                            // forward | reverse hashes of the appended k-mers | terminator hashes
        t->df = t->hashes.as<uint64_t>(); t->dr = t->df + rows; t->dtf = t->dr + rows;
        t->misc.reserve(rows * 4 + (n + 1) * 16 + n * 4 + n + 128);            // counts | terminator offsets | lengths | reasons
        t->dc = t->misc.as<float>();
        t->dtoff = reinterpret_cast<int64_t *>(t->dc + rows + (rows & 1)); t->dtko = t->dtoff + (n + 1);
        t->dlen = reinterpret_cast<int32_t *>(t->dtko + (n + 1)); t->dreason = reinterpret_cast<uint8_t *>(t->dlen + n);
        // the walks' answer caches: a step of the max-coverage walk asks 4 counts, of the naive extension 7; a greedy step asks 4 per
        // neighbourhood its depth-first search opens (answer_cap, default 4 * (1 + 4 * (1 + 2 * lookahead)) — two open branches per level)
        uint32_t ccap = kind == 0 ? 4u : kind == 2 ? 8u : (uint32_t)(4 * (1 + 4 * (1 + 2 * std::max(lookahead, 1))));
        if (answer_cap > 0) ccap = std::max<uint32_t>((uint32_t)answer_cap, kind == 2 ? 8u : 4u);
        const size_t nn = std::max<size_t>(n, 1);
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 15) / 16 * 16; return o; };
        const size_t o_f = take(nn * 8), o_r = take(nn * 8), o_key = take(nn * ccap * 8), o_val = take(nn * ccap * 4), o_cn = take(nn * 4), o_len = take(nn * 4),
                     o_phase = take(nn), o_over = take(nn), o_ctr = take(64);
        t->state.reserve(off);
        char *base = static_cast<char *>(t->state.p);
        RB_HIP(hipMemsetAsync(base, 0, off, s));
        t->reqw.reserve(nn * 8 * 4);
        t->ta = TravArrays{reinterpret_cast<uint64_t *>(base + o_f), reinterpret_cast<uint64_t *>(base + o_r), reinterpret_cast<int32_t *>(base + o_len),
                           reinterpret_cast<uint8_t *>(base + o_phase), reinterpret_cast<uint64_t *>(base + o_key), reinterpret_cast<float *>(base + o_val),
                           reinterpret_cast<uint32_t *>(base + o_cn), ccap, reinterpret_cast<uint8_t *>(base + o_over), nullptr, t->reqw.as<uint32_t>(),
                           reinterpret_cast<uint32_t *>(base + o_ctr), (uint32_t)(nn * 8)};
        if (n) {
            RB_HIP(hipMemcpyAsync(t->dseed, seeds, n * k, hipMemcpyHostToDevice, s));
            if (targets) RB_HIP(hipMemcpyAsync(t->dtarget, targets, n * k, hipMemcpyHostToDevice, s));
            if (kind == 2) {
                if (mode == 0 && tbytes) RB_HIP(hipMemcpyAsync(t->dterm, term_seq + term_off[0], tbytes, hipMemcpyHostToDevice, s));
                if (mode == 0 && nt) RB_HIP(hipMemcpyAsync(t->dtf, tf.data(), nt * 8, hipMemcpyHostToDevice, s));
                RB_HIP(hipMemcpyAsync(t->dtoff, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
                RB_HIP(hipMemcpyAsync(t->dtko, tko.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
            }
        }
        RB_HIP(hipStreamSynchronize(s));                      // the host vectors above go out of scope
    });
}

int rb_shard_trav_set_gate(rb_graph *g, rb_graph *gate) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && g->trav, "rb_shard_trav_set_gate: no traversal in progress on this handle");
        RB_REQUIRE(g->trav->kind == 1 && g->trav->rounds == 0, "rb_shard_trav_set_gate: a gate belongs to a greedy extension that has not started");
        RB_REQUIRE(gate && gate != g && gate->shard && gate->dbg.bits && gate->p.device == g->p.device && gate->k == g->k && gate->stranded == g->stranded &&
                   gate->shard_count == g->shard_count && gate->shard_rank == g->shard_rank,
                   "rb_shard_trav_set_gate: the gate must be this rank's shard of a graph with the same k, strandedness and rank count");
        g->trav->gate = gate;
    });
}

int rb_shard_trav_advance(rb_graph *g, int64_t *n_active, int64_t *bit_counts, int64_t *ctr_counts, int64_t *gate_bit_counts) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && g->trav && n_active && bit_counts && ctr_counts, "rb_shard_trav_advance: no traversal in progress on this handle");
        RB_REQUIRE(!g->trav->gate || gate_bit_counts, "rb_shard_trav_advance: this traversal has a gate: gate_bit_counts is needed");
        rb_trav *t = g->trav;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        const size_t n = t->n;
        uint32_t ctr[2] = {0, 0};
        if (n) {
            t->ta.req = rb::shard_query_h0(g, n * 8);
            RB_HIP(hipMemsetAsync(t->ta.ctr, 0, 8, s));
            const ReplayCounts src{t->ta};
            const dim3 gr(blocks_for((int64_t)n, 64)), th(64);
            if (t->kind == 0)
                hipLaunchKernelGGL(k_walk_max_cov<ReplayCounts>, gr, th, 0, s, src, (int)g->stranded, g->k, t->direction, t->dseed,
                                   t->has_targets ? t->dtarget : (const uint8_t *)nullptr, n, t->bound, t->min_cov, t->dseq, t->dbases, t->df, t->dr, t->dc, t->dlen, t->dreason);
            else if (t->kind == 1)
                hipLaunchKernelGGL(k_greedy_extend<ReplayCounts>, gr, th, 0, s, src, kmul_of(g->k), WalkGate{nullptr, g->dbg.mod, 0}, (int)g->stranded, g->k, t->direction,
                                   t->dseed, n, t->lookahead, t->bound, t->dseq, t->dbases, t->dc, t->dlen, t->dreason);
            else
                hipLaunchKernelGGL(k_naive_extend<ReplayCounts>, gr, th, 0, s, src, (int)g->stranded, g->k, t->direction, t->mode, t->dseed, n, t->bound, t->cap, t->min_cov,
                                   t->dterm, t->dtoff, t->dtf, t->dtko, t->dseq, t->dbases, t->df, t->dlen, t->dreason);
            RB_HIP(hipGetLastError());
            RB_HIP(hipMemcpyAsync(ctr, t->ta.ctr, 8, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
            // A branchy greedy step files 4 requests per neighbourhood its lookahead opens (12-16 with two or three candidates), so few walks
            // on a rank can ask for more than the buffer holds.  The kernel drops what does not fit (p < req_cap) and leaves the walk
            // suspended at the start of its step; the answers that did fit are cached, the replayed step asks for the rest next round.
            ctr[0] = std::min(ctr[0], t->ta.req_cap);
        }
        t->n_req = ctr[0];
        *n_active = (int64_t)ctr[1];
        ++t->rounds;
        rb::shard_query_make_dev(g, 2, RB_DBGBF, t->n_req, bit_counts, ctr_counts);
        if (t->gate) {                                        // the same hashes, as a lookup in the gate's dbgbf (slot Q_BIDX of the gate handle)
            std::vector<int64_t> none((size_t)g->shard_count, 0);
            if (t->n_req) RB_HIP(hipMemcpyAsync(rb::shard_query_h0(t->gate, t->n_req), t->ta.req, (size_t)t->n_req * 8, hipMemcpyDeviceToDevice, t->gate->stream));
            rb::shard_query_make_dev(t->gate, 0, RB_DBGBF, t->n_req, gate_bit_counts, none.data());
        }
    });
}

int rb_shard_trav_absorb(rb_graph *g, const void *breply_dev, const void *creply_dev, const void *gate_breply_dev) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && g->trav, "rb_shard_trav_absorb: no traversal in progress on this handle");
        rb_trav *t = g->trav;
        if (!t->n_req) return;
        const float *ans = static_cast<const float *>(rb::shard_query_combine_dev(g, RB_DBGBF, breply_dev, creply_dev));
        const uint8_t *gate = nullptr;
        if (t->gate) {
            gate = static_cast<const uint8_t *>(rb::shard_query_combine_dev(t->gate, RB_DBGBF, gate_breply_dev, nullptr));
            RB_HIP(hipStreamSynchronize(t->gate->stream));
        }
        hipLaunchKernelGGL(k_trav_absorb, dim3(blocks_for((int64_t)t->n_req)), dim3(TPB), 0, g->stream, t->ta, ans, gate, t->n_req);
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(g->stream));
    });
}

int rb_shard_trav_end(rb_graph *g, char *out_bases, uint64_t *out_f, uint64_t *out_r, float *out_count, int32_t *out_len, uint8_t *out_reason, int64_t *rounds) {
    return guarded([&] {
        RB_REQUIRE(g && g->shard && g->trav, "rb_shard_trav_end: no traversal in progress on this handle");
        rb_trav *t = g->trav;
        RB_REQUIRE(t->n == 0 || (out_bases && out_len && out_reason), "rb_shard_trav_end: null argument");
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        const size_t n = t->n, rows = n * t->per_walk;
        if (n) {
            RB_HIP(hipMemcpyAsync(out_len, t->dlen, n * 4, hipMemcpyDeviceToHost, s));
            RB_HIP(hipMemcpyAsync(out_reason, t->dreason, n, hipMemcpyDeviceToHost, s));
            RB_HIP(hipMemcpyAsync(out_bases, t->dbases, rows, hipMemcpyDeviceToHost, s));
            if (out_f && t->kind != 1) RB_HIP(hipMemcpyAsync(out_f, t->df, rows * 8, hipMemcpyDeviceToHost, s));
            if (out_r && t->kind == 0) RB_HIP(hipMemcpyAsync(out_r, t->dr, rows * 8, hipMemcpyDeviceToHost, s));
            if (out_count && t->kind != 2) RB_HIP(hipMemcpyAsync(out_count, t->dc, rows * 4, hipMemcpyDeviceToHost, s));
            RB_HIP(hipStreamSynchronize(s));
        }
        if (rounds) *rounds = t->rounds;
        rb::trav_free(g);
    });
}

}  // extern "C"
