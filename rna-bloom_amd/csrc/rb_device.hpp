// rb_device.hpp — gfx950 device primitives for the ntHash / Bloom-dBG path.
// 64-bit integer/bit work only (no MFMA: nothing here is GEMM shaped).
// Reference citations: R/ = src/rnabloom/ of bcgsc/RNA-Bloom v2.0.1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rb {

// 2-bit base codes of the packed device format: A=0 C=1 G=2 T/U=3, complement = 3 - code.
// Seeds: R/bloom/hash/NTHash.java:39-43.
__device__ __forceinline__ uint64_t seed_of(uint32_t code) {
    // select chain compiles to v_cndmask pairs; no table memory traffic
    uint64_t s = 0x3c8bfbb395c60474ull;                  // A
    s = (code == 1) ? 0x3193c18562a02b4cull : s;         // C
    s = (code == 2) ? 0x20323ed082572324ull : s;         // G
    s = (code == 3) ? 0x295549f54be24456ull : s;         // T/U
    return s;
}
// rotation by ONE bit — the rolling step of both strands — as two funnel shifts (v_alignbit_b32) instead of the 64-bit shift,
// 32-bit shift and OR the compiler makes of the general form: the window walkers are bound by their instruction count
__device__ __forceinline__ uint64_t rotl1(uint64_t v) {
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    return ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 31) << 32) | __builtin_amdgcn_alignbit(lo, hi, 31);
}
__device__ __forceinline__ uint64_t rotr1(uint64_t v) {
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    return ((uint64_t)__builtin_amdgcn_alignbit(lo, hi, 1) << 32) | __builtin_amdgcn_alignbit(hi, lo, 1);
}
__device__ __forceinline__ uint64_t rotl(uint64_t v, uint32_t s) {
    if (__builtin_constant_p(s) && s == 1u) return rotl1(v);
    s &= 63u;
    return (v << s) | (v >> ((64u - s) & 63u));
}
__device__ __forceinline__ uint64_t rotr(uint64_t v, uint32_t s) {
    if (__builtin_constant_p(s) && s == 1u) return rotr1(v);
    s &= 63u;
    return (v >> s) | (v << ((64u - s) & 63u));
}
__host__ __device__ __forceinline__ int64_t as_signed(uint64_t v) { return (int64_t)v; }

// canonical = SIGNED min (rhVal<fhVal ? rhVal : fhVal), R/bloom/hash/NTHash.java:488,494
__device__ __forceinline__ uint64_t canonical(uint64_t f, uint64_t r) {
    return ((int64_t)r < (int64_t)f) ? r : f;
}
__device__ __forceinline__ uint64_t smin(uint64_t a, uint64_t b) {   // Math.min(long,long)
    return ((int64_t)a < (int64_t)b) ? a : b;
}

// NTM64(bVal, hVal, k, m): hVal[i] = t ^ (t >>> 27), t = bVal * (i ^ k*multiSeed)
// R/bloom/hash/NTHash.java:518-527
__host__ __device__ __forceinline__ uint64_t multi_hash(uint64_t b, uint32_t i, uint64_t kmul) {
    if (i == 0) return b;
    uint64_t t = b * ((uint64_t)i ^ kmul);
    return t ^ (t >> 27);
}
__host__ __device__ __forceinline__ uint64_t kmul_of(int k) {
    return (uint64_t)(int64_t)k * 0x90b45d39fb6da1faull;
}
// HashFunction.combineHashValues, R/bloom/hash/HashFunction.java:260-263 (sign-extended int literal)
__host__ __device__ __forceinline__ uint64_t combine(uint64_t a, uint64_t b) {
    return a ^ (b + 0xFFFFFFFF9E3779B9ull + (a << 6) + (b >> 2));
}

// getIndex: (hashVal >>> 1) % size, R/bloom/BloomFilter.java:108-111.
// Exact remainder of a 63-bit x by an arbitrary size d (Barrett with a 64-bit reciprocal R = floor(2^64 / d) and one
// correction): q' = mulhi(x, R) satisfies q - 1 <= q' <= q for q = floor(x / d), because x / d - x R / 2^64 =
// x (2^64 / d - R) / 2^64 < x / 2^64 < 1/2 for x < 2^63; so r' = x - q' d lies in [0, 2d) and one conditional subtraction
// gives x mod d.  One mulhi64 + one mul64 instead of the three + three of a 128-bit fastmod (round 2): the index arithmetic of
// every probe kernel and of the sharded engine's per-record owner test.  d = 1: R is clamped to 2^64 - 1, the result is 0.
struct Mod {
    uint64_t d, m_lo, m_hi;      // m_lo = R; m_hi unused (kept: the struct travels by value through every kernel signature)
};
__host__ inline Mod make_mod(uint64_t d) {
    Mod m;
    m.d = d;
    m.m_lo = d > 1 ? (uint64_t)(((unsigned __int128)1 << 64) / d) : ~0ull;
    m.m_hi = 0;
    return m;
}
__host__ __device__ __forceinline__ uint64_t fastmod(uint64_t x, const Mod &m) {      // x < 2^63
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t q = __umul64hi(x, m.m_lo);
#else
    const uint64_t q = (uint64_t)(((unsigned __int128)x * m.m_lo) >> 64);
#endif
    const uint64_t r = x - q * m.d;
    return r >= m.d ? r - m.d : r;
}
__device__ __forceinline__ uint64_t index_of(uint64_t h, const Mod &m) { return fastmod(h >> 1, m); }

// shared counter-based generator (see oracle/rb_oracle.c rbo_rng31): 31 uniform bits from
// (seed, op ordinal, position in read)
__host__ __device__ __forceinline__ uint32_t rng_read_state(uint64_t seed, uint64_t ordinal) {
    // per-op (per-read) part: one splitmix64 round, folded to 32 bits — hoisted out of per-window loops
    uint64_t z = seed ^ (ordinal * 0x9E3779B97F4A7C15ull);
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32) ^ (uint32_t)z;
}
__host__ __device__ __forceinline__ uint32_t rng_pos(uint32_t read_state, uint32_t pos) {
    // per-window part: murmur3 fmix32 over (state, position) — 32-bit multiplies only
    uint32_t x = read_state ^ (pos * 0x9E3779B1u);
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x >> 1;                                   // 31 uniform bits
}
__host__ __device__ __forceinline__ uint32_t rng31(uint64_t seed, uint64_t ordinal, uint32_t pos) {
    return rng_pos(rng_read_state(seed, ordinal), pos);
}
// MiniFloat.increment R/util/MiniFloat.java:31-38 on counter bytes 0..127; rnd31 replaces
// (int)(Math.random()*Integer.MAX_VALUE).  Returns the byte after the attempt.
__host__ __device__ __forceinline__ uint32_t minifloat_inc(uint32_t b, uint32_t rnd31) {
    if (b <= 7u) return b + 1u;
    if (b < 127u) {
        uint32_t s = (b >> 3) - 1u;
        if ((rnd31 & ((1u << s) - 1u)) == 0u) return b + 1u;
    }
    return b;
}
// MiniFloat.toFloat :40-45
__host__ __device__ __forceinline__ float minifloat_to_float(uint32_t b) {
    if (b <= 7u) return (float)b;
    uint32_t e = (b >> 3) - 1u;                    // 0..14
    return (float)(((b & 7u) | 8u) << e);          // exact: < 2^24
}

// k-mer owner in the sharded engine = the rank that holds the k-mer's FIRST COUNTER: idx_0 = (h0 >>> 1) % cbf_size lies in
// exactly one rank's index range [lo, hi).  With the usual sizing (dbgbf bits = cbf bytes, both from
// getExpectedSize(nk) with the same number of hash functions) idx_0 is also the k-mer's first Bloom bit, so probe 0 of both
// filters is local to the k-mer's owner and only the other probes travel.  (Round 2 split the hash space by bits 40.. of the
// hash: uniform too, but unrelated to where the filters live, so EVERY probe, claim and write was a routed request.)
// hi == 0: no ownership test (single GPU).
struct OwnRange {
    Mod mod;
    uint64_t lo, hi;
};
__device__ __forceinline__ bool own_mine(const OwnRange &o, uint64_t h0) {
    if (!o.hi) return true;
    const uint64_t idx = index_of(h0, o.mod);
    return idx >= o.lo && idx < o.hi;
}
// rank of an index for shards of `span` indices each (span = roundup64(ceil(size / G)); inv = floor(2^64 / span)): one mulhi and one fix-up
struct OwnSpan { uint64_t span, inv; };
__host__ inline OwnSpan make_own_span(uint64_t span) { OwnSpan o; o.span = span; o.inv = span > 1 ? (uint64_t)(((unsigned __int128)1 << 64) / span) : ~0ull; return o; }
__device__ __forceinline__ uint32_t own_rank_of(const OwnSpan &o, uint64_t idx) {
    uint64_t q = __umul64hi(idx, o.inv);
    if (idx - q * o.span >= o.span) ++q;
    return (uint32_t)q;
}

// ---- no-op prefilter cache (DESIGN.md §3 "no-op prefilter") ----
// 8-way set-associative table of 8-byte entries keyed by the FULL 64-bit base hash: a bucket is one
// 64-byte line (one memory request per lookup, like a direct-mapped table), bucket = low B bits of h0
// (ntHash bits are all equally mixed; the signed canonical minimum only skews the TOP bits),
// entry = (h0 >> B) << 4 | s — bucket and tag together are all 64 bits, so a match is exact.  An entry
// asserts "this k-mer is in dbgbf and the exponent (min_counter>>3)-1 of its counting-Bloom minimum
// is >= s" — counters only grow, so a stale entry stays true.  An occurrence whose draw strength is
// below s cannot change any counter and may be dropped before sorting.  (Direct-mapped, the same
// 2^28 entries missed ~25 % of the hot k-mers to slot collisions; every miss lets all occurrences
// of the k-mer through to the sort.)
struct Npf {
    unsigned long long *tab;   // nullptr => disabled
    uint32_t log2n;            // log2 of the number of entries (8 per bucket), 16..28
};
// What an entry remembers: the exponent (min counter >> 3) - 1 = 1..14 of the k-mer's counting-Bloom minimum — or that the minimum
// stands at 127, where MiniFloat.increment changes nothing any more (R/util/MiniFloat.java:32): EVERY further occurrence of such a
// k-mer is a no-op.  (Config 2's most expressed transcripts get there within the first pass; capped at exponent 7 their k-mers
// kept 1/128 of their occurrences: up to 29 000 records per k-mer and sub-batch, 7 % of all sorted records.)
constexpr uint32_t RB_EXP_SATURATED = 15u;
__host__ __device__ __forceinline__ uint32_t cache_exp(uint32_t mn) { return mn >= 127u ? RB_EXP_SATURATED : (mn >> 3) - 1u; }
__device__ __forceinline__ uint32_t npf_lookup(const Npf &c, uint64_t h0) {       // 0 = unknown, 16 = saturated
    const uint32_t B = c.log2n - 3u;
    const ulonglong2 *b = reinterpret_cast<const ulonglong2 *>(c.tab + ((h0 & ((1ull << B) - 1ull)) << 3));
    const uint64_t tag = h0 >> B;
    uint32_t s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const ulonglong2 e = b[q];   // (non-temporal loads measured 2x slower: 8 separate requests instead of one line)
        if ((e.x >> 4) == tag) { const uint32_t v = (uint32_t)(e.x & 15ull); s = v > s ? v : s; }
        if ((e.y >> 4) == tag) { const uint32_t v = (uint32_t)(e.y & 15ull); s = v > s ? v : s; }
    }
    return s == RB_EXP_SATURATED ? 16u : s;      // a saturated k-mer: no draw is strong enough (strengths stop at 15)
}
// s in 1..14.  8-byte stores are never torn; two threads racing for one slot lose one entry at worst,
// and a k-mer that ends up in two slots is harmless (lookup takes the larger exponent, both are true).
__device__ __forceinline__ bool npf_store(const Npf &c, uint64_t h0, uint32_t s) {      // true: the table changed
    const uint32_t B = c.log2n - 3u;
    unsigned long long *b = c.tab + ((h0 & ((1ull << B) - 1ull)) << 3);
    const uint64_t tag = h0 >> B;
    const uint32_t rot = (uint32_t)(h0 >> 52) & 7u;          // where the victim search starts
    uint32_t victim = rot, vmin = 16u;
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i) {
        const uint32_t q = (i + rot) & 7u;
        const unsigned long long e = __hip_atomic_load(&b[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((e >> 4) == tag && e != 0ull) {                   // already here: raise, never lower
            if ((uint32_t)(e & 15ull) >= s) return false;
            __hip_atomic_store(&b[q], (tag << 4) | (unsigned long long)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        const uint32_t es = (uint32_t)(e & 15ull);            // empty slots read 0: preferred victims
        if (es < vmin) { vmin = es; victim = q; }
    }
    __hip_atomic_store(&b[victim], (tag << 4) | (unsigned long long)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
// ---- minimizer-bucketed prefilter cache: same contract as Npf, different address ----
// The device serves ~54 G random 64-byte lines/s; one line request per window is what
// bounds the prefilter.  Consecutive k-mers of a read share their minimizer (the smallest canonical
// m-mer inside the k-mer) for ~(k-m+2)/2 windows, so a table whose bucket is chosen by the MINIMIZER
// lets a thread keep one bucket for several windows.  The bucket says nothing about h0, so exactness
// comes from the position inside it: a bucket is 16 slots of 8 bytes (two 64 B lines) seen as 8 bins of
// an A slot and a B slot.  k-mer h0 may live in the A slot of bin (h0 & 7), entry = (h0 >> 3) << 3 | s',
// or in the B slot of bin ((h0 >> 3) & 7), entry = (h0 >> 6) << 6 | (h0 & 7) << 3 | s' — the slot gives
// back three bits of h0, the entry the other 61: a match is exact, and an entry is one naturally atomic
// 8-byte word (never torn).  0 = empty; s' = min(s, 7) (a lower bound stays a lower bound).  Two
// independent candidate slots per k-mer (2-choice hashing without relocation) keep the ~5–10 k-mers of
// one minimizer apart: with both candidates taken by hotter k-mers a k-mer stays uncached (≈ 2 %); the
// first layout — both candidates in the same bin — lost 7 %, and spilling those into the hash-addressed
// table made nearly every wavefront iteration wait for a second dependent lookup (filter 134 -> 207 ms).
// A reader loads the bucket's two lines when its minimizer changes (every ~5 windows).
// Both strands of a k-mer have the same canonical m-mers, hence the same bucket.
// The minimizer-bucketed cache serves k <= 31 (ring of k - m + 1 <= 16 orders).  For 32 <= k <= 64 it was measured
// and lost to the hash-bucketed table: a minimizer of a 35..63-mer is shared by 10..24 consecutive k-mers, the 16-slot
// bucket overflows, the uncached k-mers keep all their occurrences (config 2 at k = 35 / 47 / 63: 3.5 / 4.7 / 5.0 G
// records sorted instead of 1.9 / 1.8 / 1.6 G; step 552 / 664 / 726 ms against 498 / 466 / 440 ms).
constexpr uint32_t RB_MPF_MAX_RING = 16u;
constexpr int RB_MPF_MAX_K = 31;
// k <= 21: the bucket of a k-mer is that of the minimizer (over canonical m-mers) of the whole k-mer.  22 <= k <= 63 (round 3; above 31
// through k_filter_reads_pipe<., true> only): of its MIDDLE mpf_kp(k) = 21 or 20 bases — any function of the k-mer that consecutive windows mostly
// share will do, but it has to be the same for both strands of a canonical k-mer (a suffix of the k-mer is not: reads of the two
// orientations then look a k-mer up in different buckets — measured: 3.6 G records instead of 1.9 G at k = 35), hence the middle, with
// k - kp even; and 10 k-mers per minimizer occurrence, the headline's shape, fit a bucket's 16 two-choice slots where 16 do not.
constexpr int RB_MPF_WIDE_MAX_K = 63;
#ifndef RB_MPF_KP_TARGET
#define RB_MPF_KP_TARGET 21u      // (odd.  25 / 23 / 21 / 19 / 17 measured: profiles/r03_kp.txt — narrower = fewer k-mers per bucket = fewer records, but more fetches)
#endif
__host__ __device__ __forceinline__ uint32_t mpf_kp(uint32_t k) { return k <= RB_MPF_KP_TARGET ? k : RB_MPF_KP_TARGET - ((k & 1u) ^ 1u); }
__host__ __device__ __forceinline__ uint32_t mpf_lag(uint32_t k) { return (k - mpf_kp(k)) >> 1; }      // bases between the sub-window's end and the k-mer's
struct Mpf {
    unsigned long long *tab;   // nullptr => disabled
    uint32_t log2b;            // log2 of the number of buckets (16 slots = 128 B each)
    uint32_t m;                // minimizer length, <= 16 (2 bits per base in a u32)
};
// order of a canonical m-mer among the m-mers of a k-mer: a bijective mix of its 2-bit code
__host__ __device__ __forceinline__ uint32_t mmer_order(uint32_t canon) {
    uint32_t x = canon * 0x9E3779B1u;
    x ^= x >> 15; x *= 0x85EBCA77u;
    x ^= x >> 13;
    return x;
}
__host__ __device__ __forceinline__ uint64_t mpf_bucket(const Mpf &c, uint32_t order_min) {
    uint32_t x = order_min * 0xC2B2AE3Du;            // minima are small numbers: mix again before masking
    x ^= x >> 16; x *= 0x27D4EB2Fu;
    x ^= x >> 15;
    return (uint64_t)(x & ((c.log2b >= 32u) ? 0xFFFFFFFFu : ((1u << c.log2b) - 1u)));
}
// minimizer order of the k-mer at position p of a read whose packed words start at `rw` (store side:
// the resolve stages know a k-mer by hash + one occurrence id)
__device__ __forceinline__ uint32_t window_min_order(const uint64_t *__restrict__ rw, uint32_t p, uint32_t k, uint32_t m) {
    if (k > mpf_kp(k)) { p += mpf_lag(k); k = mpf_kp(k); }            // (the minimizer of the k-mer's middle mpf_kp(k) bases)
    const uint32_t w = p >> 5, o = p & 31u;
    uint64_t lo = rw[w], hi = (o + k > 32u) ? rw[w + 1] : 0ull, h2 = (o + k > 64u) ? rw[w + 2] : 0ull;
    if (o) {                                           // bases p.. as a 192-bit stream (k <= 64: 3 words suffice)
        lo = (lo >> (2u * o)) | (hi << (64u - 2u * o));
        hi = (hi >> (2u * o)) | (h2 << (64u - 2u * o));
        h2 >>= 2u * o;
    }
    const uint32_t mmask = (m >= 16u) ? 0xFFFFFFFFu : ((1u << (2u * m)) - 1u);
    uint32_t mf = 0, mr = 0, best = 0xFFFFFFFFu;
    for (uint32_t j = 0; j < k; ++j) {
        const uint32_t code = (uint32_t)lo & 3u;
        lo = (lo >> 2) | (hi << 62); hi = (hi >> 2) | (h2 << 62); h2 >>= 2;
        mf = ((mf << 2) | code) & mmask;
        mr = (mr >> 2) | ((3u - code) << (2u * (m - 1u)));
        if (j + 1u >= m) { const uint32_t ord = mmer_order(mf < mr ? mf : mr); best = ord < best ? ord : best; }
    }
    return best;
}
__host__ __device__ __forceinline__ uint32_t mpf_slot_a(uint64_t h0) { return ((uint32_t)h0 & 7u) * 2u; }
__host__ __device__ __forceinline__ uint32_t mpf_slot_b(uint64_t h0) { return (((uint32_t)h0 >> 3) & 7u) * 2u + 1u; }
__host__ __device__ __forceinline__ unsigned long long mpf_tag_a(uint64_t h0) { return h0 >> 3; }
__host__ __device__ __forceinline__ unsigned long long mpf_tag_b(uint64_t h0) { return ((h0 >> 6) << 3) | (h0 & 7ull); }
// s in 1..14, or RB_EXP_SATURATED.  The entry has three bits for it: 1..6 = that exponent, 7 = "7 to 10", 0 = "11 or more" (the
// entry of a k-mer whose tag is 0 would then read as an empty slot: such a k-mer is simply not cached).  mpf_rank orders the codes:
// when a bucket is full the k-mer of the higher class takes the slot — an uncached k-mer keeps ALL its occurrences, and the hotter
// it is the more that costs (config 2: the records of uncached hot k-mers were what filled the buckets that do not fit LDS).
// Raise an existing entry; else take an empty candidate slot; else replace the candidate with the
// smaller exponent if ours is larger (the coldest k-mer costs the least when it misses).
constexpr uint32_t RB_MPF_TOP_EXP = 11u;
__host__ __device__ __forceinline__ uint32_t mpf_rank(uint32_t code) { return code ? code : 8u; }
__device__ __forceinline__ bool mpf_store(const Mpf &c, uint64_t bucket, uint64_t h0, uint32_t s) {
    unsigned long long *b = c.tab + (bucket << 4);
    const uint32_t sa = mpf_slot_a(h0), sb = mpf_slot_b(h0), sv = s >= RB_MPF_TOP_EXP ? 0u : (s > 7u ? 7u : s), rv = mpf_rank(sv);
    const unsigned long long na = (mpf_tag_a(h0) << 3) | sv, nb = (mpf_tag_b(h0) << 3) | sv;
    const unsigned long long ea = __hip_atomic_load(&b[sa], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long eb = __hip_atomic_load(&b[sb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ea && (ea >> 3) == mpf_tag_a(h0)) { if (mpf_rank((uint32_t)(ea & 7ull)) < rv) { __hip_atomic_store(&b[sa], na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; } return false; }
    if (eb && (eb >> 3) == mpf_tag_b(h0)) { if (mpf_rank((uint32_t)(eb & 7ull)) < rv) { __hip_atomic_store(&b[sb], nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; } return false; }
    if (!ea) { __hip_atomic_store(&b[sa], na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
    if (!eb) { __hip_atomic_store(&b[sb], nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
    {   // both taken: one cuckoo step — move an occupant to ITS other candidate slot if that one is free
        // (an entry plus its slot give the occupant's whole hash; every store writes a word that is valid for
        // the slot it goes to, so racing stores can lose an entry but never forge one)
        const uint64_t ha = ((ea >> 3) << 3) | (uint64_t)(sa >> 1);                                  // occupant of my A slot
        const uint32_t ha_b = mpf_slot_b(ha);
        if (!__hip_atomic_load(&b[ha_b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(&b[ha_b], (mpf_tag_b(ha) << 3) | (ea & 7ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b[sa], na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        const uint64_t hb = (((eb >> 6) << 3 | (uint64_t)(sb >> 1)) << 3) | ((eb >> 3) & 7ull);    // occupant of my B slot
        const uint32_t hb_a = mpf_slot_a(hb);
        if (!__hip_atomic_load(&b[hb_a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(&b[hb_a], (mpf_tag_a(hb) << 3) | (eb & 7ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b[sb], nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
    }
    {   // ... and a second step: the occupant's other slot is taken too, but ITS occupant can move on (the k-mers of one minimizer
        // crowd one bucket: up to 10 of them for 8 + 8 slots with two choices each; every extra placement is a k-mer that stops
        // sending all its occurrences through the pipeline)
        const uint64_t ha = ((ea >> 3) << 3) | (uint64_t)(sa >> 1);
        const uint32_t ha_b = mpf_slot_b(ha);
        const unsigned long long ec = __hip_atomic_load(&b[ha_b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // a B-type entry
        if (ec && ha_b != sb) {
            const uint64_t hc = (((ec >> 6) << 3 | (uint64_t)(ha_b >> 1)) << 3) | ((ec >> 3) & 7ull);
            const uint32_t hc_a = mpf_slot_a(hc);
            if (hc_a != sa && !__hip_atomic_load(&b[hc_a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&b[hc_a], (mpf_tag_a(hc) << 3) | (ec & 7ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&b[ha_b], (mpf_tag_b(ha) << 3) | (ea & 7ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&b[sa], na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return true;
            }
        }
        const uint64_t hb = (((eb >> 6) << 3 | (uint64_t)(sb >> 1)) << 3) | ((eb >> 3) & 7ull);
        const uint32_t hb_a = mpf_slot_a(hb);
        const unsigned long long ed = __hip_atomic_load(&b[hb_a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // an A-type entry
        if (ed && hb_a != sa) {
            const uint64_t hd = ((ed >> 3) << 3) | (uint64_t)(hb_a >> 1);
            const uint32_t hd_b = mpf_slot_b(hd);
            if (hd_b != sb && !__hip_atomic_load(&b[hd_b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&b[hd_b], (mpf_tag_b(hd) << 3) | (ed & 7ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&b[hb_a], (mpf_tag_a(hb) << 3) | (eb & 7ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&b[sb], nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return true;
            }
        }
    }
    const bool pick_b = mpf_rank((uint32_t)(eb & 7ull)) < mpf_rank((uint32_t)(ea & 7ull));
    if (mpf_rank((uint32_t)((pick_b ? eb : ea) & 7ull)) < rv) { __hip_atomic_store(&b[pick_b ? sb : sa], pick_b ? nb : na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
    return false;      // nothing stored: the k-mer stays uncached (its bucket is full of k-mers that are at least as hot)
}
// lookup in a bucket image held by the caller (16 words strided by `stride`)
__device__ __forceinline__ uint32_t mpf_match(const unsigned long long *bkt, uint32_t stride, uint64_t h0) {
    const unsigned long long ea = bkt[mpf_slot_a(h0) * stride], eb = bkt[mpf_slot_b(h0) * stride];
    uint32_t s = 0;                               // 0 = unknown, else a lower bound of the exponent: 1..7, or 11
    if (ea && (ea >> 3) == mpf_tag_a(h0)) { const uint32_t v = (uint32_t)(ea & 7ull); s = v ? v : RB_MPF_TOP_EXP; }
    if (eb && (eb >> 3) == mpf_tag_b(h0)) { const uint32_t v = (uint32_t)(eb & 7ull), r = v ? v : RB_MPF_TOP_EXP; s = r > s ? r : s; }
    return s;
}

// the same without a branch (k_filter_reads_pipe is bound by the NUMBER of instructions it issues, scalar and branch ones included)
__device__ __forceinline__ uint32_t mpf_match_entries(unsigned long long ea, unsigned long long eb, uint64_t h0);
__device__ __forceinline__ uint32_t mpf_match_flat(const unsigned long long *bkt, uint32_t stride, uint64_t h0) {
    return mpf_match_entries(bkt[mpf_slot_a(h0) * stride], bkt[mpf_slot_b(h0) * stride], h0);
}
// ... on the two candidate entries, wherever the caller keeps them
__device__ __forceinline__ uint32_t mpf_match_entries(unsigned long long ea, unsigned long long eb, uint64_t h0) {
    const uint32_t va = (uint32_t)(ea & 7ull), vb = (uint32_t)(eb & 7ull);
    // an entry is (tag << 3) | exponent code: it matches iff it differs from (tag << 3) in the low three bits only
    const uint32_t ha = (uint32_t)(((ea ^ (mpf_tag_a(h0) << 3)) < 8ull) & (ea != 0ull));
    const uint32_t hb = (uint32_t)(((eb ^ (mpf_tag_b(h0) << 3)) < 8ull) & (eb != 0ull));
    const uint32_t ra = ha ? (va ? va : RB_MPF_TOP_EXP) : 0u;
    const uint32_t rb = hb ? (vb ? vb : RB_MPF_TOP_EXP) : 0u;
    return ra > rb ? ra : rb;
}

// trailing-zero strength of a draw, capped at 15 (see k_strength)
__host__ __device__ __forceinline__ uint32_t draw_strength(uint32_t rnd31) {
    const uint32_t r = rnd31 | 0x8000u;
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__ffs((int)r) - 1u;
#else
    return (uint32_t)__builtin_ctz(r);
#endif
}

// bit filters are addressed as 32-bit little-endian words: bit i -> word i>>5, mask 1<<(i&31),
// which is byte i>>3, mask 1<<(i&7) of the reference layout (UnsafeBitBuffer.java:42-48).
__device__ __forceinline__ bool bit_test(const uint32_t *bits, uint64_t i) {
    return (bits[i >> 5] >> (uint32_t)(i & 31u)) & 1u;
}
__device__ __forceinline__ void bit_set(uint32_t *bits, uint64_t i) {
    uint32_t m = 1u << (uint32_t)(i & 31u);
    if (!(bits[i >> 5] & m)) atomicOr(&bits[i >> 5], m);   // test-before-set halves write traffic
}
// BloomFilter.lookup(long[]) with its early exit, R/bloom/BloomFilter.java:170-178
__device__ __forceinline__ bool bits_lookup(const uint32_t *bits, const Mod &mod, int num_hash, uint64_t kmul, uint64_t h0) {
    for (int j = 0; j < num_hash; ++j)
        if (!bit_test(bits, index_of(multi_hash(h0, (uint32_t)j, kmul), mod))) return false;
    return true;
}

}  // namespace rb
