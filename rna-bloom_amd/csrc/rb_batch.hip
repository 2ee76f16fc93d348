// rb_batch.hip — device-resident read batches: ASCII -> packed 2-bit + validity encode, synthetic
// read generation, download, and the window-hash kernels (ntHash over every usable segment).
//
// Replaces, for the hot path: FastqReader/FastaReader record fetch + the regex segmentation of
// R/RNABloom.java:572-577 (R/util/SeqUtils.java:1432-1438) + {,Canonical,ReverseComplement}
// NTHashIterator (R/bloom/hash/NTHashIterator.java:45-69 etc.).  A k-mer window is hashed iff all
// of its k bases are usable, which is exactly the set of windows the nested regex runs yield.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "rb_internal.hpp"
#include "rb_kernels.hpp"

using namespace rb;

namespace {

constexpr int TPB = 256;
inline unsigned blocks_for(int64_t n, int tpb = TPB) { return (unsigned)((n + tpb - 1) / tpb); }

// ---------------------------------------------------------------- encode ----
// PIECE: reads [r_first, r_first + n_reads) of a larger batch's tables (off, woff: the whole batch's), whose bytes start at seq[0] = base seq_base of
// the caller's array (a staging buffer holds one piece at a time: rb_packed.hip, the streamed ASCII ingest); n_words is then an upper bound for
// the grid, the piece's words are [woff[r_first], woff[r_first + n_reads)).  Otherwise a whole batch: off relative to seq[0], words [0, n_words).
template <bool PIECE>
__global__ void k_encode_ascii_t(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                 const int64_t *__restrict__ off, const uint32_t *__restrict__ woff,
                                 int64_t n_reads, int64_t n_words, int min_q,
                                 uint64_t *__restrict__ codes, uint32_t *__restrict__ valid,
                                 uint32_t *__restrict__ word_read, uint32_t *__restrict__ rnz,
                                 int64_t r_first, int64_t seq_base) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (PIECE) { w += woff[r_first]; if (w >= (int64_t)woff[r_first + n_reads]) return; }
    else if (w >= n_words) return;
    // owning read: largest r with woff[r] <= w  (reads with zero words are skipped automatically).  A piece finds it in word_read, filled for the whole
    // batch by one pass over the reads (rb_packed.hip k_packed_word_read): the 21-step search per word made a 256 M-base piece's encode 2 ms on a
    // stream whose copies wait behind it
    int64_t lo = r_first, hi = r_first + n_reads;   // invariant woff[lo] <= w < woff[hi]
    if (PIECE) lo = word_read[w];
    else
        while (hi - lo > 1) {
            int64_t mid = (lo + hi) >> 1;
            if (woff[mid] <= (uint32_t)w) lo = mid; else hi = mid;
        }
    const int64_t r = lo;
    const int64_t base0 = off[r] - seq_base, len = off[r + 1] - off[r];
    const int64_t b0 = (w - woff[r]) * 32;
    uint64_t c = 0;
    uint32_t v = 0, rz = 0;
    for (int i = 0; i < 32; ++i) {
        int64_t b = b0 + i;
        if (b >= len) break;
        uint32_t ch = seq[base0 + b];
        if (rnz) {   // reverse-strand seed = seedTab[ch & 7] (NTHash.java:30, 133-166): classes 1 T, 3 G, 4 A, 5 A, 7 C, the others 0
            const uint32_t cls = ch & 7u;
            if ((0xBAu >> cls) & 1u) {          // classes 1, 3, 4, 5, 7
                rz |= 1u << i;
                // the code whose COMPLEMENT carries that seed (for A C G T U a c g t u: the base's own code)
                const uint32_t rc = cls == 1u ? 0u : cls == 3u ? 1u : cls == 7u ? 2u : 3u;
                c |= (uint64_t)rc << (2 * i);
            }
        }
        uint32_t code = 4;
        switch (ch) {   // [ACGTU], CASE_INSENSITIVE  (R/util/SeqUtils.java:1436-1438)
            case 'A': case 'a': code = 0; break;
            case 'C': case 'c': code = 1; break;
            case 'G': case 'g': code = 2; break;
            case 'T': case 't': case 'U': case 'u': code = 3; break;
            default: break;
        }
        bool ok = code < 4;
        if (qual) {     // PHRED33.substring(minQual): '!'+minQual .. '~'  (SeqUtils.java:1426-1434)
            uint32_t q = qual[base0 + b];
            ok = ok && (q >= (uint32_t)(33 + min_q)) && (q <= (uint32_t)'~');
        }
        if (ok) {
            c |= (uint64_t)code << (2 * i);       // (with rnz: the same two bits again)
            v |= 1u << i;
        }
    }
    codes[w] = c;
    valid[w] = v;
    if (rnz) rnz[w] = rz;
    if (!PIECE) word_read[w] = (uint32_t)r;
}

}  // namespace
namespace rb {
// one piece of a streamed ASCII ingest (rb_packed.hip): reads [r_first, r_first + n_reads) of the batch whose tables off_all / woff_all are, their bytes
// in the staging arrays from the caller's base seq_base on; the grid covers words_ub >= the piece's words
void launch_encode_ascii_piece(const uint8_t *seq, const uint8_t *qual, const int64_t *off_all, const uint32_t *woff_all, int64_t r_first, int64_t n_reads,
                               int64_t seq_base, int64_t words_ub, int min_q, uint64_t *codes, uint32_t *valid, uint32_t *word_read, hipStream_t st) {
    if (n_reads <= 0 || words_ub <= 0) return;
    hipLaunchKernelGGL(k_encode_ascii_t<true>, dim3(blocks_for(words_ub)), dim3(TPB), 0, st, seq, qual, off_all, woff_all, n_reads, words_ub, min_q, codes, valid,
                       word_read, (uint32_t *)nullptr, r_first, seq_base);
}
}  // namespace rb
namespace {
__global__ void k_decode_ascii(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                               const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ woff,
                               const uint32_t *__restrict__ len, int64_t w0, int64_t nw,
                               const int64_t *__restrict__ out_off, uint32_t first_read,
                               uint8_t *__restrict__ out) {
    int64_t w = w0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= w0 + nw) return;
    uint32_t r = word_read[w];
    uint32_t b0 = (uint32_t)(w - woff[r]) * 32u, L = len[r];
    uint64_t c = codes[w];
    uint32_t v = valid[w];
    uint8_t *dst = out + out_off[r - first_read];
    for (uint32_t i = 0; i < 32 && b0 + i < L; ++i)
        dst[b0 + i] = ((v >> i) & 1u) ? (uint8_t)("ACGT"[(c >> (2 * i)) & 3]) : (uint8_t)'N';
}

// ------------------------------------------------------------- synthetic ----
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void k_synth_genome(uint64_t *g, int64_t n_words, uint64_t seed) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_words) g[w] = mix64(seed ^ ((uint64_t)w * 0xD1B54A32D192ED03ull));
}
__device__ __forceinline__ uint32_t genome_base(const uint64_t *g, int64_t i) {
    return (uint32_t)(g[i >> 5] >> (2 * (i & 31))) & 3u;
}
__global__ void k_synth_reads(const uint64_t *__restrict__ genome, const int64_t *__restrict__ tstart,
                              const int32_t *__restrict__ tlen, const float *__restrict__ cdf,
                              int n_tx, int64_t n_pairs, int L, int words_per_read, float frag_mean,
                              float frag_sd, float sub_rate, float n_rate, uint64_t seed, int64_t pair_offset,
                              int64_t total_pairs, int keep_errors, uint64_t *__restrict__ codes, uint32_t *__restrict__ valid,
                              uint32_t *__restrict__ word_read) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n_words = 2 * n_pairs * words_per_read;
    if (w >= n_words) return;
    int64_t r = w / words_per_read;
    int c = (int)(w - r * words_per_read);
    bool right = r >= n_pairs;
    int64_t p = (right ? r - n_pairs : r) + pair_offset;       // pair id within the whole set
    const int64_t rg = right ? total_pairs + p : p;             // read id within the whole set
    uint64_t s0 = mix64(seed ^ ((uint64_t)p * 0x9E3779B97F4A7C15ull));
    uint64_t s1 = mix64(s0), s2 = mix64(s1), s3 = mix64(s2);
    float u0 = (float)(s0 >> 40) * (1.0f / 16777216.0f);
    int lo = 0, hi = n_tx - 1;   // first t with cdf[t] > u0
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (cdf[mid] > u0) hi = mid; else lo = mid + 1;
    }
    int t = lo;
    float u1 = ((float)(s1 >> 40) + 0.5f) * (1.0f / 16777216.0f);
    float u2 = (float)(s2 >> 40) * (1.0f / 16777216.0f);
    float nrm = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    int flen = (int)rintf(frag_mean + frag_sd * nrm);
    int tl = tlen[t];
    flen = flen < L ? L : flen;
    flen = flen > tl ? tl : flen;
    double u3 = (double)(s3 >> 11) * (1.0 / 9007199254740992.0);
    int64_t fstart = tstart[t] + (int64_t)(u3 * (double)(tl - flen + 1));
    uint64_t cw = 0;
    uint32_t vw = 0;
    for (int i = 0; i < 32; ++i) {
        int b = c * 32 + i;
        if (b >= L) break;
        uint32_t code = right ? 3u - genome_base(genome, fstart + flen - 1 - b)
                              : genome_base(genome, fstart + b);
        uint64_t e = mix64(seed ^ 0xA5A5A5A5ull ^ ((uint64_t)rg * 0xC2B2AE3D27D4EB4Full) ^ (uint64_t)b);
        float ue = (float)(e >> 40) * (1.0f / 16777216.0f);
        float un = (float)((e >> 16) & 0xFFFFFFu) * (1.0f / 16777216.0f);
        bool ok = true;
        if (ue < sub_rate) { code = (code + 1u + (uint32_t)(e & 0xFFFFu) % 3u) & 3u; ok = keep_errors != 0; }   // quality '#' (masked) unless asked otherwise
        if (un < n_rate) ok = false;
        cw |= (uint64_t)code << (2 * i);
        if (ok) vw |= 1u << i;
    }
    codes[w] = cw;
    valid[w] = vw;
    word_read[w] = (uint32_t)r;
}

struct HostGuard {   // frees partially built batches on exceptions
    rb_batch *b;
    ~HostGuard() { if (b) rb_batch_destroy(b); }
};

void alloc_batch_arrays(rb_batch *b) {
    size_t nw = (size_t)std::max<int64_t>(b->n_words, 1), nr = (size_t)std::max<int64_t>(b->n_reads, 1);
    if (b->pool) {
        b->codes = static_cast<uint64_t *>(b->pool->get(nw * 8)); b->valid = static_cast<uint32_t *>(b->pool->get(nw * 4));
        b->word_read = static_cast<uint32_t *>(b->pool->get(nw * 4));
        b->woff = static_cast<uint32_t *>(b->pool->get((nr + 1) * 4)); b->len = static_cast<uint32_t *>(b->pool->get(nr * 4));
    } else {
        RB_HIP(hipMalloc(&b->codes, nw * 8));
        RB_HIP(hipMalloc(&b->valid, nw * 4));
        RB_HIP(hipMalloc(&b->word_read, nw * 4));
        RB_HIP(hipMalloc(&b->woff, (nr + 1) * 4));
        RB_HIP(hipMalloc(&b->len, nr * 4));
    }
    b->device_bytes = nw * 16 + (nr + 1) * 4 + nr * 4;
}

}  // namespace

// --------------------------------------------------------------- window kernels ----
namespace rb {

__global__ void k_count_windows(const uint32_t *__restrict__ valid, const uint32_t *__restrict__ word_read,
                                const uint32_t *__restrict__ woff, const uint32_t *__restrict__ len,
                                int64_t w0, int64_t nw, int span, uint32_t *__restrict__ cnt) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw) return;
    int64_t w = w0 + i;
    uint32_t r = word_read[w];
    uint32_t wr = woff[r];
    uint32_t L = len[r];
    uint32_t b0 = (uint32_t)(w - wr) * 32u;
    uint32_t n = 0;
    if ((uint64_t)b0 + (uint64_t)span <= L) {
        uint64_t bend64 = (uint64_t)b0 + 32u + (uint64_t)span - 1u;
        uint32_t bend = bend64 < L ? (uint32_t)bend64 : L;
        uint32_t run = 0, cur = 0;
        for (uint32_t b = b0; b < bend; ++b) {
            if ((b & 31u) == 0) cur = valid[wr + (b >> 5)];
            run = ((cur >> (b & 31u)) & 1u) ? run + 1u : 0u;
            n += run >= (uint32_t)span;
        }
    }
    cnt[i] = n;
}

// One thread per 32-base word of a read ("chunk"): it walks the bases of its chunk (+k-1 look-ahead),
// growing / rolling the forward and reverse-strand ntHash, and emits the base hash of every usable
// window.  Outputs of a 256-thread block are contiguous in the dense output array, so they are
// staged through LDS in slabs of HASH_SLAB records and written back as full, coalesced lines
// (direct per-thread 8-byte stores cost ~5x write amplification: 62 B/k-mer measured by WRITE_SIZE).
#ifndef RB_HASH_TPB
#define RB_HASH_TPB 64
#endif
#ifndef RB_EMIT_SLAB
#define RB_EMIT_SLAB 512
#endif
constexpr int HASH_TPB = RB_HASH_TPB;   // one wavefront per block: every lane is busy in the single slab round
constexpr uint32_t HASH_SLAB = 32u * RB_HASH_TPB;
template <int MODE>
__global__ void __launch_bounds__(HASH_TPB)
k_hash_windows(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
               const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ woff,
               const uint32_t *__restrict__ len, int64_t w0, int64_t nw, int k,
               const uint32_t *__restrict__ chunk_off, uint32_t first_read, uint32_t pos_bits,
               uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t *__restrict__ out_read,
               uint32_t *__restrict__ out_pos, int64_t nthreads, int cw, int wpr, int cpr) {
    // threads write runs of ~32 records at a stride of ~32 records: without the +1-per-32 skew all
    // lanes of a wavefront would hit the same LDS banks (64-way conflict on every ds_write)
    __shared__ uint64_t s_key[HASH_SLAB + HASH_SLAB / 32 + 1];
    __shared__ uint32_t s_val[HASH_SLAB + HASH_SLAB / 32 + 1];
    __shared__ uint32_t s_pos[HASH_SLAB + HASH_SLAB / 32 + 1];
    // roll tables: entry [out*4+in] holds everything a rolling step xors in besides the rotated hash
    //   forward: rotl(seed(out),k) ^ seed(in)            (NTHash.java:584-586)
    //   reverse: rotr(seedc(out),1) ^ rotl(seedc(in),k-1) (NTHash.java:491-495 / :627-629)
    __shared__ uint64_t s_tf[16], s_tr[16];
    const uint32_t uk = (uint32_t)k;
    if (threadIdx.x < 16) {
        const uint32_t oc = threadIdx.x >> 2, ic = threadIdx.x & 3u;
        s_tf[threadIdx.x] = rotl(seed_of(oc), uk) ^ seed_of(ic);
        s_tr[threadIdx.x] = rotr(seed_of(3u - oc), 1) ^ rotl(seed_of(3u - ic), uk - 1u);
    }
    __syncthreads();
    // thread -> chunk of `cw` consecutive words of ONE read.  Ragged batches use one word per thread
    // (cw = 1, any word); uniform batches (wpr words per read) use cw > 1 so that the k-1 warm-up bases
    // are amortised over more windows (150 bp reads: one thread per read, 150 steps for 126 windows).
    const int64_t blk0 = (int64_t)blockIdx.x * HASH_TPB;
    const int64_t t = blk0 + threadIdx.x;
    const int64_t blk_end = (blk0 + HASH_TPB < nthreads) ? blk0 + HASH_TPB : nthreads;
    auto first_word = [&](int64_t tt) -> int64_t {   // index (relative to w0) of the first word of thread tt
        if (cw <= 1) return tt;
        if (tt >= nthreads) return nw;
        return (tt / cpr) * wpr + (tt % cpr) * cw;
    };
    const uint32_t O0 = chunk_off[first_word(blk0)], O1 = chunk_off[first_word(blk_end)];   // chunk_off has nw+1 entries
    if (O0 == O1) return;
    // per-thread walker state
    uint32_t r = 0, L = 0, b = 0, bend = 0, run = 0, cur_v = 0, out = 0;
    uint64_t cur_c = 0, f = 0, rv = 0, hist = 0;   // hist: 2-bit codes of the last 32 bases
    const uint64_t *cw_ = codes;
    const uint32_t *vw = valid;
    if (t < nthreads) {
        const int64_t i = first_word(t);
        const int64_t w = w0 + i;
        r = word_read[w];
        const uint32_t wr = woff[r];
        L = len[r];
        const uint32_t b0 = (uint32_t)(w - wr) * 32u;
        cw_ = codes + wr;
        vw = valid + wr;
        b = b0;
        if ((uint64_t)b0 + uk <= L) {
            const uint64_t bend64 = (uint64_t)b0 + 32u * (uint64_t)(cw > 1 ? cw : 1) + uk - 1u;
            bend = bend64 < L ? (uint32_t)bend64 : L;
        } else bend = b0;                       // no window starts in this chunk
        out = chunk_off[i];
    }
    const bool short_k = uk <= 31u;             // the outgoing base is still in `hist`
    for (uint32_t slab0 = O0; slab0 < O1; slab0 += HASH_SLAB) {
        const uint32_t slab1 = (slab0 + HASH_SLAB < O1) ? slab0 + HASH_SLAB : O1;
        bool reload = true;                     // the walker may resume in the middle of a word
        while (b < bend && out < slab1) {
            if (reload || (b & 31u) == 0) { cur_c = cw_[b >> 5]; cur_v = vw[b >> 5]; reload = false; }
            if (!((cur_v >> (b & 31u)) & 1u)) { run = 0; f = 0; rv = 0; ++b; continue; }
            const uint32_t code = (uint32_t)(cur_c >> (2u * (b & 31u))) & 3u;
            if (run < uk) {
                // growing window: after k bases these equal NTP64 / NTP64RC from scratch
                // (R/bloom/hash/NTHash.java:332-337, 367-373)
                if (MODE != 2) f = rotl(f, 1) ^ seed_of(code);
                if (MODE != 0) rv ^= rotl(seed_of(3u - code), run);
                ++run;
            } else {
                uint32_t oc;
                if (short_k) oc = (uint32_t)(hist >> (2u * (uk - 1u))) & 3u;
                else { const uint32_t bo = b - uk; oc = (uint32_t)(cw_[bo >> 5] >> (2u * (bo & 31u))) & 3u; }
                const uint32_t t = oc * 4u + code;
                if (MODE != 2) f = rotl(f, 1) ^ s_tf[t];
                if (MODE != 0) rv = rotr(rv, 1) ^ s_tr[t];
            }
            hist = (hist << 2) | code;
            if (run >= uk) {
                const uint32_t p = b - uk + 1u;
                const uint32_t o0 = out - slab0, o = o0 + (o0 >> 5);
                s_key[o] = (MODE == 0) ? f : (MODE == 2) ? rv : canonical(f, rv);
                s_val[o] = out_read ? r : (((r - first_read) << pos_bits) | p);
                if (out_read) s_pos[o] = p;
                ++out;
            }
            ++b;
        }
        __syncthreads();
        const uint32_t n = slab1 - slab0;
        for (uint32_t j = threadIdx.x; j < n; j += HASH_TPB) {
            const uint32_t q = j + (j >> 5);
            keys[slab0 + j] = s_key[q];
            if (vals) vals[slab0 + j] = s_val[q];
            if (out_read) {
                out_read[slab0 + j] = s_val[q];
                out_pos[slab0 + j] = s_pos[q];
            }
        }
        __syncthreads();
    }
}

// The bases a word-per-lane walker consumes — the word's 32 window starts need 32 + k - 1 bases — as a stream of 2-bit
// codes / usable bits, and the history the rolling hash takes its outgoing base from.  WIDE = 32 <= k <= 64: three
// words in, 128 bits of code history; otherwise (k <= 31) two words and 64 bits, exactly the registers the
// kernels below used before this struct existed.
template <bool WIDE> struct WordWalk {
    uint64_t clo = 0, chi = 0, c3 = 0, vs = 0, hc = 0, hc2 = 0, hv = 0;
    uint32_t v3 = 0;
    __device__ __forceinline__ void load(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid, int64_t w, uint32_t c, uint32_t nwords) {
        clo = codes[w]; chi = (c + 1u < nwords) ? codes[w + 1] : 0ull;
        vs = (uint64_t)valid[w] | ((c + 1u < nwords) ? ((uint64_t)valid[w + 1] << 32) : 0ull);
        if (WIDE && c + 2u < nwords) { c3 = codes[w + 2]; v3 = valid[w + 2]; }
    }
    __device__ __forceinline__ void next(uint32_t &code, uint32_t &ok) {
        code = (uint32_t)clo & 3u; ok = (uint32_t)vs & 1u;
        if (WIDE) {
            clo = (clo >> 2) | (chi << 62); chi = (chi >> 2) | (c3 << 62); c3 >>= 2;
            vs = (vs >> 1) | ((uint64_t)(v3 & 1u) << 63); v3 >>= 1;
        } else {
            clo = (clo >> 2) | (chi << 62); chi >>= 2; vs >>= 1;
        }
    }
    // 0 = null (unusable or before the walk), 1..4 = A,C,G,T: the base k - 1 steps back (sh_c = 2(k-1), sh_v = k-1)
    __device__ __forceinline__ uint32_t out5(uint32_t sh_c, uint32_t sh_v) const {
        const uint32_t oc = (!WIDE || sh_c < 64u) ? (uint32_t)(hc >> sh_c) & 3u : (uint32_t)(hc2 >> (sh_c - 64u)) & 3u;
        return ((uint32_t)(hv >> sh_v) & 1u) ? oc + 1u : 0u;
    }
    __device__ __forceinline__ void push(uint32_t code, uint32_t ok) {
        if (WIDE) hc2 = (hc2 << 2) | (hc >> 62);
        hc = (hc << 2) | code; hv = (hv << 1) | ok;
    }
};

// Fast path for k <= 31, one 32-base word per thread, one wavefront per block.
// Branch-free walker: a base that is unusable (or lies before the chunk) is treated as a "null" base
// whose seed is 0 — it contributes nothing when it enters the window and nothing when it leaves, so
// the rolling formulas (NTHash.java:491-495, 584-586) hold from the very first step and across
// unusable bases; a window is emitted only when its last k bases were all usable (run >= k), and at
// that point f / r equal NTP64 / NTP64RC from scratch.  Roll terms come from a 5x5 LDS table
// indexed by (outgoing, incoming) in {null,A,C,G,T}.
template <int MODE, bool WIDE>
__global__ void __launch_bounds__(64)
k_hash_windows_fast(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                    const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ woff,
                    const uint32_t *__restrict__ len, int64_t w0, int64_t nw, int k,
                    const uint32_t *__restrict__ chunk_off, uint32_t first_read, uint32_t pos_bits,
                    uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, const uint32_t *__restrict__ keepmask) {
    // Records of the block's 64 words are contiguous in the output: they are staged through LDS and
    // written back as full lines, SLAB records per round.  After the prefilter ~1/5 of the windows
    // survive, so one round of 512 is the common case; a small slab keeps LDS at 6 KB per wavefront
    // (occupancy 8 instead of 2 with a 2048-record slab) and this walker is latency-bound.
    constexpr uint32_t SLAB = RB_EMIT_SLAB;
    __shared__ uint64_t s_key[SLAB + SLAB / 32 + 1];
    __shared__ uint32_t s_val[SLAB + SLAB / 32 + 1];
    __shared__ uint64_t s_tf[25], s_tr[25];
    const uint32_t uk = (uint32_t)k;
    if (threadIdx.x < 25) {
        const uint32_t o = threadIdx.x / 5u, in = threadIdx.x % 5u;   // 0 = null, 1..4 = A,C,G,T
        const uint64_t so = o ? seed_of(o - 1u) : 0ull, si = in ? seed_of(in - 1u) : 0ull;
        const uint64_t sco = o ? seed_of(4u - o) : 0ull, sci = in ? seed_of(4u - in) : 0ull;
        s_tf[threadIdx.x] = rotl(so, uk) ^ si;
        s_tr[threadIdx.x] = rotr(sco, 1) ^ rotl(sci, uk - 1u);
    }
    __syncthreads();
    const int64_t blk0 = (int64_t)blockIdx.x * 64;
    const int64_t i = blk0 + threadIdx.x;
    const int64_t blk_end = (blk0 + 64 < nw) ? blk0 + 64 : nw;
    const uint32_t O0 = chunk_off[blk0], O1 = chunk_off[blk_end];
    if (O0 == O1) return;
    // walker state (a lane without windows keeps nb = 0)
    WordWalk<WIDE> ww;                                    // incoming bases + codes / usable bits of the previous ones
    uint64_t f = 0, rv = 0;
    uint32_t nb = 0, j = 0, run = 0, out = 0, rel = 0, b0 = 0, keep = 0;
    if (i < nw) {
        const int64_t w = w0 + i;
        const uint32_t r = word_read[w], wr = woff[r], L = len[r];
        const uint32_t c = (uint32_t)(w - wr);
        b0 = c * 32u;
        keep = keepmask ? keepmask[i] : 0xFFFFFFFFu;      // bit p-b0: window p survived the prefilter
        if ((uint64_t)b0 + uk <= L && keep) {
            const uint32_t nwords = (L + 31u) >> 5;
            // 64 (96) bases of codes / validity starting at b0 (later words only if the read has them)
            ww.load(codes, valid, w, c, nwords);
            nb = ((b0 + 32u + uk - 1u < L) ? b0 + 32u + uk - 1u : L) - b0;   // bases to walk (<= 62; <= 95 WIDE)
            out = chunk_off[i] - O0;
            rel = (r - first_read) << pos_bits;
        }
    }
    const uint32_t sh_c = 2u * (uk - 1u), sh_v = uk - 1u;
    const uint32_t n = O1 - O0;
    for (uint32_t slab0 = 0; slab0 < n; slab0 += SLAB) {
        const uint32_t slab1 = (slab0 + SLAB < n) ? slab0 + SLAB : n;
        while (j < nb && out < slab1) {
            uint32_t code, ok;
            ww.next(code, ok);
            const uint32_t in5 = ok ? code + 1u : 0u;
            const uint32_t t = ww.out5(sh_c, sh_v) * 5u + in5;
            if (MODE != 2) f = rotl(f, 1) ^ s_tf[t];
            if (MODE != 0) rv = rotr(rv, 1) ^ s_tr[t];
            ww.push(code, ok);
            run = ok ? run + 1u : 0u;
            if (run >= uk && ((keep >> (j + 1u - uk)) & 1u)) {
                const uint32_t o = out - slab0, q = o + (o >> 5);
                s_key[q] = (MODE == 0) ? f : (MODE == 2) ? rv : canonical(f, rv);
                s_val[q] = rel | (b0 + j + 1u - uk);
                ++out;
            }
            ++j;
        }
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < slab1 - slab0; x += 64u) {
            const uint32_t q = x + (x >> 5);
            keys[O0 + slab0 + x] = s_key[q];
            vals[O0 + slab0 + x] = s_val[q];
        }
        __syncthreads();
    }
}

// Emit pass after the prefilter, for sparse keep masks.  In steady state ~45% of the words keep no
// window at all and most others keep 1-4, but k_hash_windows_fast walks all 64 words of a wavefront as
// long as one of them keeps something — and the walker is bound by instruction issue.  Here a
// wavefront takes RB_SPARSE_WORDS consecutive words, lists the ones that keep a window (ordered, in
// LDS) and walks the list 64 entries at a time; the records of the listed words still follow each
// other in the output, so the LDS slab / coalesced write-back is the same.
#ifndef RB_SPARSE_WORDS
#define RB_SPARSE_WORDS 512
#endif
template <int MODE, bool WIDE>
__global__ void __launch_bounds__(64)
k_hash_windows_sparse(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                      const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ woff,
                      const uint32_t *__restrict__ len, int64_t w0, int64_t nw, int k,
                      const uint32_t *__restrict__ chunk_off, uint32_t first_read, uint32_t pos_bits,
                      uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, const uint32_t *__restrict__ keepmask) {
    constexpr uint32_t SLAB = RB_EMIT_SLAB, BW = RB_SPARSE_WORDS;
    __shared__ uint64_t s_key[SLAB + SLAB / 32 + 1];
    __shared__ uint32_t s_val[SLAB + SLAB / 32 + 1];
    __shared__ uint64_t s_tf[25], s_tr[25];
    __shared__ uint16_t s_list[BW];
    const uint32_t uk = (uint32_t)k, lane = threadIdx.x;
    const int64_t blk0 = (int64_t)blockIdx.x * BW;
    const int64_t blk_end = (blk0 + BW < nw) ? blk0 + BW : nw;
    const uint32_t O0 = chunk_off[blk0], O1 = chunk_off[blk_end];
    if (O0 == O1) return;
    if (threadIdx.x < 25) {
        const uint32_t o = threadIdx.x / 5u, in = threadIdx.x % 5u;   // 0 = null, 1..4 = A,C,G,T
        const uint64_t so = o ? seed_of(o - 1u) : 0ull, si = in ? seed_of(in - 1u) : 0ull;
        const uint64_t sco = o ? seed_of(4u - o) : 0ull, sci = in ? seed_of(4u - in) : 0ull;
        s_tf[threadIdx.x] = rotl(so, uk) ^ si;
        s_tr[threadIdx.x] = rotr(sco, 1) ^ rotl(sci, uk - 1u);
    }
    uint32_t n_list = 0;                                           // uniform
    for (uint32_t q = 0; q < BW; q += 64u) {
        const int64_t i = blk0 + q + lane;
        const bool ne = i < blk_end && chunk_off[i + 1] != chunk_off[i];
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ne);
        if (ne) s_list[n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)(q + lane);
        n_list += (uint32_t)__popcll(m);
    }
    __syncthreads();
    const uint32_t sh_c = 2u * (uk - 1u), sh_v = uk - 1u;
    for (uint32_t e0 = 0; e0 < n_list; e0 += 64u) {
        const uint32_t e_last = (e0 + 64u < n_list) ? e0 + 63u : n_list - 1u;
        // output range of this round's words, relative to O0
        const uint32_t R0 = chunk_off[blk0 + s_list[e0]] - O0, R1 = chunk_off[blk0 + s_list[e_last] + 1] - O0;
        // walker state (a lane without a word keeps nb = 0)
        WordWalk<WIDE> ww;
        uint64_t f = 0, rv = 0;
        uint32_t nb = 0, j = 0, run = 0, out = 0, rel = 0, b0 = 0, keep = 0;
        if (e0 + lane < n_list) {
            const int64_t i = blk0 + s_list[e0 + lane];
            const int64_t w = w0 + i;
            const uint32_t r = word_read[w], wr = woff[r], L = len[r];
            const uint32_t c = (uint32_t)(w - wr);
            b0 = c * 32u;
            keep = keepmask[i];
            const uint32_t nwords = (L + 31u) >> 5;
            ww.load(codes, valid, w, c, nwords);
            nb = ((b0 + 32u + uk - 1u < L) ? b0 + 32u + uk - 1u : L) - b0;   // bases to walk (<= 62; <= 95 WIDE)
            out = chunk_off[i] - O0;
            rel = (r - first_read) << pos_bits;
        }
        for (uint32_t slab0 = R0; slab0 < R1; slab0 += SLAB) {
            const uint32_t slab1 = (slab0 + SLAB < R1) ? slab0 + SLAB : R1;
            while (j < nb && out < slab1) {
                uint32_t code, ok;
                ww.next(code, ok);
                const uint32_t in5 = ok ? code + 1u : 0u;
                const uint32_t t = ww.out5(sh_c, sh_v) * 5u + in5;
                if (MODE != 2) f = rotl(f, 1) ^ s_tf[t];
                if (MODE != 0) rv = rotr(rv, 1) ^ s_tr[t];
                ww.push(code, ok);
                run = ok ? run + 1u : 0u;
                if (run >= uk && ((keep >> (j + 1u - uk)) & 1u)) {
                    const uint32_t o = out - slab0, q = o + (o >> 5);
                    s_key[q] = (MODE == 0) ? f : (MODE == 2) ? rv : canonical(f, rv);
                    s_val[q] = rel | (b0 + j + 1u - uk);
                    ++out;
                }
                ++j;
            }
            __syncthreads();
            for (uint32_t x = threadIdx.x; x < slab1 - slab0; x += 64u) {
                const uint32_t q = x + (x >> 5);
                keys[O0 + slab0 + x] = s_key[q];
                vals[O0 + slab0 + x] = s_val[q];
            }
            __syncthreads();
        }
    }
}

// Prefilter pass: same walker as k_hash_windows_fast, but instead of emitting it decides for every
// usable window whether the occurrence can change anything: it is dropped iff the cache knows the
// k-mer (full 64-bit match) with counter exponent >= s and the occurrence's draw strength is < s.
// MPF = the minimizer-bucketed cache (rb_device.hpp): the walker also rolls the canonical m-mer of every
// position (order values in an LDS ring), the window's minimizer picks the bucket, and the bucket
// image (16 words, LDS) is reloaded only when the minimizer changes — every ~5 windows.
template <int MODE, bool MPF, bool WIDE>
__global__ void __launch_bounds__(64)
k_filter_windows_fast(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                      const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ woff,
                      const uint32_t *__restrict__ len, int64_t w0, int64_t nw, int k, uint32_t first_read,
                      uint32_t pos_bits, uint64_t seed, uint64_t ordinal0, Npf cache, Mpf mcache, uint32_t *__restrict__ cnt,
                      uint32_t *__restrict__ keepmask, uint32_t *__restrict__ total_spread, uint32_t dbg_flags,
                      OwnRange own) {
    __shared__ uint64_t s_tf[25], s_tr[25];
    extern __shared__ uint32_t s_ring[];                        // [k-m+1][lane], dynamic: orders of the current block of m-mers / suffix minima of the previous one
    __shared__ unsigned long long s_bkt[MPF ? 16 * 64 : 1];     // [slot][lane]: image of the current bucket
    const uint32_t uk = (uint32_t)k, lane = threadIdx.x;
    if (threadIdx.x < 25) {
        const uint32_t o = threadIdx.x / 5u, in = threadIdx.x % 5u;
        const uint64_t so = o ? seed_of(o - 1u) : 0ull, si = in ? seed_of(in - 1u) : 0ull;
        const uint64_t sco = o ? seed_of(4u - o) : 0ull, sci = in ? seed_of(4u - in) : 0ull;
        s_tf[threadIdx.x] = rotl(so, uk) ^ si;
        s_tr[threadIdx.x] = rotr(sco, 1) ^ rotl(sci, uk - 1u);
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    uint32_t kept = 0, mask = 0, total = 0;
    if (i < nw) {
        const int64_t w = w0 + i;
        const uint32_t r = word_read[w], wr = woff[r], L = len[r];
        const uint32_t c = (uint32_t)(w - wr), b0 = c * 32u;
        if ((uint64_t)b0 + uk <= L) {
            const uint32_t nwords = (L + 31u) >> 5;
            WordWalk<WIDE> ww;
            ww.load(codes, valid, w, c, nwords);
            const uint32_t nb = ((b0 + 32u + uk - 1u < L) ? b0 + 32u + uk - 1u : L) - b0;
            uint64_t f = 0, rv = 0;
            uint32_t run = 0;
            const uint32_t rstate = rng_read_state(seed, ordinal0 + (uint64_t)(r - first_read));
            const uint32_t sh_c = 2u * (uk - 1u), sh_v = uk - 1u;
            // minimizer state (MPF)
            const uint32_t um = MPF ? mcache.m : 1u, uw = mpf_kp(uk) - um + 1u;           // m-mers per k-mer (of its middle mpf_kp(k) bases)
            const uint32_t lag = MPF ? mpf_lag(uk) : 0u;                                  // ... whose last base is `lag` bases behind the k-mer's
            const uint32_t mmask = (um >= 16u) ? 0xFFFFFFFFu : ((1u << (2u * um)) - 1u);
            uint32_t mf = 0, mr = 0, blk_a = 0, blk_p = 0;               // position inside the current block of uw m-mers, its prefix minimum
            uint64_t cur_bkt = ~0ull;
            for (uint32_t j = 0; j < nb; ++j) {
                uint32_t code, ok;
                ww.next(code, ok);
                const uint32_t in5 = ok ? code + 1u : 0u;
                const uint32_t t = ww.out5(sh_c, sh_v) * 5u + in5;
                if (MODE != 2) f = rotl(f, 1) ^ s_tf[t];
                if (MODE != 0) rv = rotr(rv, 1) ^ s_tr[t];
                ww.push(code, ok);
                run = ok ? run + 1u : 0u;
                uint32_t o_cur = 0;
                if (MPF) {   // canonical m-mer ending at this base (garbage while run < m: never consulted then)
                    const uint32_t mcode = (uint32_t)(ww.hc >> (2u * lag)) & 3u;      // the base `lag` steps back (this step's base is in already)
                    mf = ((mf << 2) | mcode) & mmask;
                    mr = (mr >> 2) | ((3u - mcode) << (2u * (um - 1u)));
                    o_cur = mmer_order(mf < mr ? mf : mr);
                    s_ring[blk_a * 64u + lane] = o_cur;
                    blk_p = blk_a ? (o_cur < blk_p ? o_cur : blk_p) : o_cur;      // prefix minimum of the current block of uw positions
                }
                if (run >= uk) {
                    const uint32_t p = b0 + j + 1u - uk;
                    const uint64_t h0 = (MODE == 0) ? f : (MODE == 2) ? rv : canonical(f, rv);
                    // sharded engine: every rank walks all reads and keeps the k-mers it owns
                    if (own_mine(own, h0)) {
                        uint32_t s_known = 0;
                        if (MPF) {
                            // sliding-window minimum in O(1) (van Herk / Gil-Werman): the window's uw m-mers are the tail of the
                            // previous block (its suffix minima replaced the block's ring entries when it completed) and
                            // the head of the current block (running prefix minimum)
                            uint32_t omin = blk_p;
                            if (blk_a + 1u < uw) { const uint32_t sfx = s_ring[(blk_a + 1u) * 64u + lane]; omin = sfx < omin ? sfx : omin; }
                            const uint64_t bkt = mpf_bucket(mcache, omin);
                            if (bkt != cur_bkt) {     // new minimizer: fetch the two lines of its bucket
                                const ulonglong2 *bp = reinterpret_cast<const ulonglong2 *>(mcache.tab + (bkt << 4));
#pragma unroll
                                for (int q = 0; q < 8; ++q) { const ulonglong2 e = bp[q]; s_bkt[(2 * q) * 64 + lane] = e.x; s_bkt[(2 * q + 1) * 64 + lane] = e.y; }
                                cur_bkt = bkt;
                            }
                            s_known = mpf_match(&s_bkt[lane], 64u, h0);
                        } else if (!(dbg_flags & 1u))
                            s_known = npf_lookup(cache, h0);
                        bool keep = true;
                        if (s_known && !(dbg_flags & 2u)) keep = draw_strength(rng_pos(rstate, p)) >= s_known;
                        ++total;
                        if (keep) { ++kept; mask |= 1u << (p - b0); }
                    }
                }
                if (MPF) {
                    if (blk_a + 1u == uw) {   // block complete: turn its ring entries into suffix minima (uniform across the wavefront)
                        uint32_t sm = 0xFFFFFFFFu;
                        for (uint32_t q = uw; q-- > 0u;) { const uint32_t v = s_ring[q * 64u + lane]; sm = v < sm ? v : sm; s_ring[q * 64u + lane] = sm; }
                        blk_a = 0;
                    } else
                        ++blk_a;
                }
            }
        }
        cnt[i] = kept;
        keepmask[i] = mask;
    }
    for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o, 64);
    if (threadIdx.x == 0 && total) atomicAdd(&total_spread[16u * (blockIdx.x & 31u)], total);
}

// Emit pass that resumes from the rolling state the read-per-lane prefilter saved for every word (k_filter_reads:
// (f, r) after base word_start + k - 2, i.e. just before the word's first window ends): 32 steps per word instead of
// 32 + k - 1, and no history registers — at step t the base that leaves is base word_start - 1 + t (the word itself
// shifted by one base, its predecessor's last base in front) and the base that enters is word_start + k - 1 + t
// (the word and its successor, funnel-shifted).  A window is written iff its keep bit is set (keep bits exist only
// for usable windows).  Otherwise k_hash_windows_sparse: list of the words that keep something, 512-record LDS slab.
template <int MODE>
__global__ void __launch_bounds__(64)
k_hash_windows_resume(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                      const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ woff,
                      const uint32_t *__restrict__ len, int64_t w0, int64_t nw, int k,
                      const uint32_t *__restrict__ chunk_off, uint32_t first_read, uint32_t pos_bits,
                      uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, const uint32_t *__restrict__ keepmask,
                      const ulonglong2 *__restrict__ wstate) {
    constexpr uint32_t SLAB = RB_EMIT_SLAB, BW = RB_SPARSE_WORDS;
    __shared__ uint64_t s_key[SLAB + SLAB / 32 + 1];
    __shared__ uint32_t s_val[SLAB + SLAB / 32 + 1];
    __shared__ uint64_t s_tf[25], s_tr[25];
    __shared__ uint16_t s_list[BW];
    const uint32_t uk = (uint32_t)k, lane = threadIdx.x;
    const int64_t blk0 = (int64_t)blockIdx.x * BW;
    const int64_t blk_end = (blk0 + BW < nw) ? blk0 + BW : nw;
    const uint32_t O0 = chunk_off[blk0], O1 = chunk_off[blk_end];
    if (O0 == O1) return;
    if (threadIdx.x < 25) {
        const uint32_t o = threadIdx.x / 5u, in = threadIdx.x % 5u;   // 0 = null, 1..4 = A,C,G,T
        const uint64_t so = o ? seed_of(o - 1u) : 0ull, si = in ? seed_of(in - 1u) : 0ull;
        const uint64_t sco = o ? seed_of(4u - o) : 0ull, sci = in ? seed_of(4u - in) : 0ull;
        s_tf[threadIdx.x] = rotl(so, uk) ^ si;
        s_tr[threadIdx.x] = rotr(sco, 1) ^ rotl(sci, uk - 1u);
    }
    uint32_t n_list = 0;                                           // uniform
    for (uint32_t q = 0; q < BW; q += 64u) {
        const int64_t i = blk0 + q + lane;
        const bool ne = i < blk_end && chunk_off[i + 1] != chunk_off[i];
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ne);
        if (ne) s_list[n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)(q + lane);
        n_list += (uint32_t)__popcll(m);
    }
    __syncthreads();
    for (uint32_t e0 = 0; e0 < n_list; e0 += 64u) {
        const uint32_t e_last = (e0 + 64u < n_list) ? e0 + 63u : n_list - 1u;
        const uint32_t R0 = chunk_off[blk0 + s_list[e0]] - O0, R1 = chunk_off[blk0 + s_list[e_last] + 1] - O0;
        uint64_t ins = 0, outs = 0, f = 0, rv = 0;                 // entering / leaving codes of steps t, t+1, ... at bits 0.., 2..
        uint32_t inv = 0, outv = 0, j = 0, out = 0, rel = 0, b0 = 0, keep = 0;
        if (e0 + lane < n_list) {
            const int64_t i = blk0 + s_list[e0 + lane];
            const int64_t w = w0 + i;
            const uint32_t r = word_read[w], wr = woff[r], L = len[r];
            const uint32_t c = (uint32_t)(w - wr);
            b0 = c * 32u;
            keep = keepmask[i];
            const uint32_t nwords = (L + 31u) >> 5;
            const uint64_t cur = codes[w];
            const uint32_t curv = valid[w];
            const uint32_t pl = c ? (uint32_t)(codes[w - 1] >> 62) : 0u, plv = c ? (valid[w - 1] >> 31) : 0u;
            outs = (cur << 2) | pl;  outv = (curv << 1) | plv;                       // base b0 - 1 + t
            const uint32_t q = (uk - 1u) >> 5, sh = (uk - 1u) & 31u;                  // base b0 + k - 1 + t: q words on (k <= 63: 0 or 1)
            const uint64_t lo = q ? ((c + q < nwords) ? codes[w + q] : 0ull) : cur, hi = (c + q + 1u < nwords) ? codes[w + q + 1u] : 0ull;
            const uint32_t lov = q ? ((c + q < nwords) ? valid[w + q] : 0u) : curv, hiv = (c + q + 1u < nwords) ? valid[w + q + 1u] : 0u;
            ins = sh ? ((lo >> (2u * sh)) | (hi << (64u - 2u * sh))) : lo;
            inv = sh ? ((lov >> sh) | (hiv << (32u - sh))) : lov;
            const ulonglong2 st = wstate[i];
            f = st.x; rv = st.y;
            out = chunk_off[i] - O0;
            rel = (r - first_read) << pos_bits;
        }
        for (uint32_t slab0 = R0; slab0 < R1; slab0 += SLAB) {
            const uint32_t slab1 = (slab0 + SLAB < R1) ? slab0 + SLAB : R1;
            while (j < 32u && (keep >> j) != 0u && out < slab1) {   // nothing kept at or after window j: done
                const uint32_t in5 = (inv & 1u) ? ((uint32_t)ins & 3u) + 1u : 0u;
                const uint32_t out5 = (outv & 1u) ? ((uint32_t)outs & 3u) + 1u : 0u;
                ins >>= 2; inv >>= 1; outs >>= 2; outv >>= 1;
                const uint32_t t = out5 * 5u + in5;
                if (MODE != 2) f = rotl(f, 1) ^ s_tf[t];
                if (MODE != 0) rv = rotr(rv, 1) ^ s_tr[t];
                if ((keep >> j) & 1u) {
                    const uint32_t o = out - slab0, q = o + (o >> 5);
                    s_key[q] = (MODE == 0) ? f : (MODE == 2) ? rv : canonical(f, rv);
                    s_val[q] = rel | (b0 + j);
                    ++out;
                }
                ++j;
            }
            __syncthreads();
            for (uint32_t x = threadIdx.x; x < slab1 - slab0; x += 64u) {
                const uint32_t q = x + (x >> 5);
                keys[O0 + slab0 + x] = s_key[q];
                vals[O0 + slab0 + x] = s_val[q];
            }
            __syncthreads();
        }
    }
}

// ---- one read per lane (k <= 31, uniform batches) ----
// k_filter_windows_fast / k_hash_windows_fast give every lane one 32-base word: 32 windows for 32 + k-1
// walker steps, i.e. 1.94 steps per window at k = 25 — and the SQ counters show both kernels bound by
// instruction issue (filter: the vector ALU is busy 57% of all SIMD cycles at 3 wavefronts per SIMD),
// not by memory.  When every read of the batch has the same number of words W <= RB_READ_WORDS, a lane
// takes a whole read instead: 150 steps for the 126 windows of a 150-base read (1.19 per window), and
// all lanes of a wavefront cross their word boundaries in the same step.  The read's code / usable
// words are loaded into registers up front — a load inside the walk would stall the wavefront at every
// word boundary.  The interface is that of the one-word kernel: one count and one 32-bit keep mask per
// word (the word a window STARTS in).  Window start decided at base b: p = b + 1 - k.
// (The emit pass stays one word per lane: its records are staged through LDS in output order, and with
// a read per lane the 64 lanes of a wavefront fill a slab one after the other instead of together —
// measured 36 -> 70 ms.)
constexpr int RB_READ_WORDS = 10;     // reads of up to 320 bases (2 x 300 libraries)

template <int MODE, bool MPF>
__global__ void __launch_bounds__(64)
k_filter_reads(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
               const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ len, int64_t w0, int64_t nw, int k,
               uint32_t W_, uint32_t first_read, uint32_t pos_bits, uint64_t seed, uint64_t ordinal0, Npf cache, Mpf mcache,
               uint32_t *__restrict__ cnt, uint32_t *__restrict__ keepmask, uint32_t *__restrict__ total_spread,
               uint32_t dbg_flags, OwnRange own, ulonglong2 *__restrict__ wstate, const uint32_t *__restrict__ woff, uint32_t rd0, uint32_t n_rd) {
    __shared__ uint64_t s_tf[25], s_tr[25];
    extern __shared__ uint32_t s_ring[];                        // [k-m+1][lane], dynamic: orders of the current block of m-mers / suffix minima of the previous one
    __shared__ unsigned long long s_bkt[MPF ? 16 * 64 : 1];     // [slot][lane]: image of the current bucket
    const uint32_t uk = (uint32_t)k, lane = threadIdx.x;
    if (threadIdx.x < 25) {
        const uint32_t o = threadIdx.x / 5u, in = threadIdx.x % 5u;
        const uint64_t so = o ? seed_of(o - 1u) : 0ull, si = in ? seed_of(in - 1u) : 0ull;
        const uint64_t sco = o ? seed_of(4u - o) : 0ull, sci = in ? seed_of(4u - in) : 0ull;
        s_tf[threadIdx.x] = rotl(so, uk) ^ si;
        s_tr[threadIdx.x] = rotr(sco, 1) ^ rotl(sci, uk - 1u);
    }
    __syncthreads();
    // the lane's read: every read has W words (uniform batch), or — W_ == 0 — lane t takes read rd0 + t with its own word count
    const int64_t lane_t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool active = W_ ? lane_t * (int64_t)W_ < nw : lane_t < (int64_t)n_rd;
    const int64_t w = !active ? nw : W_ ? lane_t * (int64_t)W_ : (int64_t)woff[rd0 + (uint32_t)lane_t] - w0;   // the read's first word, relative to w0
    uint32_t total = 0;
    if (active) {
        const int64_t gw = w0 + w;
        const uint32_t r = W_ ? word_read[gw] : rd0 + (uint32_t)lane_t, L = len[r];
        const uint32_t W = W_ ? W_ : (L + 31u) >> 5;
        uint32_t done = 0;                                       // words whose count and mask are written
        if (uk <= L) {
            uint64_t carr[RB_READ_WORDS];
            uint32_t varr[RB_READ_WORDS];
#pragma unroll
            for (int q = 0; q < RB_READ_WORDS; ++q) {
                carr[q] = 0; varr[q] = 0;
                if ((uint32_t)q < W) { carr[q] = codes[gw + q]; varr[q] = valid[gw + q]; }
            }
            const uint32_t rstate = rng_read_state(seed, ordinal0 + (uint64_t)(r - first_read));
            const uint32_t sh_c = 2u * (uk - 1u), sh_v = uk - 1u;
            const uint32_t um = MPF ? mcache.m : 1u, uw = mpf_kp(uk) - um + 1u;           // m-mers per k-mer (of its middle mpf_kp(k) bases)
            const uint32_t lag = MPF ? mpf_lag(uk) : 0u;
            const uint32_t mmask = (um >= 16u) ? 0xFFFFFFFFu : ((1u << (2u * um)) - 1u);
            uint32_t mf = 0, mr = 0, blk_a = 0, blk_p = 0;       // minimizer state, see k_filter_windows_fast
            uint64_t cur_bkt = ~0ull;
            uint64_t f = 0, rv = 0, hc = 0, hv = 0, cur_c = 0;
            uint32_t run = 0, kept = 0, mask = 0, cur_v = 0;
            uint32_t b = 0;
            while (b < L) {
                if ((b & 31u) == 0u) {                           // next code word
                    const uint32_t wi = b >> 5;
#pragma unroll
                    for (int q = 0; q < RB_READ_WORDS; ++q) if ((uint32_t)q == wi) { cur_c = carr[q]; cur_v = varr[q]; }
                }
                const uint32_t pb = b + 1u - uk;                 // window start decided in this step (wraps while b + 1 < k)
                if ((pb & 31u) == 0u && (int32_t)pb > 0) {       // first window of a new word: the previous word is final
                    cnt[w + done] = kept; keepmask[w + done] = mask; ++done; kept = 0; mask = 0;
                }
                const uint32_t d1 = 32u - (b & 31u), d2 = 32u - (pb & 31u);
                uint32_t stop = b + (d1 < d2 ? d1 : d2);
                stop = stop < L ? stop : L;
#pragma nounroll
                for (; b < stop; ++b) {
                    const uint32_t code = (uint32_t)cur_c & 3u, ok = cur_v & 1u;
                    cur_c >>= 2; cur_v >>= 1;
                    const uint32_t in5 = ok ? code + 1u : 0u;
                    const uint32_t out5 = ((uint32_t)(hv >> sh_v) & 1u) ? ((uint32_t)(hc >> sh_c) & 3u) + 1u : 0u;
                    const uint32_t tt = out5 * 5u + in5;
                    if (MODE != 2) f = rotl(f, 1) ^ s_tf[tt];
                    if (MODE != 0) rv = rotr(rv, 1) ^ s_tr[tt];
                    hc = (hc << 2) | code; hv = (hv << 1) | ok;
                    run = ok ? run + 1u : 0u;
                    uint32_t o_cur = 0;
                    if (MPF) {   // canonical m-mer ending at this base (garbage while run < m: never consulted then)
                        const uint32_t mcode = (uint32_t)(hc >> (2u * lag)) & 3u;     // the base `lag` steps back (this step's base is in already)
                        mf = ((mf << 2) | mcode) & mmask;
                        mr = (mr >> 2) | ((3u - mcode) << (2u * (um - 1u)));
                        o_cur = mmer_order(mf < mr ? mf : mr);
                        s_ring[blk_a * 64u + lane] = o_cur;
                        blk_p = blk_a ? (o_cur < blk_p ? o_cur : blk_p) : o_cur;
                    }
                    if (run >= uk) {
                        const uint32_t p = b + 1u - uk;
                        const uint64_t h0 = (MODE == 0) ? f : (MODE == 2) ? rv : canonical(f, rv);
                        // sharded engine: every rank walks all reads and keeps the k-mers it owns
                        if (own_mine(own, h0)) {
                            uint32_t s_known = 0;
                            if (MPF) {
                                uint32_t omin = blk_p;           // sliding-window minimum, see k_filter_windows_fast
                                if (blk_a + 1u < uw) { const uint32_t sfx = s_ring[(blk_a + 1u) * 64u + lane]; omin = sfx < omin ? sfx : omin; }
                                const uint64_t bkt = mpf_bucket(mcache, omin);
                                if (bkt != cur_bkt) {            // new minimizer: fetch the two lines of its bucket
                                    const ulonglong2 *bp = reinterpret_cast<const ulonglong2 *>(mcache.tab + (bkt << 4));
#pragma unroll
                                    for (int q = 0; q < 8; ++q) { const ulonglong2 e = bp[q]; s_bkt[(2 * q) * 64 + lane] = e.x; s_bkt[(2 * q + 1) * 64 + lane] = e.y; }
                                    cur_bkt = bkt;
                                }
                                s_known = mpf_match(&s_bkt[lane], 64u, h0);
                            } else if (!(dbg_flags & 1u))
                                s_known = npf_lookup(cache, h0);
                            bool keep = true;
                            if (s_known && !(dbg_flags & 2u)) keep = draw_strength(rng_pos(rstate, p)) >= s_known;
                            ++total;
                            if (keep) { ++kept; mask |= 1u << (p & 31u); }
                        }
                    }
                    if (MPF) {
                        if (blk_a + 1u == uw) {   // block complete: turn its ring entries into suffix minima
                            uint32_t sm = 0xFFFFFFFFu;
                            for (uint32_t q = uw; q-- > 0u;) { const uint32_t v = s_ring[q * 64u + lane]; sm = v < sm ? v : sm; s_ring[q * 64u + lane] = sm; }
                            blk_a = 0;
                        } else
                            ++blk_a;
                    }
                }
                // the rolling state just before the first window of a word ends (the next base is word start + k - 1):
                // the emit pass resumes from it instead of re-walking the k-1 bases in front of the word's windows
                if (wstate && ((b + 1u - uk) & 31u) == 0u && b + 1u >= uk) wstate[w + ((b + 1u - uk) >> 5)] = make_ulonglong2(f, rv);
            }
            cnt[w + done] = kept; keepmask[w + done] = mask; ++done;       // the word the last window starts in
        }
        for (; done < W; ++done) { cnt[w + done] = 0; keepmask[w + done] = 0; }   // words no window starts in
    }
    for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o, 64);
    if (threadIdx.x == 0 && total) atomicAdd(&total_spread[16u * (blockIdx.x & 31u)], total);
}

// The read-per-lane prefilter with the bucket fetch taken off the critical path (RB_FILTER_PIPE=1; minimizer-bucketed cache,
// single-GPU ownership).  In k_filter_reads a wavefront waits for a bucket in nearly every step — some lane's minimizer changes —
// and the SQ counters show it: 55 % of a wavefront's life is spent waiting, the vector ALU is busy half of the time
// (profiles/r03_sq_counters).  Here a step is reordered and its window is finished one step late:
//   minimizer of base b -> bucket of window b  |  finish window b-1 (bucket image, match, draw)  |  ISSUE the loads of window
//   b's bucket into registers if it changed  |  roll the hashes of base b  |  (next step) ... finish window b: registers -> LDS.
// The loads are in flight while the wavefront rolls its hashes and runs the next step's minimizer update (and the other
// wavefronts of the SIMD theirs); nothing else changes: same cache, same draws, same counts and masks per word.
template <int MODE, bool WIDE>
__global__ void __launch_bounds__(64)
k_filter_reads_pipe(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                    const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ len, int64_t w0, int64_t nw, int k,
                    uint32_t W_, uint32_t first_read, uint32_t pos_bits, uint64_t seed, uint64_t ordinal0, Mpf mcache,
                    uint32_t *__restrict__ cnt, uint32_t *__restrict__ keepmask, uint32_t *__restrict__ total_spread,
                    uint32_t dbg_flags, ulonglong2 *__restrict__ wstate, const uint32_t *__restrict__ woff, uint32_t rd0, uint32_t n_rd) {
    __shared__ uint64_t s_tf[25], s_tr[25];
    extern __shared__ uint32_t s_ring[];                        // [k-m+1][lane]
    __shared__ unsigned long long s_bkt[16 * 64];               // [slot][lane]: image of the current bucket
    const uint32_t uk = (uint32_t)k, lane = threadIdx.x;
    if (threadIdx.x < 25) {
        const uint32_t o = threadIdx.x / 5u, in = threadIdx.x % 5u;
        const uint64_t so = o ? seed_of(o - 1u) : 0ull, si = in ? seed_of(in - 1u) : 0ull;
        const uint64_t sco = o ? seed_of(4u - o) : 0ull, sci = in ? seed_of(4u - in) : 0ull;
        s_tf[threadIdx.x] = rotl(so, uk) ^ si;
        s_tr[threadIdx.x] = rotr(sco, 1) ^ rotl(sci, uk - 1u);
    }
    __syncthreads();
    const int64_t lane_t = (int64_t)blockIdx.x * 64 + threadIdx.x;          // (lane -> read as in k_filter_reads)
    const bool active = W_ ? lane_t * (int64_t)W_ < nw : lane_t < (int64_t)n_rd;
    const int64_t w = !active ? nw : W_ ? lane_t * (int64_t)W_ : (int64_t)woff[rd0 + (uint32_t)lane_t] - w0;
    uint32_t total = 0;
    if (active) {
        const int64_t gw = w0 + w;
        const uint32_t r = W_ ? word_read[gw] : rd0 + (uint32_t)lane_t, L = len[r];
        const uint32_t W = W_ ? W_ : (L + 31u) >> 5;
        uint32_t done = 0;
        if (uk <= L) {
            uint64_t carr[RB_READ_WORDS];
            uint32_t varr[RB_READ_WORDS];
#pragma unroll
            for (int q = 0; q < RB_READ_WORDS; ++q) {
                carr[q] = 0; varr[q] = 0;
                if ((uint32_t)q < W) { carr[q] = codes[gw + q]; varr[q] = valid[gw + q]; }
            }
            const uint32_t rstate = rng_read_state(seed, ordinal0 + (uint64_t)(r - first_read));
            // WIDE (32 <= k <= 63): 64 bases of history in two words, and the minimizer is that of the k-mer's last 31 bases
            const uint32_t sh_c = 2u * ((uk - 1u) & 31u), sh_v = uk - 1u;
            const bool far = WIDE && uk > 32u;                   // the outgoing base sits in the older history word
            const uint32_t um = mcache.m, uw = mpf_kp(uk) - um + 1u;
            const uint32_t lag = mpf_lag(uk);                  // the minimizer's sub-window ends `lag` bases before the k-mer does (0 for k <= 25, at most 19)
            const uint32_t mmask = (um >= 16u) ? 0xFFFFFFFFu : ((1u << (2u * um)) - 1u);
            uint32_t mf = 0, mr = 0, blk_a = 0, blk_p = 0;
            uint32_t cur_bkt = ~0u;                               // (bucket numbers have at most 25 bits)
            uint64_t f = 0, rv = 0, hc = 0, hc2 = 0, hvw = 0, cur_c = 0;
            uint32_t run = 0, kept = 0, mask = 0, cur_v = 0, hv = 0;     // (k <= 31: 32 bits of usable-base history suffice; WIDE: hvw)
            uint32_t b = 0;
            // the window whose decision is pending: its hash, its position, whether its bucket is still in the registers
            ulonglong2 R[8];
            uint64_t pend_h0 = 0;
            uint32_t pend_p = 0;
            bool pend = false, pend_sw = false;
            auto land = [&]() {                                  // the pending window's bucket: registers -> LDS image
                if (pend && pend_sw) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { s_bkt[(2 * q) * 64 + lane] = R[q].x; s_bkt[(2 * q + 1) * 64 + lane] = R[q].y; }
                }
                pend_sw = false;
            };
            const uint32_t no_drop = (dbg_flags & 2u) ? 1u : 0u;
            auto decide = [&]() {                                // ... and its lookup, draw and bookkeeping — straight-line: a lane without a
                const uint32_t s_known = mpf_match_flat(&s_bkt[lane], 64u, pend_h0);     // pending window computes on stale values and adds zero
                const uint32_t strength = draw_strength(rng_pos(rstate, pend_p));
                const uint32_t on = pend ? 1u : 0u;                                     // (bitwise: no short circuit, no branch)
                const uint32_t keep = on & ((s_known == 0u ? 1u : 0u) | no_drop | (strength >= s_known ? 1u : 0u));
                total += on;
                kept += keep;
                mask |= keep << (pend_p & 31u);
                pend = false;
            };
            auto finish = [&]() { land(); decide(); };
            while (b < L) {
                if ((b & 31u) == 0u) {
                    const uint32_t wi = b >> 5;
#pragma unroll
                    for (int q = 0; q < RB_READ_WORDS; ++q) if ((uint32_t)q == wi) { cur_c = carr[q]; cur_v = varr[q]; }
                }
                const uint32_t pb = b + 1u - uk;
                if ((pb & 31u) == 0u && (int32_t)pb > 0) {       // first window of a new word: the previous word is final once its last window is
                    finish();
                    cnt[w + done] = kept; keepmask[w + done] = mask; ++done; kept = 0; mask = 0;
                }
                const uint32_t d1 = 32u - (b & 31u), d2 = 32u - (pb & 31u);
                uint32_t stop = b + (d1 < d2 ? d1 : d2);
                stop = stop < L ? stop : L;
#pragma nounroll
                for (; b < stop; ++b) {
                    const uint32_t code = (uint32_t)cur_c & 3u, ok = cur_v & 1u;
                    cur_c >>= 2; cur_v >>= 1;
                    run = ok ? run + 1u : 0u;
                    // minimizer of the window that ends at this base (garbage while run < m: never consulted then)
                    // (WIDE: the base `lag` steps back, out of the history — base b-1 sits in bits 0..1 of hc before this step's push)
                    const uint32_t mcode = lag ? (uint32_t)(hc >> (2u * (lag - 1u))) & 3u : code;
                    mf = ((mf << 2) | mcode) & mmask;
                    mr = (mr >> 2) | ((3u - mcode) << (2u * (um - 1u)));
                    const uint32_t o_cur = mmer_order(mf < mr ? mf : mr);
                    s_ring[blk_a * 64u + lane] = o_cur;
                    blk_p = blk_a ? (o_cur < blk_p ? o_cur : blk_p) : o_cur;
                    const bool win = run >= uk;
                    // (computed for every lane: while run < k the minimum is garbage and `win` keeps it from being used)
                    const uint32_t nxt = blk_a + 1u < uw ? blk_a + 1u : blk_a;          // the block's last slot has no suffix to look at
                    const uint32_t sfx = s_ring[nxt * 64u + lane];
                    const uint32_t omin = (blk_a + 1u < uw && sfx < blk_p) ? sfx : blk_p;
                    const uint32_t bkt = (uint32_t)mpf_bucket(mcache, omin);
                    land();                                      // the window before: its bucket has had a step to arrive
                    const bool sw = win && bkt != cur_bkt;
                    if (sw) {                                    // new minimizer: the two lines of its bucket, into registers
                        const ulonglong2 *bp = reinterpret_cast<const ulonglong2 *>(mcache.tab + ((uint64_t)bkt << 4));
#pragma unroll
                        for (int q = 0; q < 8; ++q) R[q] = bp[q];
                        cur_bkt = bkt;
                    }
                    decide();
                    const uint32_t in5 = ok ? code + 1u : 0u;
                    const uint32_t ok_out = WIDE ? (uint32_t)(hvw >> sh_v) & 1u : (hv >> sh_v) & 1u;
                    const uint32_t out5 = ok_out ? ((uint32_t)((far ? hc2 : hc) >> sh_c) & 3u) + 1u : 0u;
                    const uint32_t tt = out5 * 5u + in5;
                    if (MODE != 2) f = rotl(f, 1) ^ s_tf[tt];
                    if (MODE != 0) rv = rotr(rv, 1) ^ s_tr[tt];
                    if (WIDE) { hc2 = (hc2 << 2) | (hc >> 62); hvw = (hvw << 1) | ok; } else hv = (hv << 1) | ok;
                    hc = (hc << 2) | code;
                    pend = win; pend_sw = sw;
                    pend_h0 = (MODE == 0) ? f : (MODE == 2) ? rv : canonical(f, rv); pend_p = b + 1u - uk;   // (unused unless pend)
                    if (blk_a + 1u == uw) {   // block complete: turn its ring entries into suffix minima
                        uint32_t sm = 0xFFFFFFFFu;
                        for (uint32_t q = uw; q-- > 0u;) { const uint32_t v = s_ring[q * 64u + lane]; sm = v < sm ? v : sm; s_ring[q * 64u + lane] = sm; }
                        blk_a = 0;
                    } else
                        ++blk_a;
                }
                if (wstate && ((b + 1u - uk) & 31u) == 0u && b + 1u >= uk) wstate[w + ((b + 1u - uk) >> 5)] = make_ulonglong2(f, rv);
            }
            finish();
            cnt[w + done] = kept; keepmask[w + done] = mask; ++done;
        }
        for (; done < W; ++done) { cnt[w + done] = 0; keepmask[w + done] = 0; }
    }
    for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o, 64);
    if (threadIdx.x == 0 && total) atomicAdd(&total_spread[16u * (blockIdx.x & 31u)], total);
}

// The same walk with the bucket fetch done by the WAVEFRONT instead of by the lane that wants it (RB_FILTER_PIPE=3).
// k_filter_reads_pipe issues eight 16-byte loads under the mask of the ~10 lanes (15 %) whose minimizer changed in a step, in nearly every step; the texture
// path pays for a vector-memory instruction by its lanes' distinct lines, not by its bytes, and that — not the ALU, not HBM — bounds the kernel
// (tools/microbench/bucket_fetch.hip: 28.6 G buckets/s for that shape against 56 G/s for the one below; the kernel ran at 21.6 G/s).  Here the lanes that
// want a bucket post (bucket, lane) to a list in LDS and every group of four lanes fetches one posted bucket with two 16-byte loads per lane — a 64-byte
// segment per group and instruction — for whoever asked; a step later the helpers store what arrived into the asker's image.  Two vector-memory
// instructions a step instead of eight, 8 staging registers instead of 32.  For the helpers to exist every lane has to be in the loop: the step counter is
// wave-uniform (a scalar), a lane whose read is shorter idles along with `live` off.  More than 16 askers in a step (the first window of a read: all 64)
// are served in place after the lookup, sixteen at a time.  LDS is cut to 10 192 bytes for a fourth wavefront per SIMD: the images have no padding (slot s
// of lane l sits at 16 l + (s ^ 2 (l & 7)): the XOR spreads the helpers' 16-byte stores and the lookups over the banks), the list has 16 entries.
// Same cache, same draws, same counts and masks per word (RB_FILTER_CHECK=1 runs the other two walkers beside this one and compares every word).
template <int MODE, bool WIDE>
__global__ void __launch_bounds__(64)
k_filter_reads_coop(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
                    const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ len, int64_t w0, int64_t nw, int k,
                    uint32_t W_, uint32_t first_read, uint32_t pos_bits, uint64_t seed, uint64_t ordinal0, Mpf mcache,
                    uint32_t *__restrict__ cnt, uint32_t *__restrict__ keepmask, uint32_t *__restrict__ total_spread,
                    uint32_t dbg_flags, ulonglong2 *__restrict__ wstate, const uint32_t *__restrict__ woff, uint32_t rd0, uint32_t n_rd) {
    __shared__ uint64_t s_tf[25], s_tr[25];
    extern __shared__ uint32_t s_ring[];                        // [k-m+1][lane]
    __shared__ __attribute__((aligned(16))) unsigned long long s_img[64 * 16];
    __shared__ uint32_t s_list[16];                             // up to 16 askers of this step: bucket | lane << 26 (bucket numbers have at most 26 bits)
    // A workgroup is ONE wavefront (launch bounds 64, launched with 64 threads): lanes hand data to each other through LDS with no s_barrier, only
    // wave-level ordering (__builtin_amdgcn_wave_barrier below) — a second wavefront in the group would break the s_list / s_img hand-over.
    const uint32_t uk = (uint32_t)k, lane = threadIdx.x;
    if (threadIdx.x < 25) {
        const uint32_t o = threadIdx.x / 5u, in = threadIdx.x % 5u;
        const uint64_t so = o ? seed_of(o - 1u) : 0ull, si = in ? seed_of(in - 1u) : 0ull;
        const uint64_t sco = o ? seed_of(4u - o) : 0ull, sci = in ? seed_of(4u - in) : 0ull;
        s_tf[threadIdx.x] = rotl(so, uk) ^ si;
        s_tr[threadIdx.x] = rotr(sco, 1) ^ rotl(sci, uk - 1u);
    }
    __syncthreads();
    const int64_t lane_t = (int64_t)blockIdx.x * 64 + threadIdx.x;          // (lane -> read as in k_filter_reads)
    const bool active = W_ ? lane_t * (int64_t)W_ < nw : lane_t < (int64_t)n_rd;
    const int64_t w = !active ? nw : W_ ? lane_t * (int64_t)W_ : (int64_t)woff[rd0 + (uint32_t)lane_t] - w0;
    uint32_t L = 0, W = 0, r = 0;
    if (active) {
        r = W_ ? word_read[w0 + w] : rd0 + (uint32_t)lane_t; L = len[r];
        W = W_ ? W_ : (L + 31u) >> 5;
    }
    const bool walk = active && uk <= L;
    const uint32_t Lw = walk ? L : 0u;                          // the bases this lane walks
    uint32_t Lmax = Lw;
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)Lmax, o, 64); Lmax = v > Lmax ? v : Lmax; }
    Lmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)Lmax);
    uint64_t carr[RB_READ_WORDS];
    uint32_t varr[RB_READ_WORDS];
#pragma unroll
    for (int q = 0; q < RB_READ_WORDS; ++q) {
        carr[q] = 0; varr[q] = 0;
        if (walk && (uint32_t)q < W) { carr[q] = codes[w0 + w + q]; varr[q] = valid[w0 + w + q]; }
    }
    const uint32_t rstate = rng_read_state(seed, ordinal0 + (uint64_t)(r - first_read));
    // WIDE (32 <= k <= 63): 64 bases of history in two words, and the minimizer is that of the k-mer's last 31 bases
    const uint32_t sh_c = 2u * ((uk - 1u) & 31u), sh_v = uk - 1u;
    const bool far = WIDE && uk > 32u;                   // the outgoing base sits in the older history word
    const uint32_t um = mcache.m, uw = mpf_kp(uk) - um + 1u;
    const uint32_t lag = mpf_lag(uk);                  // the minimizer's sub-window ends `lag` bases before the k-mer does (0 for k <= 25, at most 19)
    const uint32_t mmask = (um >= 16u) ? 0xFFFFFFFFu : ((1u << (2u * um)) - 1u);
    uint32_t mf = 0, mr = 0, blk_a = 0, blk_p = 0;
    uint32_t cur_bkt = ~0u;
    uint64_t f = 0, rv = 0, hc = 0, hc2 = 0, hvw = 0, cur_c = 0;
    uint32_t run = 0, kept = 0, mask = 0, cur_v = 0, hv = 0, total = 0, done = 0;
    // what this lane fetched for somebody in the step before: the asker (64 = nobody) and its quarter of the two 64-byte halves of the bucket
    const uint32_t grp = lane >> 2, sub = lane & 3u;
    const ulonglong2 *const tab2 = reinterpret_cast<const ulonglong2 *>(mcache.tab) + sub;
    uint32_t tgt = 64u;
    ulonglong2 V0 = make_ulonglong2(0, 0), V1 = make_ulonglong2(0, 0);
    uint64_t pend_h0 = 0;
    bool pend = false;
    const uint32_t no_drop = (dbg_flags & 2u) ? 1u : 0u;
    const unsigned long long *img = &s_img[lane * 16u];
    const uint32_t swz = (lane & 7u) << 1;
    // 16-byte granules `sub` and `4 + sub` of asker t's bucket -> its image (stored as the u64 words the lookups read)
    auto put = [&](uint32_t t, const ulonglong2 &x0, const ulonglong2 &x1) {
        unsigned long long *d = &s_img[t * 16u + ((sub ^ (t & 7u)) << 1)];
        d[0] = x0.x; d[1] = x0.y;
        unsigned long long *e = &s_img[t * 16u + (((sub ^ (t & 7u)) ^ 4u) << 1)];
        e[0] = x1.x; e[1] = x1.y;
    };
    // the pending window (position p, the one that ended a base ago) on its two candidate entries: draw and bookkeeping — straight-line: a lane without one
    // computes on stale values and adds zero
    auto decide = [&](unsigned long long ea, unsigned long long eb, uint32_t p) {
        const uint32_t s_known = mpf_match_entries(ea, eb, pend_h0);
        const uint32_t strength = draw_strength(rng_pos(rstate, p));
        const uint32_t on = pend ? 1u : 0u;
        const uint32_t keep = on & ((s_known == 0u ? 1u : 0u) | no_drop | (strength >= s_known ? 1u : 0u));
        total += on;
        kept += keep;
        mask |= keep << (p & 31u);
        pend = false;
    };
    auto step = [&](const uint32_t b) {
        const bool live = b < Lw;
        const uint32_t pb = b + 1u - uk;
        const uint32_t code = (uint32_t)cur_c & 3u, ok = live ? (cur_v & 1u) : 0u;
        cur_c >>= 2; cur_v >>= 1;
        run = ok ? run + 1u : 0u;
        // (the roll tables' entries are asked for here, the ring's suffix minimum and the pending window's candidate slots below: LDS reads ahead of the
        // arithmetic that needs them)
        const uint32_t in5 = ok ? code + 1u : 0u;
        const uint32_t ok_out = WIDE ? (uint32_t)(hvw >> sh_v) & 1u : (hv >> sh_v) & 1u;
        const uint32_t out5 = ok_out ? ((uint32_t)((far ? hc2 : hc) >> sh_c) & 3u) + 1u : 0u;
        const uint32_t tt = out5 * 5u + in5;
        const uint64_t tf_v = (MODE != 2) ? s_tf[tt] : 0ull, tr_v = (MODE != 0) ? s_tr[tt] : 0ull;
        // minimizer of the window that ends at this base (garbage while run < m: never consulted then)
        const uint32_t mcode = lag ? (uint32_t)(hc >> (2u * (lag - 1u))) & 3u : code;
        mf = ((mf << 2) | mcode) & mmask;
        mr = (mr >> 2) | ((3u - mcode) << (2u * (um - 1u)));
        const uint32_t o_cur = mmer_order(mf < mr ? mf : mr);
        s_ring[blk_a * 64u + lane] = o_cur;
        blk_p = blk_a ? (o_cur < blk_p ? o_cur : blk_p) : o_cur;
        const bool win = run >= uk;
        const uint32_t nxt = blk_a + 1u < uw ? blk_a + 1u : blk_a;          // the block's last slot has no suffix to look at
        const uint32_t sfx = s_ring[nxt * 64u + lane];
        if (tgt < 64u) put(tgt, V0, V1);                 // the step before's buckets have had a step to arrive
        tgt = 64u;
        __builtin_amdgcn_wave_barrier();                 // helpers stored into OTHER lanes' images: the lookups below must not move above those stores
        const unsigned long long ea = img[mpf_slot_a(pend_h0) ^ swz], eb = img[mpf_slot_b(pend_h0) ^ swz];      // the pending window's candidates
        const uint32_t omin = (blk_a + 1u < uw && sfx < blk_p) ? sfx : blk_p;
        const uint32_t bkt = (uint32_t)mpf_bucket(mcache, omin);
        const bool sw = win && bkt != cur_bkt;
        const unsigned long long asks = __ballot(sw);
        uint32_t n_ask = 0, rank = 0;
        if (asks) {                                      // (wave-uniform)
            rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(asks >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)asks, 0u));
            if (sw) { if (rank < 16u) s_list[rank] = bkt | (lane << 26); cur_bkt = bkt; }
            n_ask = (uint32_t)__popcll(asks);
            __builtin_amdgcn_wave_barrier();             // the askers' posts above are read by other lanes below: keep the LDS accesses in this order
            if (grp < n_ask) {
                const uint32_t ent = s_list[grp];
                const ulonglong2 *bp = tab2 + ((size_t)(ent & 0x3FFFFFFu) << 3);
                V0 = bp[0]; V1 = bp[4];
                tgt = ent >> 26;
            }
        }
        decide(ea, eb, pb - 1u);
        for (uint32_t e0 = 16u; e0 < n_ask; e0 += 16u) {   // a crowded step: the askers beyond the sixteenth, in place (their lookups are a step away)
            __builtin_amdgcn_wave_barrier();             // (the list's entries of the round before have been read)
            if (sw && rank >= e0 && rank < e0 + 16u) s_list[rank - e0] = bkt | (lane << 26);
            __builtin_amdgcn_wave_barrier();
            if (e0 + grp < n_ask) {
                const uint32_t ent = s_list[grp];
                const ulonglong2 *bp = tab2 + ((size_t)(ent & 0x3FFFFFFu) << 3);
                const ulonglong2 x0 = bp[0], x1 = bp[4];
                put(ent >> 26, x0, x1);
            }
        }
        if (MODE != 2) f = rotl(f, 1) ^ tf_v;
        if (MODE != 0) rv = rotr(rv, 1) ^ tr_v;
        if (WIDE) { hc2 = (hc2 << 2) | (hc >> 62); hvw = (hvw << 1) | ok; } else hv = (hv << 1) | ok;
        hc = (hc << 2) | code;
        pend = win;
        pend_h0 = (MODE == 0) ? f : (MODE == 2) ? rv : canonical(f, rv);   // (unused unless pend)
        if (blk_a + 1u == uw) {   // block complete: turn its ring entries into suffix minima
            uint32_t sm = 0xFFFFFFFFu;
            for (uint32_t q = uw; q-- > 0u;) { const uint32_t v = s_ring[q * 64u + lane]; sm = v < sm ? v : sm; s_ring[q * 64u + lane] = sm; }
            blk_a = 0;
        } else
            ++blk_a;
    };
    uint32_t b = 0;                                        // the same in every lane: scalar bookkeeping
    while (b < Lmax) {
        if ((b & 31u) == 0u) {
            const uint32_t wi = b >> 5;
#pragma unroll
            for (int q = 0; q < RB_READ_WORDS; ++q) if ((uint32_t)q == wi) { cur_c = carr[q]; cur_v = varr[q]; }
        }
        const uint32_t pb = b + 1u - uk;
        if ((pb & 31u) == 0u && (int32_t)pb > 0) {       // the window of this base starts a new word: the word before is final once its last window is decided
            if (tgt < 64u) put(tgt, V0, V1);
            tgt = 64u;
            decide(img[mpf_slot_a(pend_h0) ^ swz], img[mpf_slot_b(pend_h0) ^ swz], pb - 1u);
            if (b < Lw) { cnt[w + done] = kept; keepmask[w + done] = mask; ++done; kept = 0; mask = 0; }
        }
        const uint32_t d1 = 32u - (b & 31u), d2 = 32u - (pb & 31u);
        uint32_t stop = b + (d1 < d2 ? d1 : d2);
        stop = stop < Lmax ? stop : Lmax;
#pragma nounroll
        for (; b < stop; ++b) step(b);
        // the rolling state just before the first window of a word ends (the next base is word start + k - 1): the emit pass resumes from it
        if (wstate && ((b + 1u - uk) & 31u) == 0u && b + 1u >= uk && b <= Lw) wstate[w + ((b + 1u - uk) >> 5)] = make_ulonglong2(f, rv);
    }
    if (tgt < 64u) put(tgt, V0, V1);                     // the last window
    decide(img[mpf_slot_a(pend_h0) ^ swz], img[mpf_slot_b(pend_h0) ^ swz], Lmax - uk);
    if (active) {
        if (walk) { cnt[w + done] = kept; keepmask[w + done] = mask; ++done; }       // the word the last window starts in
        for (; done < W; ++done) { cnt[w + done] = 0; keepmask[w + done] = 0; }       // words no window starts in
    }
    for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o, 64);
    if (threadIdx.x == 0 && total) atomicAdd(&total_spread[16u * (blockIdx.x & 31u)], total);
}

void launch_count_windows(const rb_batch *b, int64_t w0, int64_t nw, int span, uint32_t *cnt, hipStream_t s) {
    if (nw <= 0) return;
    hipLaunchKernelGGL(k_count_windows, dim3(blocks_for(nw)), dim3(TPB), 0, s, b->valid, b->word_read,
                       b->woff, b->len, w0, nw, span, cnt);
}
void launch_hash_windows(const rb_batch *b, int64_t w0, int64_t nw, int k, int mode,
                         const uint32_t *chunk_off, uint32_t first_read, uint32_t pos_bits,
                         uint64_t *keys, uint32_t *vals, uint32_t *out_read, uint32_t *out_pos,
                         hipStream_t s) {
    if (nw <= 0) return;
    if (k <= 31 && vals && !out_read && !getenv("RB_HASH_GENERIC")) {   // fast path (stage-1 insert at k <= 31)
        dim3 g(blocks_for(nw, 64)), t(64);
#define RB_LAUNCH_FAST(M)                                                                             \
    hipLaunchKernelGGL((k_hash_windows_fast<M, false>), g, t, 0, s, b->codes, b->valid, b->word_read, b->woff, \
                       b->len, w0, nw, k, chunk_off, first_read, pos_bits, keys, vals, (const uint32_t *)nullptr)
        if (mode == 0) RB_LAUNCH_FAST(0); else if (mode == 2) RB_LAUNCH_FAST(2); else RB_LAUNCH_FAST(1);
#undef RB_LAUNCH_FAST
        return;
    }
    int cw = 1, wpr = 0, cpr = 0;
    int64_t nthreads = nw;
    if (getenv("RB_HASH_CW") && b->wpr_uniform && nw % b->wpr_uniform == 0) {   // experimental: several words per thread
        wpr = (int)b->wpr_uniform;
        cw = std::max(1, std::min(wpr, atoi(getenv("RB_HASH_CW"))));
        cpr = (wpr + cw - 1) / cw;
        nthreads = (nw / wpr) * cpr;
        if (cw == 1) { wpr = 0; cpr = 0; }
    }
    dim3 g(blocks_for(nthreads, HASH_TPB)), t(HASH_TPB);
#define RB_LAUNCH_HASH(M)                                                                        \
    hipLaunchKernelGGL(k_hash_windows<M>, g, t, 0, s, b->codes, b->valid, b->word_read, b->woff, \
                       b->len, w0, nw, k, chunk_off, first_read, pos_bits, keys, vals, out_read, out_pos, nthreads, cw, wpr, cpr)
    if (mode == 0) RB_LAUNCH_HASH(0);
    else if (mode == 2) RB_LAUNCH_HASH(2);
    else RB_LAUNCH_HASH(1);
#undef RB_LAUNCH_HASH
}

// words per read if the read-per-lane kernels apply (uniform batch of short reads), else 0: one word per lane
// (RB_READ_LANES=0 forces the one-word kernels)
static uint32_t read_lane_words(const rb_batch *b, int64_t nw, int k, int kmax = 31) {
    const bool off = getenv("RB_READ_LANES") && atoi(getenv("RB_READ_LANES")) == 0;
    if (!off && k <= kmax && b->wpr_uniform && b->wpr_uniform <= (uint32_t)RB_READ_WORDS && nw % b->wpr_uniform == 0) return b->wpr_uniform;
    return 0u;
}
// ... and for batches whose reads differ in length (trimmed reads: 4 or 5 words each) but all fit RB_READ_WORDS words: a lane still
// takes a whole read, found through the batch's word offsets (RB_RAGGED_LANES=0: the one-word kernels, as before round 3)
static bool read_lanes_ragged(const rb_batch *b, int k, int kmax = 31) {
    const bool off = (getenv("RB_READ_LANES") && atoi(getenv("RB_READ_LANES")) == 0) || (getenv("RB_RAGGED_LANES") && atoi(getenv("RB_RAGGED_LANES")) == 0);
    return !off && k <= kmax && !b->wpr_uniform && b->max_len <= 32u * (uint32_t)RB_READ_WORDS && !b->h_woff.empty();
}
// may a call with 32 <= k <= 63 use the minimizer-bucketed cache?  Only the read-per-lane kernel with the fetch ahead looks it up there.
bool filter_wide_mpf_ok(const rb_batch *b, int64_t nw, int k) {
    if (k <= RB_MPF_MAX_K || k > RB_MPF_WIDE_MAX_K) return false;
    if (getenv("RB_FILTER_PIPE") && atoi(getenv("RB_FILTER_PIPE")) != 1 && atoi(getenv("RB_FILTER_PIPE")) != 3) return false;
    if (getenv("RB_WIDE_MPF") && atoi(getenv("RB_WIDE_MPF")) == 0) return false;
    return read_lane_words(b, nw, k, RB_MPF_WIDE_MAX_K) != 0u || read_lanes_ragged(b, k, RB_MPF_WIDE_MAX_K);
}
void launch_filter_windows(const rb_batch *b, int64_t w0, int64_t nw, int k, int mode, uint32_t first_read,
                           uint32_t pos_bits, uint64_t seed, uint64_t ordinal0, Npf cache, uint32_t *cnt, uint32_t *keepmask,
                           uint32_t *total_spread, hipStream_t s, OwnRange own, Mpf mcache, void *wstate) {
    if (nw <= 0) return;
    uint32_t dbgf = getenv("RB_FILT_DBG") ? (uint32_t)atoi(getenv("RB_FILT_DBG")) : 0u;
    if (!cache.tab) dbgf |= 1u;                       // no cache: ownership test only
    dim3 g(blocks_for(nw, 64)), t(64);
#define RB_LAUNCH_FILT(M, P, W)                                                                                 \
    hipLaunchKernelGGL((k_filter_windows_fast<M, P, W>), g, t, ring_bytes, s, b->codes, b->valid, b->word_read, b->woff, b->len, \
                       w0, nw, k, first_read, pos_bits, seed, ordinal0, cache, mcache, cnt, keepmask, total_spread, dbgf, own)
    RB_REQUIRE(k <= 64, "prefilter kernels take k <= 64");
    const bool wide = mcache.tab && k > RB_MPF_MAX_K;          // 32 <= k <= 63 with the minimizer-bucketed cache: add_range asked filter_wide_mpf_ok
    if (wide) RB_REQUIRE(filter_wide_mpf_ok(b, nw, k) && own.hi == 0, "prefilter: the minimizer-bucketed cache at k = %d needs a read per lane", k);
    const bool use_m = mcache.tab && (wide || (k <= RB_MPF_MAX_K && mcache.m <= (uint32_t)k && (uint32_t)k - mcache.m + 1u <= RB_MPF_MAX_RING));
    const size_t ring_bytes = use_m ? ((size_t)mpf_kp((uint32_t)k) - mcache.m + 1u) * 64u * sizeof(uint32_t) : 0;   // LDS per wavefront: 12.7 KB -> 11.2 KB at k = 25
    const uint32_t C = read_lane_words(b, nw, k, wide ? RB_MPF_WIDE_MAX_K : 31);
    if (C || read_lanes_ragged(b, k, wide ? RB_MPF_WIDE_MAX_K : 31)) {
        uint32_t rd0 = 0, n_rd = 0;
        if (!C) {                                              // the reads of words [w0, w0 + nw): both ends are read boundaries
            const auto &wo = b->h_woff;
            const int64_t r0 = std::lower_bound(wo.begin(), wo.end(), (uint32_t)w0) - wo.begin();
            const int64_t r1 = std::lower_bound(wo.begin(), wo.end(), (uint32_t)(w0 + nw)) - wo.begin();
            RB_REQUIRE(r0 < (int64_t)wo.size() && wo[(size_t)r0] == (uint32_t)w0 && r1 < (int64_t)wo.size() && wo[(size_t)r1] == (uint32_t)(w0 + nw),
                       "prefilter: word range does not start and end at read boundaries");
            rd0 = (uint32_t)r0; n_rd = (uint32_t)(r1 - r0);
        }
        dim3 gc(blocks_for(C ? nw / C : (int64_t)n_rd, 64));
#define RB_LAUNCH_FC(M, P)                                                                                        \
    hipLaunchKernelGGL((k_filter_reads<M, P>), gc, t, ring_bytes, s, b->codes, b->valid, b->word_read, b->len, w0, nw, k, C, \
                       first_read, pos_bits, seed, ordinal0, cache, mcache, cnt, keepmask, total_spread, dbgf, own, \
                       reinterpret_cast<ulonglong2 *>(wstate), b->woff, rd0, n_rd)
        // 0: k_filter_reads (fetch in place), 1: each lane fetches its bucket one step ahead (default of rounds 3-4), 3 (default): the wavefront
        // fetches cooperatively, one step ahead (its askers' list packs a bucket number into 26 bits).  (2, the fetch two steps ahead, was built
        // twice and slower both times — HISTORY.md "Round 5" — and is gone.)  The walkers of 0 and 1 are also what RB_FILTER_CHECK compares with.
        const int pipe_raw = getenv("RB_FILTER_PIPE") ? atoi(getenv("RB_FILTER_PIPE")) : 3;
        const int pipe_env = pipe_raw == 2 ? 1 : pipe_raw;
        const int pipe_fit = (pipe_env == 3 && mcache.log2b > 26u) ? 1 : pipe_env;
        const int pipe = wide ? (pipe_fit == 3 ? 3 : 1) : pipe_fit;
        if (use_m && pipe && own.lo == 0 && own.hi == 0) {         // the whole index range is this handle's: the bucket fetch one / two steps ahead of its use
#define RB_LAUNCH_FN(M)                                                                                              \
    hipLaunchKernelGGL((k_filter_reads_pipe<M, false>), gc, t, ring_bytes, s, b->codes, b->valid, b->word_read, b->len, w0, nw, k, C, \
                       first_read, pos_bits, seed, ordinal0, mcache, cnt, keepmask, total_spread, dbgf, reinterpret_cast<ulonglong2 *>(wstate), b->woff, rd0, n_rd)
#define RB_LAUNCH_FW(M)                                                                                              \
    hipLaunchKernelGGL((k_filter_reads_pipe<M, true>), gc, t, ring_bytes, s, b->codes, b->valid, b->word_read, b->len, w0, nw, k, C, \
                       first_read, pos_bits, seed, ordinal0, mcache, cnt, keepmask, total_spread, dbgf, reinterpret_cast<ulonglong2 *>(wstate), b->woff, rd0, n_rd)
#define RB_LAUNCH_FCO(M, WD)                                                                                         \
    hipLaunchKernelGGL((k_filter_reads_coop<M, WD>), gc, t, ring_bytes, s, b->codes, b->valid, b->word_read, b->len, w0, nw, k, C, \
                       first_read, pos_bits, seed, ordinal0, mcache, cnt, keepmask, total_spread, dbgf, reinterpret_cast<ulonglong2 *>(wstate), b->woff, rd0, n_rd)
            if (pipe == 3) {
                const bool check = getenv("RB_FILTER_CHECK") && atoi(getenv("RB_FILTER_CHECK"));
                if (check && wstate) RB_HIP(hipMemsetAsync(wstate, 0xEE, (size_t)nw * 16, s));      // (words no state is stored for compare equal)
                if (wide) { if (mode == 0) RB_LAUNCH_FCO(0, true); else if (mode == 2) RB_LAUNCH_FCO(2, true); else RB_LAUNCH_FCO(1, true); }
                else { if (mode == 0) RB_LAUNCH_FCO(0, false); else if (mode == 2) RB_LAUNCH_FCO(2, false); else RB_LAUNCH_FCO(1, false); }
                if (check) {
                    // Verification (tests, and whoever touches these kernels): the cooperative walker only ever errs on the safe side if it errs — a lookup
                    // in an image that is not the window's bucket misses, the window is kept, every filter byte still matches the oracle — so parity tests
                    // cannot see a broken fetch.  Here the one-step walker (and, for k <= 31, the unpipelined one) decides the same words into scratch
                    // arrays and every count, mask and resume state is compared; a difference fails the call.
                    struct Scratch { uint32_t *c = nullptr, *m = nullptr, *t = nullptr; ulonglong2 *ws = nullptr;
                                     ~Scratch() { (void)hipFree(c); (void)hipFree(m); (void)hipFree(t); (void)hipFree(ws); } } A, B;
                    const size_t nb = (size_t)nw * 4;
                    for (Scratch *x : {&A, &B}) {
                        RB_HIP(hipMalloc(&x->c, nb)); RB_HIP(hipMalloc(&x->m, nb)); RB_HIP(hipMalloc(&x->t, 4096)); RB_HIP(hipMemsetAsync(x->t, 0, 4096, s));
                        RB_HIP(hipMemsetAsync(x->c, 0xEE, nb, s)); RB_HIP(hipMemsetAsync(x->m, 0xEE, nb, s));
                        if (wstate) { RB_HIP(hipMalloc(&x->ws, nb * 4)); RB_HIP(hipMemsetAsync(x->ws, 0xEE, nb * 4, s)); }
                    }
                    ulonglong2 *const ws_coop = reinterpret_cast<ulonglong2 *>(wstate);
                    {   // the one-step walker
                        uint32_t *cnt = A.c, *keepmask = A.m, *total_spread = A.t; void *wstate = A.ws;
                        if (wide) { if (mode == 0) RB_LAUNCH_FW(0); else if (mode == 2) RB_LAUNCH_FW(2); else RB_LAUNCH_FW(1); }
                        else { if (mode == 0) RB_LAUNCH_FN(0); else if (mode == 2) RB_LAUNCH_FN(2); else RB_LAUNCH_FN(1); }
                    }
                    if (!wide) {   // the walker that fetches in place
                        uint32_t *cnt = B.c, *keepmask = B.m, *total_spread = B.t; void *wstate = B.ws;
                        if (mode == 0) RB_LAUNCH_FC(0, true); else if (mode == 2) RB_LAUNCH_FC(2, true); else RB_LAUNCH_FC(1, true);
                    }
                    std::vector<uint32_t> hc(nw), hm(nw), ac(nw), am(nw), bc(nw), bm(nw);
                    std::vector<unsigned long long> hw, aw;
                    RB_HIP(hipMemcpyAsync(hc.data(), cnt, nb, hipMemcpyDeviceToHost, s)); RB_HIP(hipMemcpyAsync(hm.data(), keepmask, nb, hipMemcpyDeviceToHost, s));
                    RB_HIP(hipMemcpyAsync(ac.data(), A.c, nb, hipMemcpyDeviceToHost, s)); RB_HIP(hipMemcpyAsync(am.data(), A.m, nb, hipMemcpyDeviceToHost, s));
                    RB_HIP(hipMemcpyAsync(bc.data(), B.c, nb, hipMemcpyDeviceToHost, s)); RB_HIP(hipMemcpyAsync(bm.data(), B.m, nb, hipMemcpyDeviceToHost, s));
                    if (wstate) {
                        hw.resize((size_t)nw * 2); aw.resize((size_t)nw * 2);
                        RB_HIP(hipMemcpyAsync(hw.data(), ws_coop, nb * 4, hipMemcpyDeviceToHost, s)); RB_HIP(hipMemcpyAsync(aw.data(), A.ws, nb * 4, hipMemcpyDeviceToHost, s));
                    }
                    RB_HIP(hipStreamSynchronize(s));
                    int64_t bad = 0, first = -1;
                    for (int64_t i = 0; i < nw; ++i) {
                        const bool d = hc[i] != ac[i] || hm[i] != am[i] || (!wide && (hc[i] != bc[i] || hm[i] != bm[i])) ||
                                       (wstate && (hw[2 * i] != aw[2 * i] || hw[2 * i + 1] != aw[2 * i + 1]));
                        if (d) { if (first < 0) first = i; ++bad; }
                    }
                    if (atoi(getenv("RB_FILTER_CHECK")) > 1) fprintf(stderr, "[rb] filter check: %lld words, %lld differ\n", (long long)nw, (long long)bad);
                    RB_REQUIRE(bad == 0, "prefilter check: %lld of %lld words differ between the walkers (first: word %lld, count %u / %u / %u, mask %08x / %08x / %08x)",
                               (long long)bad, (long long)nw, (long long)first, hc[(size_t)std::max<int64_t>(first, 0)], ac[(size_t)std::max<int64_t>(first, 0)],
                               bc[(size_t)std::max<int64_t>(first, 0)], hm[(size_t)std::max<int64_t>(first, 0)], am[(size_t)std::max<int64_t>(first, 0)], bm[(size_t)std::max<int64_t>(first, 0)]);
                }
            }
            else if (wide) { if (mode == 0) RB_LAUNCH_FW(0); else if (mode == 2) RB_LAUNCH_FW(2); else RB_LAUNCH_FW(1); }
            else { if (mode == 0) RB_LAUNCH_FN(0); else if (mode == 2) RB_LAUNCH_FN(2); else RB_LAUNCH_FN(1); }
#undef RB_LAUNCH_FCO
#undef RB_LAUNCH_FN
#undef RB_LAUNCH_FW
        } else if (use_m) { if (mode == 0) RB_LAUNCH_FC(0, true); else if (mode == 2) RB_LAUNCH_FC(2, true); else RB_LAUNCH_FC(1, true); }
        else { if (mode == 0) RB_LAUNCH_FC(0, false); else if (mode == 2) RB_LAUNCH_FC(2, false); else RB_LAUNCH_FC(1, false); }
#undef RB_LAUNCH_FC
    } else if (use_m) {
        if (mode == 0) RB_LAUNCH_FILT(0, true, false); else if (mode == 2) RB_LAUNCH_FILT(2, true, false); else RB_LAUNCH_FILT(1, true, false);
    } else if (k <= 31) {
        if (mode == 0) RB_LAUNCH_FILT(0, false, false); else if (mode == 2) RB_LAUNCH_FILT(2, false, false); else RB_LAUNCH_FILT(1, false, false);
    } else {   // 32 <= k <= 64: hash-bucketed cache, three words per lane
        if (mode == 0) RB_LAUNCH_FILT(0, false, true); else if (mode == 2) RB_LAUNCH_FILT(2, false, true); else RB_LAUNCH_FILT(1, false, true);
    }
#undef RB_LAUNCH_FILT
}
// ... and at 32 <= k <= 63, when the call goes through the read-per-lane kernel with the minimizer-bucketed cache (add_range knows)
bool filter_saves_state_wide(const rb_batch *b, int64_t nw, int k) { return filter_wide_mpf_ok(b, nw, k) && !(getenv("RB_EMIT_RESUME") && atoi(getenv("RB_EMIT_RESUME")) == 0); }
bool filter_saves_state(const rb_batch *b, int64_t nw, int k) { return (read_lane_words(b, nw, k) != 0u || read_lanes_ragged(b, k)) && !(getenv("RB_EMIT_RESUME") && atoi(getenv("RB_EMIT_RESUME")) == 0); }
void launch_hash_windows_masked(const rb_batch *b, int64_t w0, int64_t nw, int k, int mode, const uint32_t *chunk_off,
                                const uint32_t *keepmask, uint32_t first_read, uint32_t pos_bits, uint64_t *keys, uint32_t *vals,
                                hipStream_t s, const void *wstate) {
    if (nw <= 0) return;
    const bool sparse = !(getenv("RB_SPARSE_EMIT") && atoi(getenv("RB_SPARSE_EMIT")) == 0);
    if (keepmask && wstate) {
        dim3 gs(blocks_for(nw, RB_SPARSE_WORDS)), ts(64);
#define RB_LAUNCH_RS(M)                                                                                    \
    hipLaunchKernelGGL(k_hash_windows_resume<M>, gs, ts, 0, s, b->codes, b->valid, b->word_read, b->woff, \
                       b->len, w0, nw, k, chunk_off, first_read, pos_bits, keys, vals, keepmask, reinterpret_cast<const ulonglong2 *>(wstate))
        if (mode == 0) RB_LAUNCH_RS(0); else if (mode == 2) RB_LAUNCH_RS(2); else RB_LAUNCH_RS(1);
#undef RB_LAUNCH_RS
        return;
    }
    if (keepmask && sparse) {
        dim3 gs(blocks_for(nw, RB_SPARSE_WORDS)), ts(64);
#define RB_LAUNCH_SP(M, W)                                                                                 \
    hipLaunchKernelGGL((k_hash_windows_sparse<M, W>), gs, ts, 0, s, b->codes, b->valid, b->word_read, b->woff, \
                       b->len, w0, nw, k, chunk_off, first_read, pos_bits, keys, vals, keepmask)
        if (k <= 31) { if (mode == 0) RB_LAUNCH_SP(0, false); else if (mode == 2) RB_LAUNCH_SP(2, false); else RB_LAUNCH_SP(1, false); }
        else { if (mode == 0) RB_LAUNCH_SP(0, true); else if (mode == 2) RB_LAUNCH_SP(2, true); else RB_LAUNCH_SP(1, true); }
#undef RB_LAUNCH_SP
        return;
    }
    dim3 g(blocks_for(nw, 64)), t(64);
#define RB_LAUNCH_FASTM(M, W)                                                                         \
    hipLaunchKernelGGL((k_hash_windows_fast<M, W>), g, t, 0, s, b->codes, b->valid, b->word_read, b->woff, \
                       b->len, w0, nw, k, chunk_off, first_read, pos_bits, keys, vals, keepmask)
    if (k <= 31) { if (mode == 0) RB_LAUNCH_FASTM(0, false); else if (mode == 2) RB_LAUNCH_FASTM(2, false); else RB_LAUNCH_FASTM(1, false); }
    else { if (mode == 0) RB_LAUNCH_FASTM(0, true); else if (mode == 2) RB_LAUNCH_FASTM(2, true); else RB_LAUNCH_FASTM(1, true); }
#undef RB_LAUNCH_FASTM
}

}  // namespace rb

// ------------------------------------------------------------------- C ABI ----
extern "C" {

int rb_batch_destroy(rb_batch *b) {
    if (!b) return RB_OK;
    (void)hipSetDevice(b->device);
    if (b->pool) {                      // (a chunk of a chunked ingest: the next chunk takes the blocks over — no hipFree, which waits for the device)
        for (void *p : {(void *)b->codes, (void *)b->valid, (void *)b->rnz, (void *)b->word_read, (void *)b->woff, (void *)b->len}) b->pool->put(p);
        delete b;
        return RB_OK;
    }
    if (b->codes) (void)hipFree(b->codes);
    if (b->valid) (void)hipFree(b->valid);
    if (b->rnz) (void)hipFree(b->rnz);
    if (b->word_read) (void)hipFree(b->word_read);
    if (b->woff) (void)hipFree(b->woff);
    if (b->len) (void)hipFree(b->len);
    delete b;
    return RB_OK;
}

int rb_batch_info(const rb_batch *b, int64_t *n_reads, int64_t *n_bases, int64_t *device_bytes) {
    if (!b) { set_error("rb_batch_info: null batch"); return RB_ERR_INVALID; }
    if (n_reads) *n_reads = b->n_reads;
    if (n_bases) *n_bases = b->n_bases;
    if (device_bytes) *device_bytes = (int64_t)b->device_bytes;
    return RB_OK;
}

}  // extern "C"

namespace rb {
// ASCII reads [first, first+n) -> packed batch, in two halves so that a caller can overlap the upload and the
// GPU-side 2-bit encode of one chunk with whatever the GPU does for the previous one: begin() allocates,
// enqueues the copies + encode kernel on `st` and returns; finish() waits for them and frees the staging.
void ascii_batch_begin(AsciiUpload &u, int device, const char *seq, const char *qual, const int64_t *offsets, int64_t first, int64_t n_reads,
                       int min_base_qual, hipStream_t st, bool want_rnz) {
    RB_REQUIRE(offsets && n_reads >= 0 && (seq || n_reads == 0 || offsets[first + n_reads] == offsets[first]), "rb_batch_create_ascii: null argument");
    RB_REQUIRE(min_base_qual >= 0 && min_base_qual < 94, "rb_batch_create_ascii: min_base_qual out of range");
    RB_HIP(hipSetDevice(device));
    offsets += first;
    rb_batch *b = new rb_batch();
    u.b = b; u.st = st;
    b->pool = u.pool;
    b->device = device;
    b->n_reads = n_reads;
    // host-side tables of the chunk in ONE pass over the offsets: word offset and length of every read, base offsets relative to the chunk
    // (rounds 1-5: three passes into fresh std::vectors, 16 ms per 1.7 M-read chunk against 8 ms of GPU work for it — the preparing thread was
    // what rb_graph_add_reads waited for)
    std::vector<uint32_t> woff_v;
    uint32_t *woff = nullptr, *len = nullptr;
    int64_t *rel = nullptr;
    if (u.hs) { u.hs->reserve((size_t)n_reads + 1); woff = u.hs->woff; len = u.hs->len; rel = u.hs->rel; }
    else {
        woff_v.resize((size_t)n_reads + 1); u.len.resize((size_t)std::max<int64_t>(n_reads, 1)); u.rel.resize((size_t)n_reads + 1);
        woff = woff_v.data(); len = u.len.data(); rel = u.rel.data();
    }
    const int64_t base0 = n_reads ? offsets[0] : 0;
    uint64_t words = 0;
    uint32_t max_len = 0, wpr = n_reads ? (uint32_t)((offsets[1] - offsets[0] + 31) / 32) : 0;
    bool bad = false;
    for (int64_t i = 0; i < n_reads; ++i) {
        const int64_t l = offsets[i + 1] - offsets[i];
        bad |= l < 0 || l >= (int64_t)1 << 30;
        const uint32_t w = (uint32_t)((l + 31) >> 5);
        woff[i] = (uint32_t)words; len[i] = (uint32_t)l; rel[i] = offsets[i] - base0;
        words += w;
        max_len = std::max(max_len, (uint32_t)l);
        if (w != wpr) wpr = 0;
    }
    if (n_reads == 0) len[0] = 0;
    RB_REQUIRE(!bad, "rb_batch_create_ascii: a read of the %lld from read %lld on has an invalid length", (long long)n_reads, (long long)first);
    RB_REQUIRE(words < 0xFFFFFFF0ull, "rb_batch_create_ascii: batch too large (> 2^32 words)");
    woff[n_reads] = (uint32_t)words; rel[n_reads] = offsets[n_reads] - base0;
    b->n_words = (int64_t)words;
    b->max_len = max_len;
    b->wpr_uniform = wpr;
    b->n_bases = n_reads ? offsets[n_reads] - base0 : 0;
    alloc_batch_arrays(b);
    if (u.hs) b->h_woff.borrow(woff, (size_t)n_reads + 1); else b->h_woff = woff_v;
    RB_HIP(hipMemcpyAsync(b->woff, b->h_woff.data(), ((size_t)n_reads + 1) * 4, hipMemcpyHostToDevice, st));
    if (n_reads) RB_HIP(hipMemcpyAsync(b->len, len, (size_t)n_reads * 4, hipMemcpyHostToDevice, st));
    if (words) {
        const size_t nb = (size_t)b->n_bases;
        u.d_seq = u.dev_alloc<uint8_t>(std::max<size_t>(nb, 1));
        RB_HIP(hipMemcpyAsync(u.d_seq, seq + base0, nb, hipMemcpyHostToDevice, st));
        if (qual) {
            u.d_qual = u.dev_alloc<uint8_t>(std::max<size_t>(nb, 1));
            RB_HIP(hipMemcpyAsync(u.d_qual, qual + base0, nb, hipMemcpyHostToDevice, st));
        }
        u.d_off = u.dev_alloc<int64_t>(((size_t)n_reads + 1) * 8);
        RB_HIP(hipMemcpyAsync(u.d_off, rel, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, st));
        if (want_rnz) { b->rnz = u.dev_alloc<uint32_t>((size_t)words * 4); b->device_bytes += (size_t)words * 4; }
        hipLaunchKernelGGL(k_encode_ascii_t<false>, dim3(blocks_for((int64_t)words)), dim3(TPB), 0, st, u.d_seq, u.d_qual, u.d_off, b->woff, n_reads,
                           (int64_t)words, min_base_qual, b->codes, b->valid, b->word_read, b->rnz, (int64_t)0, (int64_t)0);
        RB_HIP(hipGetLastError());
    }
}
rb_batch *ascii_batch_finish(AsciiUpload &u) {
    hipError_t e = hipStreamSynchronize(u.st);
    u.drop();
    rb_batch *b = u.b;
    u.b = nullptr;
    if (e != hipSuccess) { if (b) rb_batch_destroy(b); RB_HIP(e); }
    return b;
}
void ascii_batch_abort(AsciiUpload &u) {
    if (u.st) (void)hipStreamSynchronize(u.st);
    u.drop();
    if (u.b) { rb_batch_destroy(u.b); u.b = nullptr; }
}
}  // namespace rb

extern "C" {

int rb_batch_create_ascii(int device, const char *seq, const char *qual, const int64_t *offsets,
                          int64_t n_reads, int min_base_qual, rb_batch **out) {
    rb::AsciiUpload u;
    try {
        RB_REQUIRE(out, "rb_batch_create_ascii: null argument");
        rb::ascii_batch_begin(u, device, seq, qual, offsets, 0, n_reads, min_base_qual, nullptr);
        *out = rb::ascii_batch_finish(u);
        return RB_OK;
    } catch (const HipError &e) { rb::ascii_batch_abort(u); return e.code; }
    catch (const std::bad_alloc &) { rb::ascii_batch_abort(u); set_error("host allocation failed"); return RB_ERR_NOMEM; }
}

int rb_batch_download_ascii(const rb_batch *b, int64_t first, int64_t n, char *seq, int64_t *offsets) {
    try {
        RB_REQUIRE(b && offsets && first >= 0 && n >= 0 && first + n <= b->n_reads, "rb_batch_download_ascii: bad range");
        RB_HIP(hipSetDevice(b->device));
        std::vector<uint32_t> len((size_t)std::max<int64_t>(n, 1)), woff(2);
        if (n) RB_HIP(hipMemcpy(len.data(), b->len + first, (size_t)n * 4, hipMemcpyDeviceToHost));
        offsets[0] = 0;
        for (int64_t i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + len[(size_t)i];
        if (!n || !offsets[n] || !seq) return RB_OK;
        RB_HIP(hipMemcpy(&woff[0], b->woff + first, 4, hipMemcpyDeviceToHost));
        RB_HIP(hipMemcpy(&woff[1], b->woff + first + n, 4, hipMemcpyDeviceToHost));
        int64_t *d_off = nullptr; uint8_t *d_out = nullptr;
        RB_HIP(hipMalloc(&d_off, ((size_t)n + 1) * 8));
        RB_HIP(hipMalloc(&d_out, (size_t)offsets[n]));
        RB_HIP(hipMemcpy(d_off, offsets, ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
        int64_t nw = (int64_t)woff[1] - woff[0];
        hipLaunchKernelGGL(k_decode_ascii, dim3(blocks_for(nw)), dim3(TPB), 0, 0, b->codes, b->valid,
                           b->word_read, b->woff, b->len, (int64_t)woff[0], nw, d_off, (uint32_t)first, d_out);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(seq, d_out, (size_t)offsets[n], hipMemcpyDeviceToHost);
        (void)hipFree(d_off); (void)hipFree(d_out);
        RB_HIP(e);
        return RB_OK;
    } catch (const HipError &e) { return e.code; }
}

int rb_batch_create_synthetic(int device, const rb_synth_params *p, rb_batch **out) {
    try {
        RB_REQUIRE(p && out && p->n_pairs > 0 && p->genome_bases >= 1024 && p->read_len > 0,
                   "rb_batch_create_synthetic: bad parameters");
        RB_REQUIRE(p->pair_offset >= 0 && (p->total_pairs == 0 || p->pair_offset + p->n_pairs <= p->total_pairs),
                   "rb_batch_create_synthetic: slice [%lld,+%lld) outside the set of %lld pairs", (long long)p->pair_offset,
                   (long long)p->n_pairs, (long long)p->total_pairs);
        RB_REQUIRE(p->tx_min >= p->read_len && p->tx_max >= p->tx_min, "rb_batch_create_synthetic: transcript range must cover read_len");
        RB_HIP(hipSetDevice(device));
        // transcript table + expression CDF on the host (small)
        uint64_t st = p->seed * 0x9E3779B97F4A7C15ull + 12345;
        auto next = [&st]() { st += 0x9E3779B97F4A7C15ull; uint64_t z = st;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
        auto unif = [&next]() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); };
        std::vector<int64_t> tstart; std::vector<int32_t> tlen;
        int64_t pos = 0;
        while (pos < p->genome_bases) {
            int64_t l = p->tx_min + (int64_t)(unif() * (double)(p->tx_max - p->tx_min + 1));
            if (pos + l > p->genome_bases) l = p->genome_bases - pos;
            if (l < p->tx_min) { if (!tlen.empty()) tlen.back() += (int32_t)l; else { tstart.push_back(pos); tlen.push_back((int32_t)l); } break; }
            tstart.push_back(pos); tlen.push_back((int32_t)l); pos += l;
        }
        RB_REQUIRE(!tlen.empty() && tlen[0] >= p->read_len, "rb_batch_create_synthetic: genome too small");
        std::vector<double> wgt(tlen.size());
        double tot = 0;
        for (size_t i = 0; i < tlen.size(); ++i) {
            double e = 1.0;
            if (p->expr_sigma > 0) {
                double u1 = std::max(unif(), 1e-300), u2 = unif();
                e = exp((double)p->expr_sigma * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
            }
            wgt[i] = e * tlen[i]; tot += wgt[i];
        }
        std::vector<float> cdf(tlen.size());
        double acc = 0;
        for (size_t i = 0; i < tlen.size(); ++i) { acc += wgt[i] / tot; cdf[i] = (float)acc; }
        cdf.back() = 2.0f;

        rb_batch *b = new rb_batch();
        HostGuard guard{b};
        b->device = device;
        b->n_reads = 2 * p->n_pairs;
        int wpr = (p->read_len + 31) / 32;
        RB_REQUIRE((uint64_t)b->n_reads * (uint64_t)wpr < 0xFFFFFFF0ull, "rb_batch_create_synthetic: batch too large");
        b->n_words = b->n_reads * wpr;
        b->n_bases = b->n_reads * p->read_len;
        b->max_len = (uint32_t)p->read_len;
        b->wpr_uniform = (uint32_t)((p->read_len + 31) / 32);
        alloc_batch_arrays(b);
        {   // woff / len are arithmetic progressions: fill from host in slabs
            const size_t slab = 1 << 22;
            std::vector<uint32_t> tmp(slab);
            b->h_woff.resize((size_t)b->n_reads + 1);
            for (size_t i = 0; i <= (size_t)b->n_reads; ++i) b->h_woff[i] = (uint32_t)(i * (size_t)wpr);
            RB_HIP(hipMemcpy(b->woff, b->h_woff.data(), ((size_t)b->n_reads + 1) * 4, hipMemcpyHostToDevice));
            std::fill(tmp.begin(), tmp.end(), (uint32_t)p->read_len);
            for (size_t base = 0; base < (size_t)b->n_reads; base += slab) {
                size_t m = std::min(slab, (size_t)b->n_reads - base);
                RB_HIP(hipMemcpy(b->len + base, tmp.data(), m * 4, hipMemcpyHostToDevice));
            }
        }
        uint64_t *d_genome = nullptr; int64_t *d_ts = nullptr; int32_t *d_tl = nullptr; float *d_cdf = nullptr;
        int64_t gw = (p->genome_bases + 31) / 32 + 1;
        hipError_t err = hipSuccess;
        auto chk = [&err](hipError_t e) { if (err == hipSuccess) err = e; };
        chk(hipMalloc(&d_genome, (size_t)gw * 8));
        chk(hipMalloc(&d_ts, tstart.size() * 8));
        chk(hipMalloc(&d_tl, tlen.size() * 4));
        chk(hipMalloc(&d_cdf, cdf.size() * 4));
        if (err == hipSuccess) {
            chk(hipMemcpy(d_ts, tstart.data(), tstart.size() * 8, hipMemcpyHostToDevice));
            chk(hipMemcpy(d_tl, tlen.data(), tlen.size() * 4, hipMemcpyHostToDevice));
            chk(hipMemcpy(d_cdf, cdf.data(), cdf.size() * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_synth_genome, dim3(blocks_for(gw)), dim3(TPB), 0, 0, d_genome, gw, p->seed);
            hipLaunchKernelGGL(k_synth_reads, dim3(blocks_for(b->n_words)), dim3(TPB), 0, 0, d_genome, d_ts, d_tl,
                               d_cdf, (int)tlen.size(), p->n_pairs, (int)p->read_len, wpr, (float)p->frag_mean,
                               (float)p->frag_sd, p->sub_rate, p->n_rate, p->seed, p->pair_offset,
                               p->total_pairs > 0 ? p->total_pairs : p->n_pairs,
                               // substituted bases carry quality '#' and are masked like any base below -q 3.  RB_SYNTH_KEEP_ERRORS=1
                               // (measurement knob): they pass the threshold, so every error adds up to k k-mers that are seen once
                               (getenv("RB_SYNTH_KEEP_ERRORS") && atoi(getenv("RB_SYNTH_KEEP_ERRORS")) != 0) ? 1 : 0,
                               b->codes, b->valid, b->word_read);
            chk(hipGetLastError());
            chk(hipDeviceSynchronize());
        }
        if (d_genome) (void)hipFree(d_genome);
        if (d_ts) (void)hipFree(d_ts);
        if (d_tl) (void)hipFree(d_tl);
        if (d_cdf) (void)hipFree(d_cdf);
        RB_HIP(err);
        guard.b = nullptr;
        *out = b;
        return RB_OK;
    } catch (const HipError &e) { return e.code; }
    catch (const std::bad_alloc &) { set_error("host allocation failed"); return RB_ERR_NOMEM; }
}

int rb_nthash_batch(const rb_batch *b, int k, int mode, int64_t first, int64_t n, int64_t *count,
                    uint64_t *out_h0, uint32_t *out_read, uint32_t *out_pos) {
    try {
        RB_REQUIRE(b && count && k >= 1 && k <= RB_MAX_K && mode >= 0 && mode <= 2, "rb_nthash_batch: bad argument");
        RB_REQUIRE(first >= 0 && n >= 0 && first + n <= b->n_reads, "rb_nthash_batch: bad read range");
        RB_HIP(hipSetDevice(b->device));
        *count = 0;
        if (n == 0) return RB_OK;
        uint32_t wo[2];
        RB_HIP(hipMemcpy(&wo[0], b->woff + first, 4, hipMemcpyDeviceToHost));
        RB_HIP(hipMemcpy(&wo[1], b->woff + first + n, 4, hipMemcpyDeviceToHost));
        int64_t w0 = wo[0], nw = (int64_t)wo[1] - wo[0];
        if (nw == 0) return RB_OK;
        DevBuf cnt, off, tmp, keys, rd, ps;
        struct Rel { DevBuf *b[6]; ~Rel() { for (auto x : b) x->release(); } } rel{{&cnt, &off, &tmp, &keys, &rd, &ps}};
        cnt.reserve(((size_t)nw + 1) * 4); off.reserve(((size_t)nw + 1) * 4);
        RB_HIP(hipMemsetAsync(cnt.p, 0, ((size_t)nw + 1) * 4, 0));
        launch_count_windows(b, w0, nw, k, cnt.as<uint32_t>(), 0);
        size_t tb = scan_temp_bytes((size_t)nw + 1);
        tmp.reserve(tb);
        exclusive_scan_u32(tmp.p, tb, cnt.as<uint32_t>(), off.as<uint32_t>(), (size_t)nw + 1, 0);
        uint32_t total = 0;
        RB_HIP(hipMemcpy(&total, off.as<uint32_t>() + nw, 4, hipMemcpyDeviceToHost));
        *count = total;
        if (!out_h0 || total == 0) return RB_OK;
        keys.reserve((size_t)total * 8);
        if (out_read) { rd.reserve((size_t)total * 4); ps.reserve((size_t)total * 4); }
        launch_hash_windows(b, w0, nw, k, mode, off.as<uint32_t>(), 0, 0, keys.as<uint64_t>(), nullptr,
                            out_read ? rd.as<uint32_t>() : nullptr, out_read ? ps.as<uint32_t>() : nullptr, 0);
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpy(out_h0, keys.p, (size_t)total * 8, hipMemcpyDeviceToHost));
        if (out_read) {
            RB_HIP(hipMemcpy(out_read, rd.p, (size_t)total * 4, hipMemcpyDeviceToHost));
            if (out_pos) RB_HIP(hipMemcpy(out_pos, ps.p, (size_t)total * 4, hipMemcpyDeviceToHost));
        }
        return RB_OK;
    } catch (const HipError &e) { return e.code; }
}

}  // extern "C"

// ------------------------------------------------------------ .nbits ingest ----
// R/util/SeqBitsUtils.java:159-161, 236-263 / R/io/NucleotideBits{Reader,Writer}.java: a record is a 4-byte big-endian
// length followed by ceil(len/4) bytes, each holding 4 bases MSB first (A0 C1 G2 T3), stored minus 128.  The packed
// batch wants 32 bases per 64-bit word LSB first: one thread per output word reads 8 input bytes, undoes the offset
// (xor 0x80) and reverses the order of the four 2-bit groups of every byte — a pure bit permutation, no table.
namespace {
__device__ __forceinline__ uint32_t nbits_byte_to_lsb_first(uint32_t b) {
    b ^= 0x80u;                                                      // value + 128 (mod 256)
    return ((b >> 6) & 3u) | (((b >> 4) & 3u) << 2) | (((b >> 2) & 3u) << 4) | ((b & 3u) << 6);
}
__global__ void k_decode_nbits(const uint8_t *__restrict__ raw, const int64_t *__restrict__ rec_off /* payload offset per read */,
                               const uint32_t *__restrict__ woff, const uint32_t *__restrict__ len, int64_t n_reads, int64_t n_words,
                               uint64_t *__restrict__ codes, uint32_t *__restrict__ valid, uint32_t *__restrict__ word_read) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    int64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (woff[mid] <= (uint32_t)w) lo = mid; else hi = mid; }
    const int64_t r = lo;
    const uint32_t L = len[r], b0 = (uint32_t)(w - woff[r]) * 32u;
    const uint32_t nb = L - b0 < 32u ? L - b0 : 32u;                 // bases in this word
    const uint8_t *src = raw + rec_off[r] + (b0 >> 2);
    const uint32_t nbytes = (nb + 3u) >> 2;
    uint64_t c = 0;
    for (uint32_t i = 0; i < nbytes; ++i) c |= (uint64_t)nbits_byte_to_lsb_first(src[i]) << (8u * i);
    const uint32_t v = nb == 32u ? 0xFFFFFFFFu : (1u << nb) - 1u;
    if (nb < 32u) c &= (1ull << (2u * nb)) - 1ull;                   // the pad bases of the last byte are not part of the read
    codes[w] = c; valid[w] = v; word_read[w] = (uint32_t)r;
}
}  // namespace

extern "C" {

int rb_batch_create_nbits(int device, const void *bytes, size_t nbytes, int64_t max_reads, rb_batch **out, size_t *consumed) {
    uint8_t *d_raw = nullptr; int64_t *d_off = nullptr; rb_batch *b = nullptr;
    try {
        RB_REQUIRE(out && (bytes || nbytes == 0), "rb_batch_create_nbits: null argument");
        RB_HIP(hipSetDevice(device));
        const uint8_t *p = static_cast<const uint8_t *>(bytes);
        std::vector<int64_t> rec_off;
        std::vector<uint32_t> len, woff;
        size_t pos = 0;
        uint64_t words = 0;
        uint32_t max_len = 0;
        while (pos + 4 <= nbytes && (max_reads < 0 || (int64_t)len.size() < max_reads)) {      // NucleotideBitsReader.next(): length, then the bytes
            const uint32_t l = ((uint32_t)p[pos] << 24) | ((uint32_t)p[pos + 1] << 16) | ((uint32_t)p[pos + 2] << 8) | (uint32_t)p[pos + 3];
            RB_REQUIRE(l < (1u << 30), "rb_batch_create_nbits: record %zu has an invalid length %u", len.size(), l);
            const size_t nb = ((size_t)l + 3) / 4;
            if (l == 0) break;                                       // an empty sequence ends the iteration: fin.read(new byte[0]) is 0, not > 0, and next() returns null (NucleotideBitsReader.java:41-44)
            if (pos + 4 + nb > nbytes) break;                        // truncated record: the reader returns null
            rec_off.push_back((int64_t)(pos + 4)); len.push_back(l); woff.push_back((uint32_t)words);
            words += ((uint64_t)l + 31) / 32;
            RB_REQUIRE(words < 0xFFFFFFF0ull, "rb_batch_create_nbits: batch too large (> 2^32 words)");
            max_len = std::max(max_len, l);
            pos += 4 + nb;
        }
        woff.push_back((uint32_t)words);
        const int64_t n_reads = (int64_t)len.size();
        b = new rb_batch();
        b->device = device; b->n_reads = n_reads; b->n_words = (int64_t)words; b->max_len = max_len;
        b->n_bases = 0; for (uint32_t l : len) b->n_bases += l;
        {
            uint32_t wpr = n_reads ? (len[0] + 31) / 32 : 0;
            for (int64_t i = 0; i < n_reads && wpr; ++i) if ((len[(size_t)i] + 31) / 32 != wpr) wpr = 0;
            b->wpr_uniform = wpr;
        }
        alloc_batch_arrays(b);
        b->h_woff = woff;
        RB_HIP(hipMemcpy(b->woff, woff.data(), ((size_t)n_reads + 1) * 4, hipMemcpyHostToDevice));
        if (n_reads) RB_HIP(hipMemcpy(b->len, len.data(), (size_t)n_reads * 4, hipMemcpyHostToDevice));
        if (words) {
            RB_HIP(hipMalloc(&d_raw, std::max<size_t>(pos, 1) + 8));
            RB_HIP(hipMalloc(&d_off, (size_t)n_reads * 8));
            RB_HIP(hipMemcpy(d_raw, p, pos, hipMemcpyHostToDevice));
            RB_HIP(hipMemcpy(d_off, rec_off.data(), (size_t)n_reads * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_decode_nbits, dim3(blocks_for((int64_t)words)), dim3(TPB), 0, 0, d_raw, d_off, b->woff, b->len, n_reads, (int64_t)words,
                               b->codes, b->valid, b->word_read);
            RB_HIP(hipGetLastError());
            RB_HIP(hipDeviceSynchronize());
        }
        if (d_raw) (void)hipFree(d_raw);
        if (d_off) (void)hipFree(d_off);
        if (consumed) *consumed = pos;
        *out = b;
        return RB_OK;
    } catch (const HipError &e) {
        if (d_raw) (void)hipFree(d_raw);
        if (d_off) (void)hipFree(d_off);
        if (b) rb_batch_destroy(b);
        return e.code;
    } catch (const std::bad_alloc &) { if (b) rb_batch_destroy(b); set_error("host allocation failed"); return RB_ERR_NOMEM; }
}

int rb_nbits_encode(const char *seq, const int64_t *offsets, int64_t n_reads, void *out, size_t cap, size_t *written) {
    try {
        RB_REQUIRE(offsets && written && n_reads >= 0 && (seq || n_reads == 0), "rb_nbits_encode: null argument");
        size_t need = 0;
        for (int64_t i = 0; i < n_reads; ++i) { const int64_t l = offsets[i + 1] - offsets[i]; RB_REQUIRE(l >= 0 && l < (1ll << 30), "rb_nbits_encode: bad length"); need += 4 + ((size_t)l + 3) / 4; }
        *written = need;
        if (!out) return RB_OK;                                          // size query
        RB_REQUIRE(cap >= need, "rb_nbits_encode: buffer of %zu bytes, %zu needed", cap, need);
        uint8_t *o = static_cast<uint8_t *>(out);
        for (int64_t i = 0; i < n_reads; ++i) {
            const char *s = seq + offsets[i];
            const uint32_t l = (uint32_t)(offsets[i + 1] - offsets[i]);
            *o++ = (uint8_t)(l >> 24); *o++ = (uint8_t)(l >> 16); *o++ = (uint8_t)(l >> 8); *o++ = (uint8_t)l;   // intToFourBytes: big endian
            for (uint32_t q = 0; q < l; q += 4) {
                uint32_t v = 0;
                for (uint32_t j = 0; j < 4; ++j) {
                    uint32_t code = 0;                                   // missing bases of the last byte are 0 (SeqBitsUtils.java:252-258)
                    if (q + j < l) switch (s[q + j]) {
                        case 'A': case 'a': code = 0; break;
                        case 'C': case 'c': code = 1; break;
                        case 'G': case 'g': code = 2; break;
                        case 'T': case 't': case 'U': case 'u': code = 3; break;
                        default:   // the reference writes a RANDOM base here (SeqBitsUtils.java:154-155): not reproducible
                            RB_REQUIRE(false, "rb_nbits_encode: read %lld has a non-ACGTU base at %u (the .nbits format has no code for it)", (long long)i, q + j);
                    }
                    v = (v << 2) | code;
                }
                *o++ = (uint8_t)(v ^ 0x80u);                             // value - 128 as a signed byte
            }
        }
        return RB_OK;
    } catch (const HipError &e) { return e.code; }
}

}  // extern "C"
