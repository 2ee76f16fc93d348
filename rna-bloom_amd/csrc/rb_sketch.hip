// rb_sketch.hip — minimizer and strobemer extraction over long reads (BASELINE config 5).
// Hash-only work: one read is independent of every other, so multi-GPU = replicas over read shards.
#include "rb_pipeline.hpp"

using namespace rb;

namespace {

// ntHash of EVERY window of a read (no segmentation): unusable bases contribute seed 0, exactly like
// the zero rows of seedTab (R/bloom/hash/NTHash.java:133-166).  mode 0 forward, 1 canonical, 2 RC.
__global__ void k_hash_all(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid, const uint32_t *__restrict__ rnz,
                           const uint32_t *__restrict__ word_read, const uint32_t *__restrict__ woff,
                           const uint32_t *__restrict__ len, int64_t n_words, int k, int mode,
                           const int64_t *__restrict__ koff, uint64_t *__restrict__ out) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const uint32_t r = word_read[w], wr = woff[r], L = len[r];
    const uint32_t b0 = (uint32_t)(w - wr) * 32u, uk = (uint32_t)k;
    if ((uint64_t)b0 + uk > L) return;
    const uint64_t bend64 = (uint64_t)b0 + 32u + uk - 1u;
    const uint32_t bend = bend64 < L ? (uint32_t)bend64 : L;
    const uint64_t *cw = codes + wr;
    const uint32_t *vw = valid + wr;
    const uint32_t *zw = rnz ? rnz + wr : vw;      // reverse-strand seed by `ch & 7`: non-zero for some letters outside ACGTU (rb_batch::rnz)
    uint64_t f = 0, rv = 0;
    uint32_t filled = 0;
    for (uint32_t b = b0; b < bend; ++b) {
        const bool ok = (vw[b >> 5] >> (b & 31u)) & 1u, rok = (zw[b >> 5] >> (b & 31u)) & 1u;
        const uint32_t c = (uint32_t)(cw[b >> 5] >> (2u * (b & 31u))) & 3u;
        const uint64_t s_in = ok ? seed_of(c) : 0ull, sc_in = rok ? seed_of(3u - c) : 0ull;
        if (filled < uk) { f = rotl(f, 1) ^ s_in; rv ^= rotl(sc_in, filled); ++filled; }
        else {
            const uint32_t bo = b - uk;
            const bool oko = (vw[bo >> 5] >> (bo & 31u)) & 1u, roko = (zw[bo >> 5] >> (bo & 31u)) & 1u;
            const uint32_t oc = (uint32_t)(cw[bo >> 5] >> (2u * (bo & 31u))) & 3u;
            const uint64_t s_out = oko ? seed_of(oc) : 0ull, sc_out = roko ? seed_of(3u - oc) : 0ull;
            f = rotl(f, 1) ^ rotl(s_out, uk) ^ s_in;
            rv = rotr(rv, 1) ^ rotr(sc_out, 1) ^ rotl(sc_in, uk - 1u);
        }
        if (filled >= uk) out[koff[r] + (b - uk + 1u)] = mode == 0 ? f : mode == 2 ? rv : canonical(f, rv);
    }
}

// MinimizerHashIterator.next() for every window of a read, one thread per read: LongRollingWindow
// (R/util/LongRollingWindow.java:23-83) is replayed exactly — the value of a window is its signed minimum, but WHICH of
// several equal minima the window reports depends on the history (a new value replaces the minimum only if strictly
// smaller, :55-57; when the minimum leaves, the circular buffer is rescanned in ARRAY order, :52-54, 60-69) and
// nextMinimizer() keys on that position.  The buffer is not materialised: after the roll that brings in k-mer
// p + w - 1, slot s holds the k-mer q of the window with q mod w == s.
__global__ void k_minimizers(const uint64_t *__restrict__ h, const int64_t *__restrict__ koff, const int64_t *__restrict__ moff,
                             int64_t n_reads, int64_t total, int w, uint64_t *__restrict__ out_hash, int64_t *__restrict__ out_pos) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int64_t nwin = moff[r + 1] - moff[r];
    if (nwin <= 0) return;
    const uint64_t *hh = h + koff[r];
    uint64_t *oh = out_hash + moff[r];
    int64_t *op = out_pos ? out_pos + moff[r] : nullptr;
    // first window: k-mers 0 .. w-2 in slots 0 .. w-2 and a 0 placeholder in slot w-1 (MinimizerHashIterator.java:54-62)
    int min_index = 0;
    int64_t mv = w > 1 ? (int64_t)hh[0] : 0;
    for (int i = 1; i < w; ++i) { const int64_t v = i < w - 1 ? (int64_t)hh[i] : 0; if (v < mv) { mv = v; min_index = i; } }
    int index = w - 2;
    if (w == 1) { min_index = 0; index = -1; }
    for (int64_t p = 0; p < nwin; ++p) {
        const int64_t v = (int64_t)hh[p + w - 1];
        if (++index >= w) index = 0;
        if (min_index == index) {                      // the minimum is overwritten: rescan in array order
            int best = 0; int64_t bv = 0;
            for (int sl = 0; sl < w; ++sl) {
                int64_t d = ((int64_t)sl - p) % w; if (d < 0) d += w;
                const int64_t x = (int64_t)hh[p + d];
                if (sl == 0 || x < bv) { bv = x; best = sl; }
            }
            min_index = best; mv = bv;
        } else if (v < mv) { min_index = index; mv = v; }
        oh[p] = (uint64_t)mv;
        if (op) { int64_t d = ((int64_t)min_index - p) % w; if (d < 0) d += w; op[p] = p + d; }
    }
    (void)total;
}

// one thread per strobemer: StrobeHashIterator.getInterval (R/bloom/hash/StrobeHashIterator.java:133-164)
__global__ void k_strobemers(const uint64_t *__restrict__ h, const int64_t *__restrict__ koff, const int64_t *__restrict__ soff,
                             int64_t n_reads, int64_t total, int k, int n, int wmin, int wmax, uint64_t *__restrict__ out_hash,
                             int32_t *__restrict__ out_start, int32_t *__restrict__ out_end) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (soff[mid] <= t) lo = mid; else hi = mid; }
    const int64_t p = t - soff[lo];
    const uint64_t *hh = h + koff[lo];
    const int64_t nk = koff[lo + 1] - koff[lo];
    uint64_t sh = hh[p];
    int64_t last = p;
    for (int s = 0; s < n - 1; ++s) {
        int64_t pos2 = p + (int64_t)s * wmax + wmin;
        uint64_t pos2k = hh[pos2];
        uint64_t hv = combine(sh, pos2k);
        int64_t end = p + (int64_t)s * wmax + wmax;
        if (end > nk) end = nk;
        for (int64_t i = pos2 + 1; i < end; ++i) {
            const uint64_t alt = hh[i];
            if (alt == pos2k) pos2 = i;
            else {
                const uint64_t h2 = combine(sh, alt);
                if (hv >= h2) { pos2 = i; pos2k = alt; hv = h2; }     // Long.compareUnsigned(h, h2) >= 0
            }
        }
        sh = hv; last = pos2;
    }
    out_hash[t] = sh;
    out_start[t] = (int32_t)p;
    out_end[t] = (int32_t)(last + k - 1);
}


// StrobeHashIterator.next / get and CanonicalStrobeHashIterator.next / get (R/bloom/hash/StrobeHashIterator.java:73-131,
// R/bloom/hash/CanonicalStrobeHashIterator.java:79-140): order-n randstrobes, argmin_unsigned with ties to the right;
// RB_STROBE_SLIDE = the forward iterator's get(): the chosen strobe slides across later equal k-mer hashes; canonical:
// strobes are chosen on the forward hashes, the reverse hashes of the same positions are combined back to front and the
// SIGNED minimum of the two is returned (:107).  One thread per strobemer; out_pos: n positions each.
__global__ void k_randstrobes(const uint64_t *__restrict__ hf, const uint64_t *__restrict__ hr, const int64_t *__restrict__ koff,
                              const int64_t *__restrict__ soff, int64_t n_reads, int64_t total, int n, int wmin, int wmax, int flags,
                              uint64_t *__restrict__ out_hash, int32_t *__restrict__ out_pos) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (soff[mid] <= t) lo = mid; else hi = mid; }
    const int64_t p = t - soff[lo];
    const uint64_t *f = hf + koff[lo];
    const int64_t nk = koff[lo + 1] - koff[lo];
    const bool canonical = flags & RB_STROBE_CANONICAL, slide = (flags & RB_STROBE_SLIDE) && !canonical;
    uint64_t sh = f[p];
    int32_t positions[RB_MAX_STROBES];
    positions[0] = (int32_t)p;
    for (int s = 0; s < n - 1; ++s) {
        int64_t pos2 = p + (int64_t)s * wmax + wmin;
        uint64_t pos2k = f[pos2];
        uint64_t hv = combine(sh, pos2k);
        int64_t end = p + (int64_t)s * wmax + wmax;
        if (end > nk) end = nk;
        for (int64_t i = pos2 + 1; i < end; ++i) {
            const uint64_t alt = f[i];
            if (slide && alt == pos2k) pos2 = i;
            else {
                const uint64_t h2 = combine(sh, alt);
                if (hv >= h2) { pos2 = i; pos2k = alt; hv = h2; }     // Long.compareUnsigned(h, h2) >= 0
            }
        }
        sh = hv;
        positions[s + 1] = (int32_t)pos2;
    }
    if (canonical) {
        const uint64_t *r = hr + koff[lo];
        uint64_t rs = r[positions[n - 1]];
        for (int s = n - 2; s >= 0; --s) rs = combine(r[positions[s]], rs);
        sh = smin(sh, rs);
    }
    out_hash[t] = sh;
    if (out_pos) for (int s = 0; s < n; ++s) out_pos[t * n + s] = positions[s];
}

// Strobe3HashIterator / CanonicalStrobe3HashIterator (R/bloom/hash/Strobe3HashIterator.java:78-151,
// R/bloom/hash/CanonicalStrobe3HashIterator.java:85-225): middle k-mer p, upstream strobe in [p-wMax+1, p-wMin],
// downstream strobe in [p+wMin, p+wMax); comparisons unsigned, strict except where the reference has >= .
__global__ void k_strobe3(const uint64_t *__restrict__ hf, const uint64_t *__restrict__ hr, const int64_t *__restrict__ koff,
                          const int64_t *__restrict__ soff, int64_t n_reads, int64_t total, int wmin, int wmax, int canonical,
                          uint64_t *__restrict__ out_hash, int32_t *__restrict__ out_pos) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (soff[mid] <= t) lo = mid; else hi = mid; }
    const int64_t pos = t - soff[lo] + (canonical ? wmax : wmin);
    const uint64_t *f = hf + koff[lo];
    const int64_t nk = koff[lo + 1] - koff[lo];
    const uint64_t fk = f[pos];
    int64_t p1 = pos - wmax + 1 > 0 ? pos - wmax + 1 : 0;
    uint64_t h1 = combine(f[p1], fk);
    int64_t end = pos - wmin + 1;
    for (int64_t i = p1 + 1; i < end; ++i) { const uint64_t h = combine(f[i], fk); if (h1 > h) { p1 = i; h1 = h; } }
    int64_t p3 = pos + wmin;
    uint64_t h3 = combine(h1, f[p3]);
    end = pos + wmax < nk ? pos + wmax : nk;
    for (int64_t i = p3 + 1; i < end; ++i) {
        const uint64_t h = combine(h1, f[i]);
        if (canonical ? h3 >= h : h3 > h) { p3 = i; h3 = h; }
    }
    uint64_t hv = h3;
    if (canonical) {
        const uint64_t *r = hr + koff[lo];
        const uint64_t rk = r[pos];
        int64_t q3 = pos + wmin;
        uint64_t rh3 = combine(r[q3], rk);
        int64_t rend = pos + wmax < nk ? pos + wmax : nk;
        for (int64_t i = q3 + 1; i < rend; ++i) { const uint64_t h = combine(r[i], rk); if (rh3 >= h) { q3 = i; rh3 = h; } }
        int64_t q1 = pos - wmax + 1 > 0 ? pos - wmax + 1 : 0;
        uint64_t rh1 = combine(rh3, r[q1]);
        rend = pos - wmin + 1;
        for (int64_t i = q1 + 1; i < rend; ++i) { const uint64_t h = combine(rh3, r[i]); if (rh1 > h) { q1 = i; rh1 = h; } }
        if (h3 > rh1) { hv = rh1; p1 = q1; p3 = q3; }
    }
    out_hash[t] = hv;
    if (out_pos) { out_pos[3 * t] = (int32_t)p1; out_pos[3 * t + 1] = (int32_t)pos; out_pos[3 * t + 2] = (int32_t)p3; }
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile kernels (the ones the entry points use; the one-thread-per-output kernels above remain for windows too wide for LDS).
//
// A workgroup takes SK_TILE consecutive outputs of ONE read (host-built tile list: a read of c outputs has ceil(c / SK_TILE)
// tiles), stages the k-mer hashes the tile's windows cover in LDS once (coalesced 8-byte loads: TILE + window span values,
// instead of every thread walking its 50-120 window values through the vector caches), and every lane works on SK_PER
// outputs, lane l on outputs l, l + 256, ...: in step i of a window scan the 64 lanes of a wavefront read 64 CONSECUTIVE LDS
// words (conflict free).  The scan itself is the reference's loop: an argmin of combine(...) whose left operand differs
// from strobemer to strobemer, so there is no sliding-window shortcut — W combine + compare steps per window are the
// algorithmic work; a lane-per-strobemer loop does them at ~10 VALU instructions per step, a wavefront-wide reduction per
// strobemer would cost 6 DPP rounds of the same comparison for every window (more instructions, not fewer).
// combine(a, b) = a ^ (b + C + (a << 6) + (b >>> 2)) (HashFunction.java:260-263) is split into the part that depends on the
// window value only, g(b) = b + (b >>> 2), staged in LDS next to b, and the per-strobemer constant C + (a << 6).
constexpr int SK_TPB = 256, SK_PER = 4, SK_TILE = SK_TPB * SK_PER;
struct SkTile { uint32_t read, first; };
constexpr uint64_t SK_C = 0xFFFFFFFF9E3779B9ull;     // the sign-extended int literal of combineHashValues
__device__ __forceinline__ uint64_t sk_g(uint64_t b) { return b + (b >> 2); }

// StrobeHashIterator.next / get / getInterval and CanonicalStrobeHashIterator.next / get (see k_randstrobes above).
// out_pos: n positions per strobemer, or out_start / out_end (getInterval's interval), or neither.
// get / getInterval slide the chosen strobe across later EQUAL k-mer hashes (:96-164); an equal k-mer hash gives an equal
// combined hash, which the `>=` of the comparison takes anyway — same strobe, same hash as next() — so one loop serves all.
// The scan of a window that lies inside the read runs on the HIGH 32 bits of the combined hashes (7 VALU instructions per
// candidate instead of 9-10: add with carry, xor, compare, two selects, equality); two candidates that agree in those 32 bits
// (equal k-mers, or once in 2^32 / W otherwise) send the lane to the exact 64-bit loop for that window.
__global__ void __launch_bounds__(SK_TPB) k_randstrobes_tile(const uint64_t *__restrict__ hf, const uint64_t *__restrict__ hr, const int64_t *__restrict__ koff,
                                                             const int64_t *__restrict__ soff, const SkTile *__restrict__ tiles, int n, int wmin, int wmax,
                                                             int canonical, int k, uint64_t *__restrict__ out_hash, int32_t *__restrict__ out_pos,
                                                             int32_t *__restrict__ out_start, int32_t *__restrict__ out_end) {
    extern __shared__ uint64_t sk_lds[];
    const SkTile tl = tiles[blockIdx.x];
    const int64_t kb = koff[tl.read], ob = soff[tl.read];
    const int nk = (int)(koff[tl.read + 1] - kb), cnt = (int)(soff[tl.read + 1] - ob);
    const int t0 = (int)tl.first, n_here = min(SK_TILE, cnt - t0), span = wmax * (n - 1);
    const int n_stage = min(n_here + span, nk - t0);                 // k-mers [t0, t0 + n_stage) of the read
    uint64_t *s_f = sk_lds, *s_g = sk_lds + (SK_TILE + span), *s_r = s_g + (SK_TILE + span);
    for (int i = threadIdx.x; i < n_stage; i += SK_TPB) {
        const uint64_t v = hf[kb + t0 + i];
        s_f[i] = v; s_g[i] = sk_g(v);
        if (canonical) s_r[i] = hr[kb + t0 + i];
    }
    __syncthreads();
    const int lim = nk - t0;                                          // local index bound of the read's k-mers
    const int wn = wmax - wmin;                                       // candidates of a window that is not cut by the read's end
    for (int j = 0; j < SK_PER; ++j) {
        const int t = (int)threadIdx.x + j * SK_TPB;                  // local index of the strobemer's own k-mer
        if (t >= n_here) break;
        uint64_t sh = s_f[t];
        int pos[RB_MAX_STROBES];
        pos[0] = t;
#pragma unroll
        for (int s = 0; s < RB_MAX_STROBES - 1; ++s) {
            pos[s + 1] = 0;
            if (s < n - 1) {
                const int lo = t + s * wmax + wmin;
                const uint64_t a1 = SK_C + (sh << 6);
                int pos2 = lo;
                uint64_t hv = sh ^ (s_g[lo] + a1);
                bool exact = true;
                if (__all(lo + wn <= lim)) {                          // the whole wavefront's windows are complete: uniform trip count
                    const uint32_t shh = (uint32_t)(sh >> 32);
                    uint32_t best = (uint32_t)(hv >> 32);
                    bool amb = false;
#pragma unroll 8
                    for (int i = 1; i < wn; ++i) {
                        const uint32_t x = shh ^ (uint32_t)((s_g[lo + i] + a1) >> 32);
                        amb |= x == best;
                        if (x <= best) { best = x; pos2 = lo + i; }
                    }
                    if (!amb) { hv = sh ^ (s_g[pos2] + a1); exact = false; }
                    else pos2 = lo;
                }
                if (exact) {
                    const int end = min(lo + wn, lim);
                    for (int i = lo + 1; i < end; ++i) {
                        const uint64_t h2 = sh ^ (s_g[i] + a1);
                        if (hv >= h2) { pos2 = i; hv = h2; }          // Long.compareUnsigned(h, h2) >= 0
                    }
                }
                sh = hv;
                pos[s + 1] = pos2;
            }
        }
        if (canonical) {                                              // reverse hashes of the same positions, combined back to front (:99-107)
            uint64_t rs = 0;
#pragma unroll
            for (int s = RB_MAX_STROBES - 1; s >= 0; --s)
                if (s < n) rs = (s == n - 1) ? s_r[pos[s]] : combine(s_r[pos[s]], rs);
            sh = smin(sh, rs);
        }
        const int64_t o = ob + t0 + t;
        out_hash[o] = sh;
        if (out_pos) {
#pragma unroll
            for (int s = 0; s < RB_MAX_STROBES; ++s) if (s < n) out_pos[o * n + s] = pos[s] + t0;
        }
        if (out_start) {
            int last = t;
#pragma unroll
            for (int s = 1; s < RB_MAX_STROBES; ++s) if (s < n) last = pos[s];
            out_start[o] = t + t0;
            out_end[o] = last + t0 + k - 1;
        }
    }
}

// Strobe3HashIterator / CanonicalStrobe3HashIterator (see k_strobe3 above) on LDS-staged hashes: the tile's middle k-mers are
// pos = off + t0 .. , their windows cover k-mers [pos - wmax + 1, pos + wmax).
__global__ void __launch_bounds__(SK_TPB) k_strobe3_tile(const uint64_t *__restrict__ hf, const uint64_t *__restrict__ hr, const int64_t *__restrict__ koff,
                                                         const int64_t *__restrict__ soff, const SkTile *__restrict__ tiles, int wmin, int wmax, int canonical,
                                                         uint64_t *__restrict__ out_hash, int32_t *__restrict__ out_pos) {
    extern __shared__ uint64_t sk_lds[];
    const SkTile tl = tiles[blockIdx.x];
    const int64_t kb = koff[tl.read], ob = soff[tl.read];
    const int nk = (int)(koff[tl.read + 1] - kb), cnt = (int)(soff[tl.read + 1] - ob);
    const int t0 = (int)tl.first, n_here = min(SK_TILE, cnt - t0), off = canonical ? wmax : wmin;
    const int base = max(0, t0 + off - wmax + 1);                     // first staged k-mer
    const int n_stage = min(nk, t0 + off + n_here - 1 + wmax) - base;
    const int cap = SK_TILE + 2 * wmax;
    uint64_t *s_f = sk_lds, *s_g = sk_lds + cap, *s_r = s_g + cap;
    for (int i = threadIdx.x; i < n_stage; i += SK_TPB) {
        const uint64_t v = hf[kb + base + i];
        s_f[i] = v; s_g[i] = sk_g(v);
        if (canonical) s_r[i] = hr[kb + base + i];
    }
    __syncthreads();
    for (int j = 0; j < SK_PER; ++j) {
        const int t = (int)threadIdx.x + j * SK_TPB;
        if (t >= n_here) break;
        const int pos = t0 + t + off;                                 // k-mer position in the read
        const int lp = pos - base;                                    // its LDS index
        // upstream strobe: argmin of combine(f[i], fk) over [max(0, pos - wmax + 1), pos - wmin], strict (leftmost of equals)
        int p1 = max(pos - wmax + 1, 0) - base;
        const uint64_t b1 = s_g[lp] + SK_C;                           // fk + C + (fk >>> 2)
        uint64_t h1; { const uint64_t a = s_f[p1]; h1 = a ^ (b1 + (a << 6)); }
        for (int i = p1 + 1; i <= lp - wmin; ++i) { const uint64_t a = s_f[i]; const uint64_t h = a ^ (b1 + (a << 6)); if (h1 > h) { p1 = i; h1 = h; } }
        // downstream strobe: argmin of combine(h1, f[i]) over [pos + wmin, min(pos + wmax, nk)), rightmost of equals when canonical
        int p3 = lp + wmin;
        const uint64_t a1 = SK_C + (h1 << 6);
        uint64_t h3 = h1 ^ (s_g[p3] + a1);
        const int end = min(pos + wmax, nk) - base;
        if (canonical) { for (int i = p3 + 1; i < end; ++i) { const uint64_t h = h1 ^ (s_g[i] + a1); if (h3 >= h) { p3 = i; h3 = h; } } }
        else { for (int i = p3 + 1; i < end; ++i) { const uint64_t h = h1 ^ (s_g[i] + a1); if (h3 > h) { p3 = i; h3 = h; } } }
        uint64_t hv = h3;
        if (canonical) {                                              // the reverse strand's chain: downstream first, then upstream (:118-143)
            const uint64_t rk = s_r[lp];
            int q3 = lp + wmin;
            const uint64_t rb = rk + SK_C + (rk >> 2);
            uint64_t rh3; { const uint64_t a = s_r[q3]; rh3 = a ^ (rb + (a << 6)); }
            for (int i = q3 + 1; i < end; ++i) { const uint64_t a = s_r[i]; const uint64_t h = a ^ (rb + (a << 6)); if (rh3 >= h) { q3 = i; rh3 = h; } }
            int q1 = max(pos - wmax + 1, 0) - base;
            const uint64_t ra = SK_C + (rh3 << 6);
            uint64_t rh1 = rh3 ^ (sk_g(s_r[q1]) + ra);
            for (int i = q1 + 1; i <= lp - wmin; ++i) { const uint64_t h = rh3 ^ (sk_g(s_r[i]) + ra); if (rh1 > h) { q1 = i; rh1 = h; } }
            if (h3 > rh1) { hv = rh1; p1 = q1; p3 = q3; }
        }
        const int64_t o = ob + t0 + t;
        out_hash[o] = hv;
        if (out_pos) { out_pos[3 * o] = p1 + base; out_pos[3 * o + 1] = pos; out_pos[3 * o + 2] = p3 + base; }
    }
}

// Window minimizers, parallel over windows and still the exact replay of LongRollingWindow: the VALUE of a window is its signed
// minimum whatever happened before; its POSITION is history-free too whenever the minimum occurs once in the window (the
// buffer's min_index can only point at a slot holding the minimum) and in window 0 (array order = position order: leftmost).
// Only windows in which the minimum occurs at two or more positions depend on the history (a new value replaces the minimum
// only if strictly smaller, :55-57; a minimum that leaves is replaced by the first in SLOT order, :60-69).  Pass 1 (this
// kernel): value, leftmost position and a tie flag per window.  Pass 2 (k_minimizer_ties): every maximal run of tied windows
// is replayed by one lane from the tie-free window in front of it, whose state is known.
__global__ void __launch_bounds__(SK_TPB) k_minimizers_tile(const uint64_t *__restrict__ h, const int64_t *__restrict__ koff, const int64_t *__restrict__ moff,
                                                            const SkTile *__restrict__ tiles, int w, uint64_t *__restrict__ out_hash,
                                                            int64_t *__restrict__ out_pos, uint8_t *__restrict__ tie) {
    extern __shared__ uint64_t sk_lds[];
    const SkTile tl = tiles[blockIdx.x];
    const int64_t kb = koff[tl.read], ob = moff[tl.read];
    const int cnt = (int)(moff[tl.read + 1] - ob);
    const int t0 = (int)tl.first, n_here = min(SK_TILE, cnt - t0), n_stage = n_here + w - 1;
    for (int i = threadIdx.x; i < n_stage; i += SK_TPB) sk_lds[i] = h[kb + t0 + i];
    __syncthreads();
    for (int j = 0; j < SK_PER; ++j) {
        const int t = (int)threadIdx.x + j * SK_TPB;
        if (t >= n_here) break;
        int64_t mv = (int64_t)sk_lds[t];
        int mp = t; bool tied = false;
        for (int i = 1; i < w; ++i) {
            const int64_t v = (int64_t)sk_lds[t + i];
            if (v < mv) { mv = v; mp = t + i; tied = false; }
            else if (v == mv) tied = true;
        }
        const int64_t o = ob + t0 + t;
        out_hash[o] = (uint64_t)mv;
        if (out_pos) out_pos[o] = (int64_t)(mp + t0);
        tie[o] = (tied && t0 + t > 0) ? 1u : 0u;                     // window 0 reports the leftmost minimum
    }
}
// one lane per window; the head of a run of tied windows replays the run
__global__ void k_minimizer_ties(const uint64_t *__restrict__ h, const int64_t *__restrict__ koff, const int64_t *__restrict__ moff, int64_t n_reads,
                                 int64_t total, int w, const uint8_t *__restrict__ tie, int64_t *__restrict__ out_pos) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total || !tie[o]) return;
    int64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (moff[mid] <= o) lo = mid; else hi = mid; }
    const int64_t p0 = o - moff[lo], nwin = moff[lo + 1] - moff[lo];
    if (p0 > 0 && tie[o - 1]) return;                                // not the head of its run (window 0 is never tied)
    const uint64_t *hh = h + koff[lo];
    int64_t *op = out_pos + moff[lo];
    int64_t cur = op[p0 - 1];                                         // tie-free window (or window 0): min_index is its unique / leftmost minimum
    for (int64_t q = p0; q < nwin && tie[moff[lo] + q]; ++q) {
        const int64_t e = q + w - 1;                                  // the k-mer rolled in
        if (cur == q - 1) {                                           // the minimum is overwritten: rescan in slot order, strictly smaller wins (:60-69)
            int64_t best = -1, bv = 0;
            for (int sl = 0; sl < w; ++sl) {
                int64_t d = ((int64_t)sl - q) % w; if (d < 0) d += w;  // the window's k-mer in slot sl
                const int64_t x = (int64_t)hh[q + d];
                if (best < 0 || x < bv) { bv = x; best = q + d; }
            }
            cur = best;
        } else if ((int64_t)hh[e] < (int64_t)hh[cur]) cur = e;
        op[q] = cur;
    }
}

// SeqSubsampler.kmerBased pair hashes (R/util/SeqSubsampler.java:176-179, 266-268)
__global__ void k_kmer_pairs(const uint64_t *__restrict__ hf, const uint64_t *__restrict__ hr, const int64_t *__restrict__ koff,
                             const int64_t *__restrict__ poff, int64_t n_reads, int64_t total, int shift, int canonical,
                             uint64_t *__restrict__ out_hash) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (poff[mid] <= t) lo = mid; else hi = mid; }
    const int64_t i = t - poff[lo];
    const uint64_t *f = hf + koff[lo];
    uint64_t pf = combine(f[i], f[i + shift]);
    if (canonical) { const uint64_t *r = hr + koff[lo]; pf = smin(pf, combine(r[i + shift], r[i])); }
    out_hash[t] = pf;
}

// window minimizers -> flags: window p of a read opens a new minimizer iff it is the first window or the position of
// its minimum differs from the previous window's (MinimizerHashIterator.nextMinimizer, :97-112; the position of the
// leftmost minimum never moves left)
__global__ void k_minimizer_flags(const int64_t *__restrict__ mpos, const int64_t *__restrict__ moff, int64_t n_reads, int64_t total,
                                  uint32_t *__restrict__ flag) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (moff[mid] <= t) lo = mid; else hi = mid; }
    flag[t] = (t == moff[lo] || mpos[t] != mpos[t - 1]) ? 1u : 0u;
}
__global__ void k_minimizer_compact(const uint64_t *__restrict__ mh, const int64_t *__restrict__ mpos, const uint32_t *__restrict__ flag,
                                    const uint32_t *__restrict__ slot, int64_t total, uint64_t *__restrict__ out_hash, int64_t *__restrict__ out_pos) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total || !flag[t]) return;
    out_hash[slot[t]] = mh[t];
    if (out_pos) out_pos[slot[t]] = mpos[t];
}
// number of minimizers before each read's first window (for the per-read offsets)
__global__ void k_gather_u32(const uint32_t *__restrict__ slot, const int64_t *__restrict__ moff, int64_t n_reads, int64_t total,
                             uint32_t n_total, int64_t *__restrict__ out) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    out[r] = moff[r] < total ? (int64_t)slot[moff[r]] : (int64_t)n_total;
}
// getMinimizersSet for reads with numKmers <= windowSize: min_signed(stale, every k-mer hash) (R/util/GraphUtils.java:2480-2494)
__global__ void k_short_read_min(const uint64_t *__restrict__ h, const int64_t *__restrict__ koff, const int64_t *__restrict__ kread,
                                 int64_t n_short, const uint64_t *__restrict__ stale, uint64_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_short) return;
    const int64_t r = kread[i];
    uint64_t m = stale ? stale[r] : 0ull;
    for (int64_t q = koff[r]; q < koff[r + 1]; ++q) m = smin(m, h[q]);
    out[i] = m;
}
__global__ void k_scatter_u64(uint64_t *__restrict__ dst, const int64_t *__restrict__ where, const uint64_t *__restrict__ src, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[where[i]] = src[i];
}
__global__ void k_flip_sign(uint64_t *v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] ^= 0x8000000000000000ull;
}
__global__ void k_set_keys(const uint64_t *__restrict__ val, const int64_t *__restrict__ off, int64_t n_reads, int64_t total,
                           uint64_t *__restrict__ key_read) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (off[mid] <= t) lo = mid; else hi = mid; }
    key_read[t] = (uint64_t)lo;
}
__global__ void k_unique_flags(const uint64_t *__restrict__ read, const uint64_t *__restrict__ val, int64_t total, uint32_t *__restrict__ flag) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    flag[t] = (t == 0 || read[t] != read[t - 1] || val[t] != val[t - 1]) ? 1u : 0u;
}
__global__ void k_unique_compact(const uint64_t *__restrict__ read, const uint64_t *__restrict__ val, const uint32_t *__restrict__ flag,
                                 const uint32_t *__restrict__ slot, int64_t total, uint64_t *__restrict__ out, unsigned long long *__restrict__ per_read) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total || !flag[t]) return;
    out[slot[t]] = val[t] ^ 0x8000000000000000ull;
    atomicAdd(&per_read[read[t]], 1ull);
}

struct BatchGuard { rb_batch *b; ~BatchGuard() { if (b) rb_batch_destroy(b); } };

// tile list of the tile kernels: read r with c = off[r+1] - off[r] outputs gets ceil(c / SK_TILE) tiles
void make_tiles(const int64_t *off, int64_t n_reads, DevBuf &d_tiles, uint32_t *n_tiles) {
    std::vector<SkTile> tiles;
    tiles.reserve((size_t)(off[n_reads] / SK_TILE + n_reads));
    for (int64_t r = 0; r < n_reads; ++r)
        for (int64_t f = 0; f < off[r + 1] - off[r]; f += SK_TILE) tiles.push_back({(uint32_t)r, (uint32_t)f});
    RB_REQUIRE(tiles.size() < 0x7FFFFFFFull, "sketch: too many tiles in one call");
    *n_tiles = (uint32_t)tiles.size();
    d_tiles.reserve(std::max<size_t>(tiles.size(), 1) * sizeof(SkTile));
    if (!tiles.empty()) RB_HIP(hipMemcpy(d_tiles.p, tiles.data(), tiles.size() * sizeof(SkTile), hipMemcpyHostToDevice));
}
constexpr size_t SK_LDS_MAX = 60 * 1024;      // windows wider than this fall back to the one-thread-per-output kernels
bool sk_force_simple() { const char *e = getenv("RB_SKETCH_SIMPLE"); return e && atoi(e) != 0; }
// window minimizers of all reads: d_h = all-window hashes, d_moff = per-read window offsets (device), moff = the same on the host
void launch_minimizers(const uint64_t *d_h, const int64_t *d_koff, const int64_t *d_moff, const int64_t *moff, int64_t n_reads, int64_t total, int w,
                       uint64_t *d_oh, int64_t *d_op) {
    const size_t lds = ((size_t)SK_TILE + (size_t)w) * 8;
    if (lds > SK_LDS_MAX || sk_force_simple() || !d_op) {
        hipLaunchKernelGGL(k_minimizers, dim3(blocks_for(n_reads, 64)), dim3(64), 0, 0, d_h, d_koff, d_moff, n_reads, total, w, d_oh, d_op);
        RB_HIP(hipGetLastError());
        return;
    }
    DevBuf d_tiles, d_tie;
    struct Rel { DevBuf *a, *b; ~Rel() { a->release(); b->release(); } } rel{&d_tiles, &d_tie};
    uint32_t nt = 0;
    make_tiles(moff, n_reads, d_tiles, &nt);
    d_tie.reserve((size_t)total + 8);
    hipLaunchKernelGGL(k_minimizers_tile, dim3(nt), dim3(SK_TPB), lds, 0, d_h, d_koff, d_moff, d_tiles.as<SkTile>(), w, d_oh, d_op, d_tie.as<uint8_t>());
    hipLaunchKernelGGL(k_minimizer_ties, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_h, d_koff, d_moff, n_reads, total, w, d_tie.as<uint8_t>(), d_op);
    RB_HIP(hipGetLastError());
    RB_HIP(hipDeviceSynchronize());           // d_tiles / d_tie are released on return
}

// all-window hashes of a set of reads; returns device buffer h (caller releases) and host koff
void hash_all(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int mode, std::vector<int64_t> &koff,
              DevBuf &d_koff, DevBuf &d_h) {
    koff.assign((size_t)n_reads + 1, 0);
    for (int64_t i = 0; i < n_reads; ++i) {
        int64_t l = offsets[i + 1] - offsets[i];
        koff[(size_t)i + 1] = koff[(size_t)i] + (l >= k ? l - k + 1 : 0);
    }
    const int64_t total = koff[(size_t)n_reads];
    if (!total) return;
    rb_batch *b = nullptr;
    {
        rb::AsciiUpload up;
        try { rb::ascii_batch_begin(up, device, seq, nullptr, offsets, 0, n_reads, 0, nullptr, true); b = rb::ascii_batch_finish(up); }
        catch (...) { rb::ascii_batch_abort(up); throw; }
    }
    BatchGuard guard{b};
    d_koff.reserve(((size_t)n_reads + 1) * 8);
    d_h.reserve((size_t)total * 8);
    RB_HIP(hipMemcpy(d_koff.p, koff.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_hash_all, dim3(blocks_for(b->n_words)), dim3(TPB), 0, 0, b->codes, b->valid, b->rnz, b->word_read, b->woff, b->len,
                       b->n_words, k, mode, d_koff.as<int64_t>(), d_h.as<uint64_t>());
    RB_HIP(hipGetLastError());
    RB_HIP(hipDeviceSynchronize());
}

}  // namespace

extern "C" {

int rb_minimizers(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int w, int mode, int64_t *moffsets,
                  uint64_t *out_hash, int64_t *out_pos) {
    DevBuf d_koff, d_h, d_moff, d_oh, d_op;
    struct Rel { DevBuf *b[5]; ~Rel() { for (auto x : b) x->release(); } } rel{{&d_koff, &d_h, &d_moff, &d_oh, &d_op}};
    return guarded([&] {
        RB_REQUIRE(offsets && moffsets && n_reads >= 0 && k >= 1 && k <= RB_MAX_K && w >= 1 && mode >= 0 && mode <= 2, "rb_minimizers: bad argument");
        RB_HIP(hipSetDevice(device));
        std::vector<int64_t> koff;
        moffsets[0] = 0;
        for (int64_t i = 0; i < n_reads; ++i) {
            int64_t l = offsets[i + 1] - offsets[i], nk = l >= k ? l - k + 1 : 0;
            moffsets[i + 1] = moffsets[i] + (nk - w + 1 > 0 ? nk - w + 1 : 0);
        }
        const int64_t total = moffsets[n_reads];
        if (!out_hash || !total) return;
        rb::HostPin pin_h(out_hash, (size_t)total * 8), pin_p(out_pos, (size_t)total * 8);      // results are most of the call's time (PCIe)
        hash_all(device, seq, offsets, n_reads, k, mode, koff, d_koff, d_h);
        d_moff.reserve(((size_t)n_reads + 1) * 8); d_oh.reserve((size_t)total * 8); d_op.reserve((size_t)total * 8);
        RB_HIP(hipMemcpy(d_moff.p, moffsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
        launch_minimizers(d_h.as<uint64_t>(), d_koff.as<int64_t>(), d_moff.as<int64_t>(), moffsets, n_reads, total, w, d_oh.as<uint64_t>(), d_op.as<int64_t>());
        RB_HIP(hipMemcpy(out_hash, d_oh.p, (size_t)total * 8, hipMemcpyDeviceToHost));
        if (out_pos) RB_HIP(hipMemcpy(out_pos, d_op.p, (size_t)total * 8, hipMemcpyDeviceToHost));
    });
}

int rb_strobemers(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int n, int wmin, int wmax, int64_t *soffsets,
                  uint64_t *out_hash, int32_t *out_start, int32_t *out_end) {
    DevBuf d_koff, d_h, d_soff, d_oh, d_os, d_oe, d_tiles;
    struct Rel { DevBuf *b[7]; ~Rel() { for (auto x : b) x->release(); } } rel{{&d_koff, &d_h, &d_soff, &d_oh, &d_os, &d_oe, &d_tiles}};
    return guarded([&] {
        RB_REQUIRE(offsets && soffsets && n_reads >= 0 && k >= 1 && k <= RB_MAX_K && n >= 2 && wmin >= 1 && wmax >= wmin, "rb_strobemers: bad argument");
        RB_HIP(hipSetDevice(device));
        std::vector<int64_t> koff;
        soffsets[0] = 0;
        for (int64_t i = 0; i < n_reads; ++i) {
            int64_t l = offsets[i + 1] - offsets[i], nk = l >= k ? l - k + 1 : 0;
            int64_t cnt = (nk > (int64_t)wmax * (n - 1)) ? nk - (int64_t)wmax * (n - 2) - wmin : 0;   // StrobeHashIterator.java:53-57
            soffsets[i + 1] = soffsets[i] + cnt;
        }
        const int64_t total = soffsets[n_reads];
        if (!out_hash || !total) return;
        rb::HostPin pin_h(out_hash, (size_t)total * 8), pin_s(out_start, (size_t)total * 4), pin_e(out_end, (size_t)total * 4);
        hash_all(device, seq, offsets, n_reads, k, 0, koff, d_koff, d_h);
        d_soff.reserve(((size_t)n_reads + 1) * 8); d_oh.reserve((size_t)total * 8); d_os.reserve((size_t)total * 4); d_oe.reserve((size_t)total * 4);
        RB_HIP(hipMemcpy(d_soff.p, soffsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
        const size_t lds = ((size_t)SK_TILE + (size_t)wmax * (size_t)(n - 1)) * 16;
        if (lds > SK_LDS_MAX || n > RB_MAX_STROBES || sk_force_simple())
            hipLaunchKernelGGL(k_strobemers, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_h.as<uint64_t>(), d_koff.as<int64_t>(), d_soff.as<int64_t>(),
                               n_reads, total, k, n, wmin, wmax, d_oh.as<uint64_t>(), d_os.as<int32_t>(), d_oe.as<int32_t>());
        else {
            uint32_t nt = 0;
            make_tiles(soffsets, n_reads, d_tiles, &nt);
            hipLaunchKernelGGL(k_randstrobes_tile, dim3(nt), dim3(SK_TPB), lds, 0, d_h.as<uint64_t>(), (const uint64_t *)nullptr, d_koff.as<int64_t>(),
                               d_soff.as<int64_t>(), d_tiles.as<SkTile>(), n, wmin, wmax, 0, k, d_oh.as<uint64_t>(), (int32_t *)nullptr, d_os.as<int32_t>(), d_oe.as<int32_t>());
        }
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpy(out_hash, d_oh.p, (size_t)total * 8, hipMemcpyDeviceToHost));
        if (out_start) RB_HIP(hipMemcpy(out_start, d_os.p, (size_t)total * 4, hipMemcpyDeviceToHost));
        if (out_end) RB_HIP(hipMemcpy(out_end, d_oe.p, (size_t)total * 4, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"

namespace {
struct Bufs { std::vector<DevBuf *> v; ~Bufs() { for (auto b : v) b->release(); } };
// forward (and reverse) all-window hashes
void hash_fr(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, bool need_r, std::vector<int64_t> &koff,
             DevBuf &d_koff, DevBuf &d_f, DevBuf &d_r) {
    hash_all(device, seq, offsets, n_reads, k, 0, koff, d_koff, d_f);
    if (need_r) { std::vector<int64_t> k2; DevBuf dk2; hash_all(device, seq, offsets, n_reads, k, 2, k2, dk2, d_r); dk2.release(); }
}
void counts_out(rb_graph *count_in, int device, DevBuf &d_oh, int64_t total, float *out_count, DevBuf &d_cnt) {
    if (!count_in || !out_count || !total) return;
    RB_REQUIRE(count_in->p.device == device, "count_in lives on device %d, the reads are hashed on %d", count_in->p.device, device);
    d_cnt.reserve((size_t)total * 4);
    RB_HIP(hipDeviceSynchronize());
    std::shared_lock<std::shared_mutex> lk(count_in->rw);      // a query: may run beside other queries, not beside an insert
    cbf_counts_device(count_in, d_oh.as<uint64_t>(), (size_t)total, d_cnt.as<float>());
    RB_HIP(hipStreamSynchronize(count_in->stream));
    RB_HIP(hipMemcpy(out_count, d_cnt.p, (size_t)total * 4, hipMemcpyDeviceToHost));
}
}  // namespace

extern "C" {

int rb_randstrobes(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int n, int wmin, int wmax, int flags,
                   rb_graph *count_in, int64_t *soffsets, uint64_t *out_hash, int32_t *out_pos, float *out_count) {
    DevBuf d_koff, d_f, d_r, d_soff, d_oh, d_op, d_cnt, d_tiles;
    Bufs rel{{&d_koff, &d_f, &d_r, &d_soff, &d_oh, &d_op, &d_cnt, &d_tiles}};
    return guarded([&] {
        RB_REQUIRE(offsets && soffsets && n_reads >= 0 && k >= 1 && k <= RB_MAX_K && n >= 2 && n <= RB_MAX_STROBES && wmin >= 1 && wmax >= wmin,
                   "rb_randstrobes: bad argument");
        RB_HIP(hipSetDevice(device));
        std::vector<int64_t> koff;
        soffsets[0] = 0;
        for (int64_t i = 0; i < n_reads; ++i) {
            int64_t l = offsets[i + 1] - offsets[i], nk = l >= k ? l - k + 1 : 0;
            soffsets[i + 1] = soffsets[i] + ((nk > (int64_t)wmax * (n - 1)) ? nk - (int64_t)wmax * (n - 2) - wmin : 0);
        }
        const int64_t total = soffsets[n_reads];
        if (!out_hash || !total) return;
        hash_fr(device, seq, offsets, n_reads, k, (flags & RB_STROBE_CANONICAL) != 0, koff, d_koff, d_f, d_r);
        d_soff.reserve(((size_t)n_reads + 1) * 8); d_oh.reserve((size_t)total * 8);
        if (out_pos) d_op.reserve((size_t)total * 4 * (size_t)n);
        RB_HIP(hipMemcpy(d_soff.p, soffsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
        const bool canon = (flags & RB_STROBE_CANONICAL) != 0, slide = (flags & RB_STROBE_SLIDE) && !canon;
        const size_t lds = ((size_t)SK_TILE + (size_t)wmax * (size_t)(n - 1)) * (canon ? 24 : 16);
        if (lds > SK_LDS_MAX || sk_force_simple())
            hipLaunchKernelGGL(k_randstrobes, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_f.as<uint64_t>(), d_r.as<uint64_t>(), d_koff.as<int64_t>(),
                               d_soff.as<int64_t>(), n_reads, total, n, wmin, wmax, flags, d_oh.as<uint64_t>(), out_pos ? d_op.as<int32_t>() : nullptr);
        else {
            uint32_t nt = 0;
            make_tiles(soffsets, n_reads, d_tiles, &nt);
            (void)slide;                   // get()'s sliding across equal k-mer hashes is what `>=` does anyway (see the kernel)
            hipLaunchKernelGGL(k_randstrobes_tile, dim3(nt), dim3(SK_TPB), lds, 0, d_f.as<uint64_t>(), d_r.as<uint64_t>(), d_koff.as<int64_t>(), d_soff.as<int64_t>(),
                               d_tiles.as<SkTile>(), n, wmin, wmax, canon ? 1 : 0, k, d_oh.as<uint64_t>(), out_pos ? d_op.as<int32_t>() : nullptr, (int32_t *)nullptr, (int32_t *)nullptr);
        }
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpy(out_hash, d_oh.p, (size_t)total * 8, hipMemcpyDeviceToHost));
        if (out_pos) RB_HIP(hipMemcpy(out_pos, d_op.p, (size_t)total * 4 * (size_t)n, hipMemcpyDeviceToHost));
        counts_out(count_in, device, d_oh, total, out_count, d_cnt);
    });
}

int rb_strobe3(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int wmin, int wmax, int canonical,
               rb_graph *count_in, int64_t *soffsets, uint64_t *out_hash, int32_t *out_pos, float *out_count) {
    DevBuf d_koff, d_f, d_r, d_soff, d_oh, d_op, d_cnt, d_tiles;
    Bufs rel{{&d_koff, &d_f, &d_r, &d_soff, &d_oh, &d_op, &d_cnt, &d_tiles}};
    return guarded([&] {
        RB_REQUIRE(offsets && soffsets && n_reads >= 0 && k >= 1 && k <= RB_MAX_K && wmin >= 1 && wmax >= wmin, "rb_strobe3: bad argument");
        RB_HIP(hipSetDevice(device));
        std::vector<int64_t> koff;
        soffsets[0] = 0;
        for (int64_t i = 0; i < n_reads; ++i) {
            int64_t l = offsets[i + 1] - offsets[i], nk = l >= k ? l - k + 1 : 0, cnt = 0;
            if (nk > (int64_t)wmin * 2) cnt = canonical ? nk - 2 * (int64_t)wmax : nk - 2 * (int64_t)wmin;   // getMax() + 1 - getMin()
            soffsets[i + 1] = soffsets[i] + (cnt > 0 ? cnt : 0);
        }
        const int64_t total = soffsets[n_reads];
        if (!out_hash || !total) return;
        hash_fr(device, seq, offsets, n_reads, k, canonical != 0, koff, d_koff, d_f, d_r);
        d_soff.reserve(((size_t)n_reads + 1) * 8); d_oh.reserve((size_t)total * 8);
        if (out_pos) d_op.reserve((size_t)total * 12);
        RB_HIP(hipMemcpy(d_soff.p, soffsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
        const size_t lds = ((size_t)SK_TILE + 2 * (size_t)wmax) * (canonical ? 24 : 16);
        if (lds > SK_LDS_MAX || sk_force_simple())
            hipLaunchKernelGGL(k_strobe3, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_f.as<uint64_t>(), d_r.as<uint64_t>(), d_koff.as<int64_t>(),
                               d_soff.as<int64_t>(), n_reads, total, wmin, wmax, canonical, d_oh.as<uint64_t>(), out_pos ? d_op.as<int32_t>() : nullptr);
        else {
            uint32_t nt = 0;
            make_tiles(soffsets, n_reads, d_tiles, &nt);
            hipLaunchKernelGGL(k_strobe3_tile, dim3(nt), dim3(SK_TPB), lds, 0, d_f.as<uint64_t>(), d_r.as<uint64_t>(), d_koff.as<int64_t>(), d_soff.as<int64_t>(),
                               d_tiles.as<SkTile>(), wmin, wmax, canonical ? 1 : 0, d_oh.as<uint64_t>(), out_pos ? d_op.as<int32_t>() : nullptr);
        }
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpy(out_hash, d_oh.p, (size_t)total * 8, hipMemcpyDeviceToHost));
        if (out_pos) RB_HIP(hipMemcpy(out_pos, d_op.p, (size_t)total * 12, hipMemcpyDeviceToHost));
        counts_out(count_in, device, d_oh, total, out_count, d_cnt);
    });
}

int rb_kmer_pair_hashes(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int shift, int canonical,
                        rb_graph *count_in, int64_t *poffsets, uint64_t *out_hash, float *out_count) {
    DevBuf d_koff, d_f, d_r, d_poff, d_oh, d_cnt;
    Bufs rel{{&d_koff, &d_f, &d_r, &d_poff, &d_oh, &d_cnt}};
    return guarded([&] {
        RB_REQUIRE(offsets && poffsets && n_reads >= 0 && k >= 1 && k <= RB_MAX_K && shift >= 1, "rb_kmer_pair_hashes: bad argument");
        RB_HIP(hipSetDevice(device));
        std::vector<int64_t> koff;
        poffsets[0] = 0;
        for (int64_t i = 0; i < n_reads; ++i) {
            int64_t l = offsets[i + 1] - offsets[i], nk = l >= k ? l - k + 1 : 0;
            poffsets[i + 1] = poffsets[i] + (nk > shift ? nk - shift : 0);
        }
        const int64_t total = poffsets[n_reads];
        if (!out_hash || !total) return;
        hash_fr(device, seq, offsets, n_reads, k, canonical != 0, koff, d_koff, d_f, d_r);
        d_poff.reserve(((size_t)n_reads + 1) * 8); d_oh.reserve((size_t)total * 8);
        RB_HIP(hipMemcpy(d_poff.p, poffsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_kmer_pairs, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_f.as<uint64_t>(), d_r.as<uint64_t>(), d_koff.as<int64_t>(),
                           d_poff.as<int64_t>(), n_reads, total, shift, canonical, d_oh.as<uint64_t>());
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpy(out_hash, d_oh.p, (size_t)total * 8, hipMemcpyDeviceToHost));
        counts_out(count_in, device, d_oh, total, out_count, d_cnt);
    });
}

// window minimizers on the device (hashes, positions, per-read window offsets); returns the number of windows
static int64_t window_minimizers(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int w, int mode,
                                 std::vector<int64_t> &woff, std::vector<int64_t> &koff, DevBuf &d_koff, DevBuf &d_h, DevBuf &d_woff, DevBuf &d_mh, DevBuf &d_mp) {
    woff.assign((size_t)n_reads + 1, 0);
    for (int64_t i = 0; i < n_reads; ++i) {
        int64_t l = offsets[i + 1] - offsets[i], nk = l >= k ? l - k + 1 : 0;
        woff[(size_t)i + 1] = woff[(size_t)i] + (nk - w + 1 > 0 ? nk - w + 1 : 0);
    }
    const int64_t total = woff[(size_t)n_reads];
    hash_all(device, seq, offsets, n_reads, k, mode, koff, d_koff, d_h);
    if (!total) return 0;
    d_woff.reserve(((size_t)n_reads + 1) * 8); d_mh.reserve((size_t)total * 8); d_mp.reserve((size_t)total * 8);
    RB_HIP(hipMemcpy(d_woff.p, woff.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
    launch_minimizers(d_h.as<uint64_t>(), d_koff.as<int64_t>(), d_woff.as<int64_t>(), woff.data(), n_reads, total, w, d_mh.as<uint64_t>(), d_mp.as<int64_t>());
    return total;
}

int rb_minimizers_next(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int w, int mode, int64_t *moffsets,
                       uint64_t *out_hash, int64_t *out_pos) {
    DevBuf d_koff, d_h, d_woff, d_mh, d_mp, d_flag, d_slot, d_tmp, d_oh, d_op, d_mo;
    Bufs rel{{&d_koff, &d_h, &d_woff, &d_mh, &d_mp, &d_flag, &d_slot, &d_tmp, &d_oh, &d_op, &d_mo}};
    return guarded([&] {
        RB_REQUIRE(offsets && moffsets && out_hash && n_reads >= 0 && k >= 1 && k <= RB_MAX_K && w >= 1 && mode >= 0 && mode <= 2, "rb_minimizers_next: bad argument");
        RB_HIP(hipSetDevice(device));
        std::vector<int64_t> woff, koff;
        const int64_t total = window_minimizers(device, seq, offsets, n_reads, k, w, mode, woff, koff, d_koff, d_h, d_woff, d_mh, d_mp);
        for (int64_t i = 0; i <= n_reads; ++i) moffsets[i] = 0;
        if (!total) return;
        RB_REQUIRE(total < (int64_t)0xFFFFFFF0, "rb_minimizers_next: too many windows in one call");
        d_flag.reserve((size_t)total * 4 + 4); d_slot.reserve((size_t)total * 4 + 4); d_tmp.reserve(scan_temp_bytes((size_t)total + 1));
        d_oh.reserve((size_t)total * 8); d_op.reserve((size_t)total * 8); d_mo.reserve(((size_t)n_reads + 1) * 8);
        hipLaunchKernelGGL(k_minimizer_flags, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_mp.as<int64_t>(), d_woff.as<int64_t>(), n_reads, total, d_flag.as<uint32_t>());
        RB_HIP(hipMemsetAsync(d_flag.as<uint32_t>() + total, 0, 4, 0));
        exclusive_scan_u32(d_tmp.p, d_tmp.cap, d_flag.as<uint32_t>(), d_slot.as<uint32_t>(), (size_t)total + 1, 0);
        uint32_t n_out = 0;
        RB_HIP(hipMemcpy(&n_out, d_slot.as<uint32_t>() + total, 4, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_minimizer_compact, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_mh.as<uint64_t>(), d_mp.as<int64_t>(), d_flag.as<uint32_t>(),
                           d_slot.as<uint32_t>(), total, d_oh.as<uint64_t>(), d_op.as<int64_t>());
        hipLaunchKernelGGL(k_gather_u32, dim3(blocks_for(n_reads + 1)), dim3(TPB), 0, 0, d_slot.as<uint32_t>(), d_woff.as<int64_t>(), n_reads, total, n_out,
                           d_mo.as<int64_t>());
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpy(moffsets, d_mo.p, ((size_t)n_reads + 1) * 8, hipMemcpyDeviceToHost));
        RB_HIP(hipMemcpy(out_hash, d_oh.p, (size_t)n_out * 8, hipMemcpyDeviceToHost));
        if (out_pos) RB_HIP(hipMemcpy(out_pos, d_op.p, (size_t)n_out * 8, hipMemcpyDeviceToHost));
    });
}

int rb_minimizer_set(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int w, int mode, const uint64_t *stale,
                     int64_t *moffsets, uint64_t *out) {
    DevBuf d_koff, d_h, d_woff, d_mh, d_mp, d_flag, d_slot, d_tmp, d_v0, d_v1, d_r0, d_r1, d_eoff, d_kread, d_stale, d_per, d_out;
    Bufs rel{{&d_koff, &d_h, &d_woff, &d_mh, &d_mp, &d_flag, &d_slot, &d_tmp, &d_v0, &d_v1, &d_r0, &d_r1, &d_eoff, &d_kread, &d_stale, &d_per, &d_out}};
    return guarded([&] {
        RB_REQUIRE(offsets && moffsets && out && n_reads >= 0 && k >= 1 && k <= RB_MAX_K && w >= 1 && mode >= 0 && mode <= 2, "rb_minimizer_set: bad argument");
        RB_HIP(hipSetDevice(device));
        std::vector<int64_t> woff, koff;
        const int64_t total_w = window_minimizers(device, seq, offsets, n_reads, k, w, mode, woff, koff, d_koff, d_h, d_woff, d_mh, d_mp);
        // entries to sort: every window minimizer of the long reads + one value per short read (numKmers <= windowSize)
        std::vector<int64_t> eoff((size_t)n_reads + 1, 0), kread;
        for (int64_t i = 0; i < n_reads; ++i) {
            const int64_t nw = woff[(size_t)i + 1] - woff[(size_t)i], l = offsets[i + 1] - offsets[i];
            const bool is_short = l - k + 1 <= w;                      // numKmers <= windowSize (numKmers may be <= 0)
            if (is_short) kread.push_back(i);
            eoff[(size_t)i + 1] = eoff[(size_t)i] + (is_short ? 1 : nw);
        }
        const int64_t total = eoff[(size_t)n_reads], n_short = (int64_t)kread.size();
        for (int64_t i = 0; i <= n_reads; ++i) moffsets[i] = 0;
        if (!total) return;
        RB_REQUIRE(total < (int64_t)0xFFFFFFF0, "rb_minimizer_set: too many windows in one call");
        d_v0.reserve((size_t)total * 8); d_v1.reserve((size_t)total * 8); d_r0.reserve((size_t)total * 8); d_r1.reserve((size_t)total * 8);
        d_eoff.reserve(((size_t)n_reads + 1) * 8);
        RB_HIP(hipMemcpy(d_eoff.p, eoff.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
        // long reads: their window minimizers are consecutive in d_mh (window order = read order) and in the entry array
        // (short reads have no windows), so copy read by read ranges: one device copy per maximal stretch of long reads
        {
            int64_t i = 0;
            while (i < n_reads) {
                if (woff[(size_t)i + 1] == woff[(size_t)i]) { ++i; continue; }
                int64_t j = i;
                while (j < n_reads && (offsets[j + 1] - offsets[j]) - k + 1 > w) ++j;
                const int64_t cnt = woff[(size_t)j] - woff[(size_t)i];
                if (cnt) RB_HIP(hipMemcpyAsync(d_v0.as<uint64_t>() + eoff[(size_t)i], d_mh.as<uint64_t>() + woff[(size_t)i], (size_t)cnt * 8, hipMemcpyDeviceToDevice, 0));
                i = j > i ? j : i + 1;
            }
        }
        if (n_short) {
            d_kread.reserve((size_t)n_short * 8); d_per.reserve((size_t)n_short * 8);
            RB_HIP(hipMemcpy(d_kread.p, kread.data(), (size_t)n_short * 8, hipMemcpyHostToDevice));
            const uint64_t *d_st = nullptr;
            if (stale) { d_stale.reserve((size_t)n_reads * 8); RB_HIP(hipMemcpy(d_stale.p, stale, (size_t)n_reads * 8, hipMemcpyHostToDevice)); d_st = d_stale.as<uint64_t>(); }
            hipLaunchKernelGGL(k_short_read_min, dim3(blocks_for(n_short)), dim3(TPB), 0, 0, d_h.as<uint64_t>(), d_koff.as<int64_t>(), d_kread.as<int64_t>(), n_short,
                               d_st, d_per.as<uint64_t>());
            std::vector<int64_t> where((size_t)n_short);
            for (int64_t q = 0; q < n_short; ++q) where[(size_t)q] = eoff[(size_t)kread[(size_t)q]];
            RB_HIP(hipMemcpy(d_kread.p, where.data(), (size_t)n_short * 8, hipMemcpyHostToDevice));   // k_short_read_min has run: stream order
            hipLaunchKernelGGL(k_scatter_u64, dim3(blocks_for(n_short)), dim3(TPB), 0, 0, d_v0.as<uint64_t>(), d_kread.as<int64_t>(), d_per.as<uint64_t>(), n_short);
        }
        // sort by (read, signed value): stable radix on the value (sign bit flipped), then on the read index
        hipLaunchKernelGGL(k_set_keys, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_v0.as<uint64_t>(), d_eoff.as<int64_t>(), n_reads, total, d_r0.as<uint64_t>());
        hipLaunchKernelGGL(k_flip_sign, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_v0.as<uint64_t>(), total);
        d_tmp.reserve(std::max(sort_pairs32_temp_bytes((size_t)total), scan_temp_bytes((size_t)total + 1)));
        sort_pairs_u64_u64(d_tmp.p, d_tmp.cap, d_v0.as<uint64_t>(), d_v1.as<uint64_t>(), d_r0.as<uint64_t>(), d_r1.as<uint64_t>(), (size_t)total, 0, 64, 0);
        int rbits = 1; while (((int64_t)1 << rbits) < n_reads + 1) ++rbits;
        sort_pairs_u64_u64(d_tmp.p, d_tmp.cap, d_r1.as<uint64_t>(), d_r0.as<uint64_t>(), d_v1.as<uint64_t>(), d_v0.as<uint64_t>(), (size_t)total, 0, rbits, 0);
        // distinct (read, value) pairs
        d_flag.reserve((size_t)total * 4 + 4); d_slot.reserve((size_t)total * 4 + 4);
        hipLaunchKernelGGL(k_unique_flags, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_r0.as<uint64_t>(), d_v0.as<uint64_t>(), total, d_flag.as<uint32_t>());
        RB_HIP(hipMemsetAsync(d_flag.as<uint32_t>() + total, 0, 4, 0));
        exclusive_scan_u32(d_tmp.p, d_tmp.cap, d_flag.as<uint32_t>(), d_slot.as<uint32_t>(), (size_t)total + 1, 0);
        uint32_t n_out = 0;
        RB_HIP(hipMemcpy(&n_out, d_slot.as<uint32_t>() + total, 4, hipMemcpyDeviceToHost));
        d_out.reserve((size_t)n_out * 8 + 8);
        d_mp.reserve(((size_t)n_reads + 1) * 8);                       // reuse: per-read counts
        RB_HIP(hipMemset(d_mp.p, 0, ((size_t)n_reads + 1) * 8));
        hipLaunchKernelGGL(k_unique_compact, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_r0.as<uint64_t>(), d_v0.as<uint64_t>(), d_flag.as<uint32_t>(),
                           d_slot.as<uint32_t>(), total, d_out.as<uint64_t>(), reinterpret_cast<unsigned long long *>(d_mp.p));
        RB_HIP(hipGetLastError());
        std::vector<unsigned long long> per((size_t)n_reads + 1);
        RB_HIP(hipMemcpy(per.data(), d_mp.p, ((size_t)n_reads + 1) * 8, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n_reads; ++i) moffsets[i + 1] = moffsets[i] + (int64_t)per[(size_t)i];
        RB_HIP(hipMemcpy(out, d_out.p, (size_t)n_out * 8, hipMemcpyDeviceToHost));
        (void)total_w;
    });
}

}  // extern "C"
