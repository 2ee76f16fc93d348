// rb_sketch.hip — minimizer and strobemer extraction over long reads (BASELINE config 5).
// Hash-only work: one read is independent of every other, so multi-GPU = replicas over read shards.
#include "rb_pipeline.hpp"

using namespace rb;

namespace {

// ntHash of EVERY window of a read (no segmentation): unusable bases contribute seed 0, exactly like
// the zero rows of seedTab (R/bloom/hash/NTHash.java:133-166).  mode 0 forward, 1 canonical, 2 RC.
__global__ void k_hash_all(const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid,
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
    uint64_t f = 0, rv = 0;
    uint32_t filled = 0;
    for (uint32_t b = b0; b < bend; ++b) {
        const bool ok = (vw[b >> 5] >> (b & 31u)) & 1u;
        const uint32_t c = (uint32_t)(cw[b >> 5] >> (2u * (b & 31u))) & 3u;
        const uint64_t s_in = ok ? seed_of(c) : 0ull, sc_in = ok ? seed_of(3u - c) : 0ull;
        if (filled < uk) { f = rotl(f, 1) ^ s_in; rv ^= rotl(sc_in, filled); ++filled; }
        else {
            const uint32_t bo = b - uk;
            const bool oko = (vw[bo >> 5] >> (bo & 31u)) & 1u;
            const uint32_t oc = (uint32_t)(cw[bo >> 5] >> (2u * (bo & 31u))) & 3u;
            const uint64_t s_out = oko ? seed_of(oc) : 0ull, sc_out = oko ? seed_of(3u - oc) : 0ull;
            f = rotl(f, 1) ^ rotl(s_out, uk) ^ s_in;
            rv = rotr(rv, 1) ^ rotr(sc_out, 1) ^ rotl(sc_in, uk - 1u);
        }
        if (filled >= uk) out[koff[r] + (b - uk + 1u)] = mode == 0 ? f : mode == 2 ? rv : canonical(f, rv);
    }
}

// one thread per minimizer window: signed minimum over w consecutive k-mer hashes, leftmost on ties
__global__ void k_minimizers(const uint64_t *__restrict__ h, const int64_t *__restrict__ koff, const int64_t *__restrict__ moff,
                             int64_t n_reads, int64_t total, int w, uint64_t *__restrict__ out_hash, int64_t *__restrict__ out_pos) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int64_t lo = 0, hi = n_reads;      // read r with moff[r] <= t < moff[r+1]
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (moff[mid] <= t) lo = mid; else hi = mid; }
    const int64_t p = t - moff[lo];
    const uint64_t *hh = h + koff[lo] + p;
    int64_t best = (int64_t)hh[0], bi = 0;
    for (int i = 1; i < w; ++i) { int64_t v = (int64_t)hh[i]; if (v < best) { best = v; bi = i; } }
    out_hash[t] = (uint64_t)best;
    out_pos[t] = p + bi;
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

struct BatchGuard { rb_batch *b; ~BatchGuard() { if (b) rb_batch_destroy(b); } };

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
    int rc = rb_batch_create_ascii(device, seq, nullptr, offsets, n_reads, 0, &b);
    if (rc != RB_OK) throw HipError{rc};
    BatchGuard guard{b};
    d_koff.reserve(((size_t)n_reads + 1) * 8);
    d_h.reserve((size_t)total * 8);
    RB_HIP(hipMemcpy(d_koff.p, koff.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_hash_all, dim3(blocks_for(b->n_words)), dim3(TPB), 0, 0, b->codes, b->valid, b->word_read, b->woff, b->len,
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
        hash_all(device, seq, offsets, n_reads, k, mode, koff, d_koff, d_h);
        d_moff.reserve(((size_t)n_reads + 1) * 8); d_oh.reserve((size_t)total * 8); d_op.reserve((size_t)total * 8);
        RB_HIP(hipMemcpy(d_moff.p, moffsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_minimizers, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_h.as<uint64_t>(), d_koff.as<int64_t>(), d_moff.as<int64_t>(),
                           n_reads, total, w, d_oh.as<uint64_t>(), d_op.as<int64_t>());
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpy(out_hash, d_oh.p, (size_t)total * 8, hipMemcpyDeviceToHost));
        if (out_pos) RB_HIP(hipMemcpy(out_pos, d_op.p, (size_t)total * 8, hipMemcpyDeviceToHost));
    });
}

int rb_strobemers(int device, const char *seq, const int64_t *offsets, int64_t n_reads, int k, int n, int wmin, int wmax, int64_t *soffsets,
                  uint64_t *out_hash, int32_t *out_start, int32_t *out_end) {
    DevBuf d_koff, d_h, d_soff, d_oh, d_os, d_oe;
    struct Rel { DevBuf *b[6]; ~Rel() { for (auto x : b) x->release(); } } rel{{&d_koff, &d_h, &d_soff, &d_oh, &d_os, &d_oe}};
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
        hash_all(device, seq, offsets, n_reads, k, 0, koff, d_koff, d_h);
        d_soff.reserve(((size_t)n_reads + 1) * 8); d_oh.reserve((size_t)total * 8); d_os.reserve((size_t)total * 4); d_oe.reserve((size_t)total * 4);
        RB_HIP(hipMemcpy(d_soff.p, soffsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_strobemers, dim3(blocks_for(total)), dim3(TPB), 0, 0, d_h.as<uint64_t>(), d_koff.as<int64_t>(), d_soff.as<int64_t>(),
                           n_reads, total, k, n, wmin, wmax, d_oh.as<uint64_t>(), d_os.as<int32_t>(), d_oe.as<int32_t>());
        RB_HIP(hipGetLastError());
        RB_HIP(hipMemcpy(out_hash, d_oh.p, (size_t)total * 8, hipMemcpyDeviceToHost));
        if (out_start) RB_HIP(hipMemcpy(out_start, d_os.p, (size_t)total * 4, hipMemcpyDeviceToHost));
        if (out_end) RB_HIP(hipMemcpy(out_end, d_oe.p, (size_t)total * 4, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
