// rb_io.hip — host-side text ingest: FASTQ records -> the concatenated sequence / quality buffers + offsets that
// rb_graph_add_reads / rb_batch_create_ascii take (the 2-bit encoding and the quality segmentation happen on the GPU).
// Reference: FastqReader.nextWithoutName R/io/FastqReader.java:140-186 — four lines per record, line 1 starts with '@',
// line 3 with '+', a record cut off by the end of the input is dropped (NoSuchElementException -> null);
// BufferedReader.lines() ends a line at \n, \r\n or \r.  The reference reads under one lock (:144-149), which is what
// stops its stage 1 from scaling; here line starts are counted per chunk in parallel, and since a record is exactly four
// lines the global line number of every chunk start tells which field a line is.
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "rb_pipeline.hpp"
#include "rb_kernels.hpp"

using namespace rb;

namespace {
struct Chunk { size_t b, e; int64_t lines_before = 0; };

// a line START is position 0 and every position after an end of line (\n, or \r not followed by \n)
inline bool eol_at(const char *t, size_t n, size_t i) { return t[i] == '\n' || (t[i] == '\r' && !(i + 1 < n && t[i + 1] == '\n')); }
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// The same record structure found on the GPU: the text is uploaded as it is (one PCIe crossing of the file's bytes instead
// of a host pass over them plus two buffers), ends of line are counted per 8 KB tile, one scan turns the counts into line
// numbers, every end of line writes the start of the line after it, a record is four consecutive entries of that table, and
// the sequence line is 2-bit encoded straight from the text (the quality line is only looked at for the usable-base mask).
namespace {
constexpr uint32_t FQ_TILE = 8192, FQ_TPB = 256, FQ_PER = FQ_TILE / FQ_TPB;     // 32 bytes per thread

// bit b of the result: text[i0 + b] ends a line (\n, or \r not followed by \n); bytes at or beyond n never do
__device__ __forceinline__ uint32_t fq_eol_mask(const uint8_t *__restrict__ t, uint32_t i0, uint32_t n) {
    if (i0 >= n) return 0u;
    const uint4 a = *reinterpret_cast<const uint4 *>(t + i0), b = *reinterpret_cast<const uint4 *>(t + i0 + 16);   // padded allocation
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const uint32_t nxt = t[i0 + 32];
    uint32_t m = 0;
#pragma unroll
    for (uint32_t q = 0; q < 32u; ++q) {
        const uint32_t c = (w[q >> 2] >> (8u * (q & 3u))) & 0xFFu;
        const uint32_t d = q < 31u ? (w[(q + 1u) >> 2] >> (8u * ((q + 1u) & 3u))) & 0xFFu : nxt;
        const bool eol = c == '\n' || (c == '\r' && d != '\n');
        m |= (eol && i0 + q < n) ? 1u << q : 0u;
    }
    return m;
}
__global__ void __launch_bounds__(FQ_TPB) k_fq_eol_count(const uint8_t *__restrict__ t, uint32_t n, uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t s_w[FQ_TPB / 64];
    const uint32_t i0 = blockIdx.x * FQ_TILE + threadIdx.x * FQ_PER;
    uint32_t c = (uint32_t)__popc(fq_eol_mask(t, i0, n));
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t s = 0; for (uint32_t q = 0; q < FQ_TPB / 64; ++q) s += s_w[q]; tile_cnt[blockIdx.x] = s; }
}
// ls[0] = 0, ls[i] = start of line i = position after the i-th end of line
__global__ void __launch_bounds__(FQ_TPB) k_fq_line_starts(const uint8_t *__restrict__ t, uint32_t n, const uint32_t *__restrict__ tile_base,
                                                            uint32_t *__restrict__ ls) {
    __shared__ uint32_t s_w[FQ_TPB / 64];
    const uint32_t i0 = blockIdx.x * FQ_TILE + threadIdx.x * FQ_PER, lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t m = fq_eol_mask(t, i0, n);
    const uint32_t mine = (uint32_t)__popc(m);
    uint32_t inc = mine;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(inc, o, 64); if ((int)lane >= o) inc += x; }
    if (lane == 63u) s_w[w] = inc;
    __syncthreads();
    uint32_t base = tile_base[blockIdx.x] + inc - mine;
    for (uint32_t q = 0; q < w; ++q) base += s_w[q];
    if (blockIdx.x == 0 && threadIdx.x == 0) ls[0] = 0u;
    while (m) {
        const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
        m &= m - 1u;
        ls[++base] = i0 + b + 1u;
    }
}
// err[0]: a record whose line 1 does not start with '@'; err[1]: line 3 without '+'; err[2]: smallest record with different
// numbers of bases and qualities; err[3]: longest read; err[4..5]: min / max words per read; bases: total
__global__ void k_fq_records(const uint8_t *__restrict__ t, uint32_t n, const uint32_t *__restrict__ ls, uint32_t n_records, int use_qual,
                             uint32_t *__restrict__ seq_pos, uint32_t *__restrict__ qual_pos, uint32_t *__restrict__ len,
                             uint32_t *__restrict__ nwords, uint32_t *__restrict__ err, unsigned long long *__restrict__ bases) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t L = 0;
    if (r < n_records) {
        uint32_t s[4], e[4];
        for (int f = 0; f < 4; ++f) {
            s[f] = ls[4u * r + f];
            const uint32_t j = ls[4u * r + f + 1u] - 1u;            // the end of line that closes it (n for an open last line)
            uint32_t ce = j;
            if (j < n && t[j] == '\n' && ce > s[f] && t[ce - 1u] == '\r') --ce;   // \r\n
            e[f] = ce;
        }
        if (e[0] == s[0] || t[s[0]] != '@') err[0] = 1u;
        if (e[2] == s[2] || t[s[2]] != '+') err[1] = 1u;
        L = e[1] - s[1];
        if (use_qual && e[3] - s[3] != L) atomicMin(&err[2], r);
        seq_pos[r] = s[1]; qual_pos[r] = s[3]; len[r] = L;
        const uint32_t nwd = (L + 31u) >> 5;
        nwords[r] = nwd;
        atomicMax(&err[3], L); atomicMin(&err[4], nwd); atomicMax(&err[5], nwd);
    }
    unsigned long long tot = L;
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o, 64);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(bases, tot);
}
// k_encode_ascii (rb_batch.hip) with the bases and qualities where the text has them
__global__ void k_fq_encode(const uint8_t *__restrict__ t, const uint32_t *__restrict__ seq_pos, const uint32_t *__restrict__ qual_pos,
                            const uint32_t *__restrict__ len, const uint32_t *__restrict__ woff, int64_t n_reads, int64_t n_words, int use_qual,
                            int min_q, uint64_t *__restrict__ codes, uint32_t *__restrict__ valid, uint32_t *__restrict__ word_read) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    int64_t lo = 0, hi = n_reads;                                   // owning read: largest r with woff[r] <= w
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (woff[mid] <= (uint32_t)w) lo = mid; else hi = mid; }
    const uint32_t r = (uint32_t)lo, L = len[r], b0 = (uint32_t)(w - woff[r]) * 32u;
    const uint8_t *sq = t + seq_pos[r], *ql = t + qual_pos[r];
    uint64_t c = 0;
    uint32_t v = 0;
    for (uint32_t i = 0; i < 32u && b0 + i < L; ++i) {
        uint32_t code = 4;
        switch (sq[b0 + i]) {                                        // [ACGTU], CASE_INSENSITIVE (R/util/SeqUtils.java:1436-1438)
            case 'A': case 'a': code = 0; break;
            case 'C': case 'c': code = 1; break;
            case 'G': case 'g': code = 2; break;
            case 'T': case 't': case 'U': case 'u': code = 3; break;
            default: break;
        }
        bool ok = code < 4u;
        if (use_qual) { const uint32_t q = ql[b0 + i]; ok = ok && q >= (uint32_t)(33 + min_q) && q <= (uint32_t)'~'; }   // SeqUtils.java:1426-1434
        if (ok) { c |= (uint64_t)code << (2u * i); v |= 1u << i; }
    }
    codes[w] = c; valid[w] = v; word_read[w] = r;
}
// ---- FASTA (R/io/FastaReader.java:70-104, next()): lines are trimmed (String.trim: characters <= ' ' at both ends); a line that
// starts with '>' opens a record, the lines after it are its sequence, joined; an empty line closes the record and the line
// after it must be a header again ("Incorrect FASTA header format" otherwise) — or empty, which ENDS the iteration (next()
// returns null: nothing behind it is read); a header that is the file's last line is never returned.
// kind: 0 = empty, 1 = header, 2 = sequence
__global__ void k_fa_lines(const uint8_t *__restrict__ t, uint32_t n, const uint32_t *__restrict__ ls, uint32_t n_lines, uint32_t *__restrict__ ts,
                           uint32_t *__restrict__ tl, uint32_t *__restrict__ is_h, uint8_t *__restrict__ kind) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines) return;
    uint32_t s = ls[i], j = ls[i + 1u] - 1u;                     // [s, j): the line without its end of line (j = n for an open last line)
    j = j < n ? j : n;
    while (s < j && t[s] <= 0x20u) ++s;
    while (j > s && t[j - 1u] <= 0x20u) --j;
    const uint32_t kd = s == j ? 0u : t[s] == '>' ? 1u : 2u;
    ts[i] = s; tl[i] = kd == 2u ? j - s : 0u; is_h[i] = kd == 1u; kind[i] = (uint8_t)kd;
}
// res[0] = first sequence line in header position (an error unless the iteration ended before), res[1] = first empty line in
// header position (the end of the iteration); header position = line 0 or the line after an empty line
__global__ void k_fa_check(const uint8_t *__restrict__ kind, uint32_t n_lines, uint32_t *__restrict__ res) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines) return;
    const bool header_pos = i == 0u || kind[i - 1u] == 0u;
    if (!header_pos) return;
    if (kind[i] == 2u) atomicMin(&res[0], i);
    else if (kind[i] == 0u) atomicMin(&res[1], i);
}
__global__ void k_fa_header_lines(const uint32_t *__restrict__ is_h, const uint32_t *__restrict__ hidx, uint32_t limit, uint32_t *__restrict__ hl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < limit && is_h[i]) hl[hidx[i]] = i;
}
// per record: sequence length = sequence bytes between its header line and the next one (cum = exclusive scan of the trimmed
// lengths of the sequence lines), words, the extremes (err[3..5] as in k_fq_records), total bases
__global__ void k_fa_records(const uint32_t *__restrict__ hl, const uint32_t *__restrict__ cum, uint32_t n_records, uint32_t *__restrict__ len,
                             uint32_t *__restrict__ nwords, uint32_t *__restrict__ err, unsigned long long *__restrict__ bases) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t L = 0;
    if (r < n_records) {
        L = cum[hl[r + 1u]] - cum[hl[r]];
        len[r] = L;
        const uint32_t nwd = (L + 31u) >> 5;
        nwords[r] = nwd;
        atomicMax(&err[3], L); atomicMin(&err[4], nwd); atomicMax(&err[5], nwd);
    }
    unsigned long long tot = L;
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o, 64);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(bases, tot);
}
// 2-bit encode straight from the text: the word's first base is byte g0 of the record's joined sequence; the line that holds it is
// the last one with cum <= g0 among the record's lines; from there the bytes are walked across line ends
__global__ void k_fa_encode(const uint8_t *__restrict__ t, const uint32_t *__restrict__ hl, const uint32_t *__restrict__ cum, const uint32_t *__restrict__ ts,
                            const uint32_t *__restrict__ tl, const uint32_t *__restrict__ len, const uint32_t *__restrict__ woff, int64_t n_reads,
                            int64_t n_words, uint64_t *__restrict__ codes, uint32_t *__restrict__ valid, uint32_t *__restrict__ word_read) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    int64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (woff[mid] <= (uint32_t)w) lo = mid; else hi = mid; }
    const uint32_t r = (uint32_t)lo, L = len[r], b0 = (uint32_t)(w - woff[r]) * 32u;
    const uint32_t g0 = cum[hl[r]] + b0;
    uint32_t a = hl[r] + 1u, e = hl[r + 1u];                      // the record's lines: (header, next header)
    while (e - a > 1u) { const uint32_t mid = (a + e) >> 1; if (cum[mid] <= g0) a = mid; else e = mid; }
    uint32_t line = a, off = g0 - cum[line];
    uint64_t c = 0;
    uint32_t v = 0;
    for (uint32_t i = 0; i < 32u && b0 + i < L; ++i) {
        while (off >= tl[line]) { off -= tl[line]; ++line; }      // (also skips lines that contribute nothing)
        uint32_t code = 4;
        switch (t[ts[line] + off]) {                              // [ACGTU], CASE_INSENSITIVE (R/util/SeqUtils.java:1436-1438)
            case 'A': case 'a': code = 0; break;
            case 'C': case 'c': code = 1; break;
            case 'G': case 'g': code = 2; break;
            case 'T': case 't': case 'U': case 'u': code = 3; break;
            default: break;
        }
        if (code < 4u) { c |= (uint64_t)code << (2u * i); v |= 1u << i; }
        ++off;
    }
    codes[w] = c; valid[w] = v; word_read[w] = r;
}
struct TmpBuf {                                                      // given back on every way out
    void *p = nullptr;
    DevPool *pool = nullptr;           // pieces of a file ingest: the blocks of piece c serve piece c + 1 (no hipMalloc / hipFree — which waits for the device — per piece)
    ~TmpBuf() { if (!p) return; if (pool) pool->put(p); else (void)hipFree(p); }
    template <class T> T *alloc(size_t n) {
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        if (pool) p = pool->get(bytes); else RB_HIP(hipMalloc(&p, bytes));
        return static_cast<T *>(p);
    }
};
// the arrays of a batch made from text: out of the pool if there is one (rb_batch_destroy hands them back)
void alloc_text_batch(rb_batch *b, uint32_t R, DevPool *pool) {
    const size_t nw = (size_t)std::max<int64_t>(b->n_words, 1), nr = (size_t)std::max<uint32_t>(R, 1u);
    b->pool = pool;
    if (pool) {
        b->codes = static_cast<uint64_t *>(pool->get(nw * 8)); b->valid = static_cast<uint32_t *>(pool->get(nw * 4)); b->word_read = static_cast<uint32_t *>(pool->get(nw * 4));
        b->woff = static_cast<uint32_t *>(pool->get(((size_t)R + 2) * 4)); b->len = static_cast<uint32_t *>(pool->get(nr * 4));
    } else {
        RB_HIP(hipMalloc(&b->codes, nw * 8));
        RB_HIP(hipMalloc(&b->valid, nw * 4));
        RB_HIP(hipMalloc(&b->word_read, nw * 4));
        RB_HIP(hipMalloc(&b->woff, ((size_t)R + 2) * 4));
        RB_HIP(hipMalloc(&b->len, nr * 4));
    }
    b->device_bytes = nw * 16 + ((size_t)R + 2) * 4 + nr * 4;
}
}  // namespace

namespace rb {
FastqChunk fastq_batch_create(int device, const char *text, size_t n, bool final, int min_base_qual, bool use_qual, hipStream_t st, DevPool *pool) {
    RB_REQUIRE(text || n == 0, "rb_batch_create_fastq: null text");
    RB_REQUIRE(n < 0xFFFFFF00ull, "rb_batch_create_fastq: at most 4 GiB of text per call (got %zu bytes)", n);
    RB_REQUIRE(min_base_qual >= 0 && min_base_qual < 94, "rb_batch_create_fastq: min_base_qual out of range");
    RB_HIP(hipSetDevice(device));
    if (!final && n && text[n - 1] == '\r') --n;                     // its \n may be the first byte of the next piece
    FastqChunk out;
    rb_batch *b = new rb_batch();
    struct Guard { rb_batch *b; ~Guard() { if (b) rb_batch_destroy(b); } } guard{b};
    b->device = device;
    const uint32_t un = (uint32_t)n, ntiles = (uint32_t)(((uint64_t)n + FQ_TILE - 1) / FQ_TILE);   // 64-bit: n + FQ_TILE - 1 wraps in 32 bits just below 4 GiB
    TmpBuf d_text, d_cnt, d_base, d_tmp, d_ls, d_sp, d_qp, d_len, d_nw, d_woff, d_err;
    for (TmpBuf *q : {&d_text, &d_cnt, &d_base, &d_tmp, &d_ls, &d_sp, &d_qp, &d_len, &d_nw, &d_woff, &d_err}) q->pool = pool;
    uint8_t *t = d_text.alloc<uint8_t>((size_t)ntiles * FQ_TILE + 64);
    RB_HIP(hipMemsetAsync(t + n, 0, (size_t)ntiles * FQ_TILE + 64 - n, st));
    if (n) RB_HIP(SlabPin::copy(t, text, n, st));            // (split at the boundaries a slab-wise registration of the text would have: SlabPin)
    uint32_t *cnt = d_cnt.alloc<uint32_t>(ntiles + 1), *base = d_base.alloc<uint32_t>(ntiles + 1);
    void *tmp = d_tmp.alloc<uint8_t>(scan_temp_bytes((size_t)ntiles + 1));
    RB_HIP(hipMemsetAsync(cnt + ntiles, 0, 4, st));
    if (ntiles) hipLaunchKernelGGL(k_fq_eol_count, dim3(ntiles), dim3(FQ_TPB), 0, st, t, un, cnt);
    exclusive_scan_u32(tmp, scan_temp_bytes((size_t)ntiles + 1), cnt, base, (size_t)ntiles + 1, st);
    uint32_t eols = 0;
    RB_HIP(hipMemcpyAsync(&eols, base + ntiles, 4, hipMemcpyDeviceToHost, st));
    RB_HIP(hipStreamSynchronize(st));
    const bool open_tail = final && n > 0 && !(text[n - 1] == '\n' || text[n - 1] == '\r');
    const uint64_t n_lines = (uint64_t)eols + (open_tail ? 1u : 0u);
    const uint32_t R = (uint32_t)(n_lines / 4u);
    uint32_t *ls = d_ls.alloc<uint32_t>((size_t)eols + 2);
    if (ntiles) hipLaunchKernelGGL(k_fq_line_starts, dim3(ntiles), dim3(FQ_TPB), 0, st, t, un, base, ls);
    else RB_HIP(hipMemsetAsync(ls, 0, 4, st));
    const uint32_t sentinel = un + 1u;                                // "end of line" of an open last line = n
    RB_HIP(hipMemcpyAsync(ls + eols + 1, &sentinel, 4, hipMemcpyHostToDevice, st));
    uint32_t *sp = d_sp.alloc<uint32_t>(R), *qp = d_qp.alloc<uint32_t>(R), *ln = d_len.alloc<uint32_t>(R), *nw = d_nw.alloc<uint32_t>((size_t)R + 1),
             *woff = d_woff.alloc<uint32_t>((size_t)R + 1);
    uint32_t *err = d_err.alloc<uint32_t>(16);
    const uint32_t err0[8] = {0u, 0u, ~0u, 0u, ~0u, 0u, 0u, 0u};
    RB_HIP(hipMemcpyAsync(err, err0, sizeof err0, hipMemcpyHostToDevice, st));
    RB_HIP(hipMemsetAsync(nw + R, 0, 4, st));
    if (R) hipLaunchKernelGGL(k_fq_records, dim3((R + 255u) / 256u), dim3(256), 0, st, t, un, ls, R, use_qual ? 1 : 0, sp, qp, ln, nw, err,
                              reinterpret_cast<unsigned long long *>(err + 6));
    TmpBuf d_tmp2;
    for (TmpBuf *q : {&d_tmp2}) q->pool = pool;
    void *tmp2 = d_tmp2.alloc<uint8_t>(scan_temp_bytes((size_t)R + 1));
    exclusive_scan_u32(tmp2, scan_temp_bytes((size_t)R + 1), nw, woff, (size_t)R + 1, st);
    uint32_t herr[8], consumed32 = 0;
    b->h_woff.assign((size_t)R + 1, 0u);
    RB_HIP(hipMemcpyAsync(herr, err, sizeof herr, hipMemcpyDeviceToHost, st));
    RB_HIP(hipMemcpyAsync(b->h_woff.data(), woff, ((size_t)R + 1) * 4, hipMemcpyDeviceToHost, st));
    if ((uint64_t)4u * R <= eols) RB_HIP(hipMemcpyAsync(&consumed32, ls + (size_t)4u * R, 4, hipMemcpyDeviceToHost, st));
    RB_HIP(hipStreamSynchronize(st));
    RB_REQUIRE(!herr[0], "rb_batch_create_fastq: Line 1 of FASTQ record is expected to start with '@'");
    RB_REQUIRE(!herr[1], "rb_batch_create_fastq: Line 3 of FASTQ record is expected to start with '+'");
    RB_REQUIRE(herr[2] == ~0u, "rb_batch_create_fastq: record %u has different numbers of bases and qualities", herr[2]);
    out.consumed = (uint64_t)4u * R <= eols ? (size_t)consumed32 : n;
    out.records = R;
    b->n_reads = R;
    b->n_words = b->h_woff[R];
    b->max_len = herr[3];
    b->wpr_uniform = (R && herr[4] == herr[5]) ? herr[4] : 0u;
    unsigned long long nb; memcpy(&nb, herr + 6, 8);
    b->n_bases = (int64_t)nb;
    alloc_text_batch(b, R, pool);
    RB_HIP(hipMemcpyAsync(b->woff, woff, ((size_t)R + 1) * 4, hipMemcpyDeviceToDevice, st));
    if (R) RB_HIP(hipMemcpyAsync(b->len, ln, (size_t)R * 4, hipMemcpyDeviceToDevice, st));
    if (b->n_words)
        hipLaunchKernelGGL(k_fq_encode, dim3(blocks_for(b->n_words)), dim3(TPB), 0, st, t, sp, qp, ln, b->woff, (int64_t)R, b->n_words, use_qual ? 1 : 0,
                           min_base_qual, b->codes, b->valid, b->word_read);
    RB_HIP(hipGetLastError());
    RB_HIP(hipStreamSynchronize(st));
    out.b = b; guard.b = nullptr;
    return out;
}
// FASTA text -> packed batch on the GPU.  final = false: the text is a piece of a longer input; the last record then stays
// unread (consumed = where its header line starts) because its sequence may continue in the next piece.
FastqChunk fasta_batch_create(int device, const char *text, size_t n, bool final, hipStream_t st, bool *ended, DevPool *pool) {
    RB_REQUIRE(text || n == 0, "rb_batch_create_fasta: null text");
    RB_REQUIRE(n < 0xFFFFFF00ull, "rb_batch_create_fasta: at most 4 GiB of text per call (got %zu bytes)", n);
    RB_HIP(hipSetDevice(device));
    if (!final && n && text[n - 1] == '\r') --n;
    FastqChunk out;
    if (ended) *ended = false;
    rb_batch *b = new rb_batch();
    struct Guard { rb_batch *b; ~Guard() { if (b) rb_batch_destroy(b); } } guard{b};
    b->device = device;
    const uint32_t un = (uint32_t)n, ntiles = (uint32_t)(((uint64_t)n + FQ_TILE - 1) / FQ_TILE);   // 64-bit: n + FQ_TILE - 1 wraps in 32 bits just below 4 GiB
    TmpBuf d_text, d_cnt, d_base, d_tmp, d_ls, d_ts, d_tl, d_ish, d_kind, d_cum, d_hidx, d_hl, d_len, d_nw, d_woff, d_err;
    for (TmpBuf *q : {&d_text, &d_cnt, &d_base, &d_tmp, &d_ls, &d_ts, &d_tl, &d_ish, &d_kind, &d_cum, &d_hidx, &d_hl, &d_len, &d_nw, &d_woff, &d_err}) q->pool = pool;
    uint8_t *t = d_text.alloc<uint8_t>((size_t)ntiles * FQ_TILE + 64);
    RB_HIP(hipMemsetAsync(t + n, 0, (size_t)ntiles * FQ_TILE + 64 - n, st));
    if (n) RB_HIP(SlabPin::copy(t, text, n, st));            // (split at the boundaries a slab-wise registration of the text would have: SlabPin)
    uint32_t *cnt = d_cnt.alloc<uint32_t>(ntiles + 1), *base = d_base.alloc<uint32_t>(ntiles + 1);
    void *tmp = d_tmp.alloc<uint8_t>(scan_temp_bytes((size_t)ntiles + 1));
    RB_HIP(hipMemsetAsync(cnt + ntiles, 0, 4, st));
    if (ntiles) hipLaunchKernelGGL(k_fq_eol_count, dim3(ntiles), dim3(FQ_TPB), 0, st, t, un, cnt);
    exclusive_scan_u32(tmp, scan_temp_bytes((size_t)ntiles + 1), cnt, base, (size_t)ntiles + 1, st);
    uint32_t eols = 0;
    RB_HIP(hipMemcpyAsync(&eols, base + ntiles, 4, hipMemcpyDeviceToHost, st));
    RB_HIP(hipStreamSynchronize(st));
    const bool open_tail = final && n > 0 && !(text[n - 1] == '\n' || text[n - 1] == '\r');
    const uint32_t n_lines = eols + (open_tail ? 1u : 0u);          // a piece's unfinished last line belongs to the next piece
    uint32_t *ls = d_ls.alloc<uint32_t>((size_t)eols + 2);
    if (ntiles) hipLaunchKernelGGL(k_fq_line_starts, dim3(ntiles), dim3(FQ_TPB), 0, st, t, un, base, ls);
    else RB_HIP(hipMemsetAsync(ls, 0, 4, st));
    const uint32_t sentinel = un + 1u;
    RB_HIP(hipMemcpyAsync(ls + eols + 1, &sentinel, 4, hipMemcpyHostToDevice, st));
    const size_t nl1 = (size_t)n_lines + 1;
    uint32_t *ts = d_ts.alloc<uint32_t>(nl1), *tl = d_tl.alloc<uint32_t>(nl1), *ish = d_ish.alloc<uint32_t>(nl1), *cum = d_cum.alloc<uint32_t>(nl1),
             *hidx = d_hidx.alloc<uint32_t>(nl1);
    uint8_t *kind = d_kind.alloc<uint8_t>(nl1);
    uint32_t *err = d_err.alloc<uint32_t>(16);
    const uint32_t err0[8] = {~0u, ~0u, 0u, 0u, ~0u, 0u, 0u, 0u};   // [0] bad line, [1] end line, [3] longest, [4..5] min / max words, [6..7] bases
    RB_HIP(hipMemcpyAsync(err, err0, sizeof err0, hipMemcpyHostToDevice, st));
    RB_HIP(hipMemsetAsync(tl + n_lines, 0, 4, st));
    RB_HIP(hipMemsetAsync(ish + n_lines, 0, 4, st));
    if (n_lines) {
        hipLaunchKernelGGL(k_fa_lines, dim3((n_lines + 255u) / 256u), dim3(256), 0, st, t, un, ls, n_lines, ts, tl, ish, kind);
        hipLaunchKernelGGL(k_fa_check, dim3((n_lines + 255u) / 256u), dim3(256), 0, st, kind, n_lines, err);
    }
    uint32_t h2[2] = {~0u, ~0u};
    RB_HIP(hipMemcpyAsync(h2, err, 8, hipMemcpyDeviceToHost, st));
    RB_HIP(hipStreamSynchronize(st));
    RB_REQUIRE(h2[0] >= h2[1] || h2[0] == ~0u, "rb_batch_create_fasta: Incorrect FASTA header format");
    const bool end_seen = h2[1] != ~0u;
    const uint32_t limit = end_seen ? h2[1] : n_lines;                // lines that take part
    if (ended) *ended = end_seen;
    TmpBuf d_tmp2;
    for (TmpBuf *q : {&d_tmp2}) q->pool = pool;
    void *tmp2 = d_tmp2.alloc<uint8_t>(scan_temp_bytes(nl1));
    exclusive_scan_u32(tmp2, scan_temp_bytes(nl1), tl, cum, nl1, st);
    exclusive_scan_u32(tmp2, scan_temp_bytes(nl1), ish, hidx, nl1, st);
    uint32_t n_headers = 0;
    RB_HIP(hipMemcpyAsync(&n_headers, hidx + limit, 4, hipMemcpyDeviceToHost, st));
    RB_HIP(hipStreamSynchronize(st));
    uint32_t *hl = d_hl.alloc<uint32_t>((size_t)n_headers + 2);
    if (limit) hipLaunchKernelGGL(k_fa_header_lines, dim3((limit + 255u) / 256u), dim3(256), 0, st, ish, hidx, limit, hl);
    RB_HIP(hipMemcpyAsync(hl + n_headers, &limit, 4, hipMemcpyHostToDevice, st));
    uint32_t last_h = 0;
    if (n_headers) RB_HIP(hipMemcpyAsync(&last_h, hl + (n_headers - 1), 4, hipMemcpyDeviceToHost, st));
    RB_HIP(hipStreamSynchronize(st));
    // records handed out: all of them when the text ends here (or the iteration ended) — except a header that is the file's very
    // last line (FastaReader.next returns null when nothing follows a pending header) — else all but the last one
    uint32_t R = n_headers;
    const bool whole = final || end_seen;
    if (whole) { if (R && !end_seen && last_h == n_lines - 1u) --R; }
    else if (R) --R;
    uint32_t consumed_line = limit;                                   // first line that was not consumed
    if (!whole) consumed_line = n_headers ? last_h : 0u;
    uint32_t *ln = d_len.alloc<uint32_t>(R), *nw = d_nw.alloc<uint32_t>((size_t)R + 1), *woff = d_woff.alloc<uint32_t>((size_t)R + 1);
    RB_HIP(hipMemsetAsync(nw + R, 0, 4, st));
    if (R) hipLaunchKernelGGL(k_fa_records, dim3((R + 255u) / 256u), dim3(256), 0, st, hl, cum, R, ln, nw, err, reinterpret_cast<unsigned long long *>(err + 6));
    TmpBuf d_tmp3;
    for (TmpBuf *q : {&d_tmp3}) q->pool = pool;
    void *tmp3 = d_tmp3.alloc<uint8_t>(scan_temp_bytes((size_t)R + 1));
    exclusive_scan_u32(tmp3, scan_temp_bytes((size_t)R + 1), nw, woff, (size_t)R + 1, st);
    uint32_t herr[8], consumed32 = 0;
    b->h_woff.assign((size_t)R + 1, 0u);
    RB_HIP(hipMemcpyAsync(herr, err, sizeof herr, hipMemcpyDeviceToHost, st));
    RB_HIP(hipMemcpyAsync(b->h_woff.data(), woff, ((size_t)R + 1) * 4, hipMemcpyDeviceToHost, st));
    if (consumed_line <= eols) RB_HIP(hipMemcpyAsync(&consumed32, ls + consumed_line, 4, hipMemcpyDeviceToHost, st));
    RB_HIP(hipStreamSynchronize(st));
    out.consumed = whole ? n : (size_t)consumed32;
    out.records = R;
    b->n_reads = R;
    b->n_words = b->h_woff[R];
    b->max_len = herr[3];
    b->wpr_uniform = (R && herr[4] == herr[5]) ? herr[4] : 0u;
    unsigned long long nb; memcpy(&nb, herr + 6, 8);
    b->n_bases = (int64_t)nb;
    alloc_text_batch(b, R, pool);
    RB_HIP(hipMemcpyAsync(b->woff, woff, ((size_t)R + 1) * 4, hipMemcpyDeviceToDevice, st));
    if (R) RB_HIP(hipMemcpyAsync(b->len, ln, (size_t)R * 4, hipMemcpyDeviceToDevice, st));
    if (b->n_words)
        hipLaunchKernelGGL(k_fa_encode, dim3(blocks_for(b->n_words)), dim3(TPB), 0, st, t, hl, cum, ts, tl, ln, b->woff, (int64_t)R, b->n_words, b->codes, b->valid,
                           b->word_read);
    RB_HIP(hipGetLastError());
    RB_HIP(hipStreamSynchronize(st));
    out.b = b; guard.b = nullptr;
    return out;
}
}  // namespace rb

extern "C" {

int rb_batch_create_fasta(int device, const char *text, size_t len, int final, rb_batch **out, size_t *consumed, int *ended) {
    return guarded([&] {
        RB_REQUIRE(out && consumed, "rb_batch_create_fasta: null argument");
        bool e = false;
        const FastqChunk c = rb::fasta_batch_create(device, text, len, final != 0, nullptr, &e);
        *out = c.b; *consumed = c.consumed;
        if (ended) *ended = e ? 1 : 0;
    });
}

int rb_batch_create_fastq(int device, const char *text, size_t len, int final, int min_base_qual, int use_qual, rb_batch **out, size_t *consumed) {
    return guarded([&] {
        RB_REQUIRE(out && consumed, "rb_batch_create_fastq: null argument");
        const FastqChunk c = rb::fastq_batch_create(device, text, len, final != 0, min_base_qual, use_qual != 0, nullptr);
        *out = c.b; *consumed = c.consumed;
    });
}

// FileUtils.getTextFileReader for ".gz" (R/util/FileUtils.java:50-57: a GZIPInputStream, which reads every member of a
// concatenated file).  A gzip file is a sequence of members; a member announces its uncompressed size in its last four bytes
// (ISIZE) but not its compressed size — except in BGZF files (bgzip: every member carries an extra field 'B','C' with its total
// size and holds at most 64 KiB), where the member boundaries and therefore every member's place in the output are known up
// front: those are inflated by all threads at once.  Anything else is inflated member after member on one thread.
namespace {
struct GzMember { size_t off, csize, usize, uoff; };
// BGZF member at p (RFC 1952 header with FEXTRA and a 'B','C' subfield of length 2): its total size, else 0
size_t bgzf_member_size(const unsigned char *p, size_t left) {
    if (left < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
    if (12 + xlen > left) return 0;
    for (size_t q = 12; q + 4 <= 12 + xlen;) {
        const size_t slen = (size_t)p[q + 2] | ((size_t)p[q + 3] << 8);
        if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) {
            const size_t bsize = ((size_t)p[q + 4] | ((size_t)p[q + 5] << 8)) + 1;
            return bsize >= 26 && bsize <= left ? bsize : 0;
        }
        q += 4 + slen;
    }
    return 0;
}
void inflate_member(const unsigned char *src, size_t n, unsigned char *dst, size_t cap, size_t *produced, size_t *consumed) {
    z_stream z;
    memset(&z, 0, sizeof z);
    RB_REQUIRE(inflateInit2(&z, 15 + 16) == Z_OK, "rb_gunzip: inflateInit2 failed");
    size_t in = 0, out = 0;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        const size_t ci = std::min(n - in, (size_t)1 << 30), co = std::min(cap - out, (size_t)1 << 30);
        z.next_in = const_cast<unsigned char *>(src + in); z.avail_in = (uInt)ci;
        z.next_out = dst + out; z.avail_out = (uInt)co;
        rc = inflate(&z, Z_NO_FLUSH);
        in += ci - z.avail_in; out += co - z.avail_out;
        if (rc == Z_STREAM_END) break;
        if (rc != Z_OK || (z.avail_in == ci && z.avail_out == co)) {
            inflateEnd(&z);
            set_error((in >= n && out < cap) ? "rb_gunzip: unexpected end of the gzip data" : (out >= cap && (rc == Z_OK || rc == Z_BUF_ERROR)) ? "rb_gunzip: output buffer too small"
                                                                                                       : "rb_gunzip: not in gzip format / corrupt data (zlib %d)", rc);
            throw HipError{RB_ERR_INVALID};
        }
    }
    inflateEnd(&z);
    *produced = out; *consumed = in;
}
}  // namespace

int rb_gunzip(const void *src_, size_t n, int n_threads, void *dst_, size_t cap, size_t *out_len) {
    return guarded([&] {
        RB_REQUIRE((src_ || n == 0) && out_len, "rb_gunzip: null argument");
        const unsigned char *src = static_cast<const unsigned char *>(src_);
        unsigned char *dst = static_cast<unsigned char *>(dst_);
        // BGZF all the way?  then the members and their uncompressed sizes are known without inflating anything
        std::vector<GzMember> mem;
        size_t off = 0, total = 0;
        bool bgzf = n > 0;
        while (off < n) {
            const size_t ms = bgzf_member_size(src + off, n - off);
            if (!ms) { bgzf = false; break; }
            const unsigned char *t = src + off + ms - 4;
            const size_t us = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
            mem.push_back({off, ms, us, total});
            off += ms; total += us;
        }
        if (bgzf) {
            *out_len = total;
            if (!dst) return;
            RB_REQUIRE(cap >= total, "rb_gunzip: output buffer too small (%zu bytes needed)", total);
            const int T = std::max(1, std::min(n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency(), 64));
            std::vector<int> rcs((size_t)T, RB_OK);
            std::vector<std::string> errs((size_t)T);
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                rcs[(size_t)t] = guarded([&] {
                    for (size_t i = mem.size() * (size_t)t / (size_t)T; i < mem.size() * (size_t)(t + 1) / (size_t)T; ++i) {
                        size_t got = 0, used = 0;
                        inflate_member(src + mem[i].off, mem[i].csize, dst + mem[i].uoff, mem[i].usize, &got, &used);
                        RB_REQUIRE(got == mem[i].usize, "rb_gunzip: BGZF block %zu inflates to %zu bytes, its trailer says %zu", i, got, mem[i].usize);
                    }
                });
                if (rcs[(size_t)t] != RB_OK) errs[(size_t)t] = rb_last_error();
            });
            for (auto &x : th) x.join();
            for (int t = 0; t < T; ++t) if (rcs[(size_t)t] != RB_OK) { set_error("%s", errs[(size_t)t].c_str()); throw HipError{rcs[(size_t)t]}; }
            return;
        }
        // generic gzip: member after member, as GZIPInputStream reads them — after a member it tries to read another header and
        // treats ANYTHING that is not one (zero padding, garbage, a truncated header) as the end of the stream, silently
        // (java.util.zip.GZIPInputStream.readTrailer: "catch (IOException ze) { return true; }"); only the FIRST member must be gzip
        if (!dst) {     // size query: inflate into a scratch window and count
            std::vector<unsigned char> scratch((size_t)8 << 20);
            size_t in = 0, out = 0;
            while (in < n) {
                if (in > 0 && !(in + 1 < n && src[in] == 0x1f && src[in + 1] == 0x8b)) break;      // not another member: end of stream
                z_stream z;
                memset(&z, 0, sizeof z);
                RB_REQUIRE(inflateInit2(&z, 15 + 16) == Z_OK, "rb_gunzip: inflateInit2 failed");
                int rc = Z_OK;
                while (rc != Z_STREAM_END) {
                    const size_t ci = std::min(n - in, (size_t)1 << 30);
                    z.next_in = const_cast<unsigned char *>(src + in); z.avail_in = (uInt)ci;
                    z.next_out = scratch.data(); z.avail_out = (uInt)scratch.size();
                    rc = inflate(&z, Z_NO_FLUSH);
                    in += ci - z.avail_in; out += scratch.size() - z.avail_out;
                    if (rc == Z_BUF_ERROR && in >= n) { inflateEnd(&z); set_error("rb_gunzip: unexpected end of the gzip data"); throw HipError{RB_ERR_INVALID}; }
                    if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&z); set_error("rb_gunzip: not in gzip format / corrupt data (zlib %d)", rc); throw HipError{RB_ERR_INVALID}; }
                    if (rc == Z_OK && in >= n && z.avail_out == scratch.size()) { inflateEnd(&z); set_error("rb_gunzip: unexpected end of the gzip data"); throw HipError{RB_ERR_INVALID}; }
                }
                inflateEnd(&z);
            }
            *out_len = out;
            return;
        }
        size_t in = 0, out = 0;
        while (in < n) {
            if (in > 0 && !(in + 1 < n && src[in] == 0x1f && src[in + 1] == 0x8b)) break;          // not another member: end of stream
            size_t got = 0, used = 0;
            inflate_member(src + in, n - in, dst + out, cap - out, &got, &used);
            in += used; out += got;
        }
        *out_len = out;
    });
}

int rb_fastq_split(const char *text, size_t len, int n_threads, char *seq, char *qual, int64_t *offsets, int64_t cap_reads, int64_t *n_reads) {
    return guarded([&] {
        RB_REQUIRE((text || len == 0) && n_reads, "rb_fastq_split: null argument");
        const int T = std::max(1, std::min(n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency(), 64));
        std::vector<Chunk> ch((size_t)T);
        for (int t = 0; t < T; ++t) { ch[(size_t)t].b = len * (size_t)t / (size_t)T; ch[(size_t)t].e = len * (size_t)(t + 1) / (size_t)T; }
        // pass 1: ends of line per chunk
        std::vector<int64_t> eols((size_t)T, 0);
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                int64_t c = 0;
                for (size_t i = ch[(size_t)t].b; i < ch[(size_t)t].e; ++i) c += eol_at(text, len, i);
                eols[(size_t)t] = c;
            });
            for (auto &x : th) x.join();
        }
        int64_t total_eols = 0;
        for (int t = 0; t < T; ++t) { ch[(size_t)t].lines_before = total_eols; total_eols += eols[(size_t)t]; }
        // complete lines: one per end of line, plus a last line without one
        const bool open_tail = len > 0 && !eol_at(text, len, len - 1);
        const int64_t n_lines = total_eols + (open_tail ? 1 : 0);
        const int64_t records = n_lines / 4;                                // a truncated record is dropped
        *n_reads = records;
        if (!offsets) return;                                               // count only
        RB_REQUIRE(cap_reads >= records, "rb_fastq_split: room for %lld reads, %lld present", (long long)cap_reads, (long long)records);
        // pass 2: per chunk, the start and length of every line whose number is 4r+1 (sequence) or 4r+3 (quality)
        std::vector<int64_t> seq_len((size_t)records, 0);
        std::vector<size_t> seq_pos((size_t)records, 0), qual_pos((size_t)records, 0);
        std::vector<int64_t> qual_len((size_t)records, 0);
        std::vector<int> bad((size_t)T, 0);
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                // the lines that START in [b, e): a line starts at 0 and after every end of line; its number is the number
                // of ends of line before its start
                const size_t b = ch[(size_t)t].b, e = ch[(size_t)t].e;
                int64_t line = ch[(size_t)t].lines_before;
                size_t s = b;
                if (b > 0 && !eol_at(text, len, b - 1)) {                  // b is inside a line that started earlier
                    while (s < e && !eol_at(text, len, s)) ++s;
                    if (s >= e) return;                                     // no line starts in this chunk
                    ++s; ++line;
                }
                while (s < e && s < len) {
                    size_t j = s;
                    while (j < len && !eol_at(text, len, j)) ++j;           // j = end of line position (or len)
                    size_t ce = j;
                    if (j < len && text[j] == '\n' && ce > s && text[ce - 1] == '\r') --ce;   // \r\n
                    const int64_t rec = line >> 2, field = line & 3;
                    if (rec < records) {
                        if (field == 0) { if (ce == s || text[s] != '@') bad[(size_t)t] = 1; }
                        else if (field == 1) { seq_pos[(size_t)rec] = s; seq_len[(size_t)rec] = (int64_t)(ce - s); }
                        else if (field == 2) { if (ce == s || text[s] != '+') bad[(size_t)t] = 2; }
                        else { qual_pos[(size_t)rec] = s; qual_len[(size_t)rec] = (int64_t)(ce - s); }
                    }
                    s = j + 1; ++line;
                }
            });
            for (auto &x : th) x.join();
        }
        for (int t = 0; t < T; ++t) {
            RB_REQUIRE(bad[(size_t)t] != 1, "rb_fastq_split: Line 1 of FASTQ record is expected to start with '@'");
            RB_REQUIRE(bad[(size_t)t] != 2, "rb_fastq_split: Line 3 of FASTQ record is expected to start with '+'");
        }
        offsets[0] = 0;
        for (int64_t r = 0; r < records; ++r) {
            RB_REQUIRE(!qual || qual_len[(size_t)r] == seq_len[(size_t)r], "rb_fastq_split: record %lld has %lld bases and %lld qualities", (long long)r,
                       (long long)seq_len[(size_t)r], (long long)qual_len[(size_t)r]);
            offsets[r + 1] = offsets[r] + seq_len[(size_t)r];
        }
        if (!seq) return;
        {   // pass 3: copy the fields
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                for (int64_t r = records * t / T; r < records * (t + 1) / T; ++r) {
                    memcpy(seq + offsets[r], text + seq_pos[(size_t)r], (size_t)seq_len[(size_t)r]);
                    if (qual) memcpy(qual + offsets[r], text + qual_pos[(size_t)r], (size_t)seq_len[(size_t)r]);
                }
            });
            for (auto &x : th) x.join();
        }
    });
}

}  // extern "C"
