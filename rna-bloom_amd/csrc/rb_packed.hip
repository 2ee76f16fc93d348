// rb_packed.hip — read batches in the build's own packed format, resident in HOST memory ("input already
// resident in host memory in the build's batch format"), streamed into HBM while the insert pipeline works.
//
// Host format of n_reads reads (what rb_batch_download_packed writes and a caller that packs its reads once keeps):
//   codes[w]  u64  32 bases of a read, 2 bits each, base i of the word at bits 2i..2i+1 (A C G T = 0 1 2 3)
//   valid[w]  u32  bit i = base i is usable (one of ACGTU and quality >= the threshold used when the reads were packed)
//   len[r]    u32  bases of read r; read r owns ceil(len[r] / 32) consecutive words, reads follow each other in order
// 12 bytes per 32 bases + 4 per read (6.4 GB for config 2's 100 M reads of 150 bases: 112 ms at the 57 GB/s the link gives from
// pinned memory).  The two device-only columns of a batch — the owning read of every word and the word offset of every
// read — are computed on the GPU from len[] (one scan), not shipped.
//
// rb_packed_stream: two device batches that take turns.  begin() hands a chunk to a helper thread that enqueues the
// copies and the offset kernels on the stream's own HIP stream and waits for them; finish() joins it and returns the
// batch.  The caller inserts chunk c (rb_graph_add_batch) between begin(c + 1) and finish(c + 1): the upload of the next chunk
// runs beside the insert of this one on the copy engines.  The reference's counterpart is the reader side of
// FastqToGraphWorker (R/RNABloom.java:551-634: reads pulled from a FastqReader while other workers insert).
#include <string.h>
#include <time.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <utility>
#include <string>
#include <thread>
#include <vector>

#include "rb_pipeline.hpp"

using namespace rb;

namespace {
// wc[r] = words of read r (wc[n] = 0 closes the scan); st: [0] longest read, [1] fewest / [2] most words of a read, [4..5] sum of len (u64)
__global__ void k_packed_words(const uint32_t *__restrict__ len, int64_t n, uint32_t *__restrict__ wc, uint32_t *__restrict__ st) {
    // grid-stride: a few thousand wavefronts in all, so that the four same-address atomics at the end (they queue up at ~10 ns apiece at the memory
    // side) are thousands, not one set per 64 reads — 200 K sets cost 9 ms of an 11 ms upload prologue when every wavefront of a flat grid did them
    uint32_t mx = 0, wmin = 0xFFFFFFFFu, wmax = 0;
    unsigned long long sum = 0;
    bool any = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x) {
        if (i == n) { wc[i] = 0; break; }
        const uint32_t l = len[i], w = (l + 31u) >> 5;
        wc[i] = w;
        mx = l > mx ? l : mx; wmin = w < wmin ? w : wmin; wmax = w > wmax ? w : wmax; sum += l; any = true;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = (uint32_t)__shfl_down((int)mx, o, 64), b = (uint32_t)__shfl_down((int)wmin, o, 64), c = (uint32_t)__shfl_down((int)wmax, o, 64);
        mx = a > mx ? a : mx; wmin = b < wmin ? b : wmin; wmax = c > wmax ? c : wmax;
        sum += __shfl_down(sum, o, 64);
    }
    if ((threadIdx.x & 63u) == 0u && __ballot(any)) {
        atomicMax(&st[0], mx); atomicMin(&st[1], wmin); atomicMax(&st[2], wmax);
        atomicAdd(reinterpret_cast<unsigned long long *>(st + 4), sum);
    }
}
// streamed ASCII ingest: lengths from the caller's base offsets; an offset that runs backwards or a read of 2^30 bases and more raises st[6]
__global__ void k_ascii_lens(const int64_t *__restrict__ off, int64_t n, uint32_t *__restrict__ len, uint32_t *__restrict__ st) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t l = off[i + 1] - off[i];
        if (l < 0 || l >= ((int64_t)1 << 30)) { bad = true; l = 0; }
        len[i] = (uint32_t)l;
    }
    if (__ballot(bad) && (threadIdx.x & 63u) == 0u) atomicOr(&st[6], 1u);
}
// the owning read of every word
__global__ void k_packed_word_read(const uint32_t *__restrict__ woff, int64_t n, uint32_t *__restrict__ word_read) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t a = woff[r], e = woff[r + 1];
    for (uint32_t w = a; w < e; ++w) word_read[w] = (uint32_t)r;
}
}  // namespace

struct rb_packed_stream {
    int device = 0;
    int64_t max_reads = 0, max_words = 0;
    hipStream_t st = nullptr;
    struct Buf {
        rb_batch b;                               // its arrays point into the DevBufs below (never rb_batch_destroy'ed)
        DevBuf codes, valid, word_read, woff, len, wc, temp, stats;
        uint32_t *h_woff = nullptr;               // pinned host copy of woff, (max_reads + 1) entries
        int64_t h_woff_cap = -1;
    } buf[2];
    uint32_t *h_stats = nullptr;                  // pinned, 8 words
    int fill = 0;                                 // the buffer the next begin() fills
    bool pending = false;
    // ONE helper thread for the stream's lifetime (a thread's first HIP call sets up its context: milliseconds — per chunk, had every begin()
    // started a thread of its own): begin() posts a job, finish() waits for it
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool job_posted = false, job_done = true, quit = false;
    int worker_rc = RB_OK;
    std::string worker_err;
    void run() {
        for (;;) {
            std::function<void()> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return job_posted || quit; });
                if (quit && !job_posted) return;
                j = std::move(job); job_posted = false;
            }
            j();
            { std::lock_guard<std::mutex> lk(m); job_done = true; }
            cv.notify_all();
        }
    }
    void post(std::function<void()> j) {
        { std::lock_guard<std::mutex> lk(m); job = std::move(j); job_posted = true; job_done = false; }
        cv.notify_all();
    }
    void wait_done() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return job_done; }); }
    void stop() {
        if (!worker.joinable()) return;
        wait_done();
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv.notify_all();
        worker.join();
    }
};

namespace {
void upload_chunk(rb_packed_stream *s, int slot, const uint64_t *codes, const uint32_t *valid, const uint32_t *len, int64_t n_reads, int64_t n_words) {
    RB_HIP(hipSetDevice(s->device));
    rb_packed_stream::Buf &B = s->buf[slot];
    hipStream_t st = s->st;
    const bool tdbg = getenv("RB_HOST_TIMING") != nullptr;
    auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
    const double t_0 = now();
    const size_t nw = (size_t)std::max<int64_t>(n_words, 1), nr = (size_t)std::max<int64_t>(n_reads, 1);
    RB_REQUIRE(n_words <= s->max_words && n_reads <= s->max_reads, "packed stream: a chunk of %lld reads / %lld words, the stream was created for %lld / %lld",
               (long long)n_reads, (long long)n_words, (long long)s->max_reads, (long long)s->max_words);
    // caller's buffers: pinned for the call if they are not already (best effort; 8 ms per GB, here on the helper thread)
    HostPin p0(codes, (size_t)n_words * 8), p1(valid, (size_t)n_words * 4), p2(len, (size_t)n_reads * 4);
    uint32_t init[8] = {0u, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u};
    memcpy(s->h_stats + 8, init, sizeof init);
    RB_HIP(hipMemcpyAsync(B.stats.p, s->h_stats + 8, sizeof init, hipMemcpyHostToDevice, st));
    if (n_reads) RB_HIP(hipMemcpyAsync(B.len.p, len, (size_t)n_reads * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_packed_words, dim3(std::min<unsigned>(blocks_for(n_reads + 1), 2048u)), dim3(TPB), 0, st, B.len.as<uint32_t>(), n_reads, B.wc.as<uint32_t>(), B.stats.as<uint32_t>());
    exclusive_scan_u32(B.temp.p, B.temp.cap, B.wc.as<uint32_t>(), B.woff.as<uint32_t>(), (size_t)n_reads + 1, st);
    RB_HIP(hipMemcpyAsync(s->h_stats, B.stats.p, 32, hipMemcpyDeviceToHost, st));
    RB_REQUIRE(n_reads <= B.h_woff_cap, "packed stream: a chunk of %lld reads, the stream was created for %lld", (long long)n_reads, (long long)B.h_woff_cap);
    B.b.h_woff.borrow(B.h_woff, (size_t)n_reads + 1);                  // pinned: the copy back runs at link speed
    RB_HIP(hipMemcpyAsync(B.h_woff, B.woff.p, ((size_t)n_reads + 1) * 4, hipMemcpyDeviceToHost, st));
    double t_1 = 0;
    if (tdbg) { RB_HIP(hipStreamSynchronize(st)); t_1 = now(); }
    if (n_words) {
        RB_HIP(hipMemcpyAsync(B.codes.p, codes, (size_t)n_words * 8, hipMemcpyHostToDevice, st));
        RB_HIP(hipMemcpyAsync(B.valid.p, valid, (size_t)n_words * 4, hipMemcpyHostToDevice, st));
    }
    RB_HIP(hipStreamSynchronize(st));
    const double t_2 = now();
    // the lengths must describe exactly the words that were handed over — checked before anything indexes by them
    RB_REQUIRE((int64_t)B.b.h_woff[(size_t)n_reads] == n_words, "packed batch: the lengths of %lld reads add up to %u words, %lld were passed",
               (long long)n_reads, B.b.h_woff[(size_t)n_reads], (long long)n_words);
    if (n_reads) hipLaunchKernelGGL(k_packed_word_read, dim3(blocks_for(n_reads)), dim3(TPB), 0, st, B.woff.as<uint32_t>(), n_reads, B.word_read.as<uint32_t>());
    RB_HIP(hipGetLastError());
    RB_HIP(hipStreamSynchronize(st));
    if (tdbg) fprintf(stderr, "[rb] packed chunk of %lld reads / %lld words: lengths + offsets %.2f ms, codes + valid %.2f ms (%.1f GB/s), word owners %.2f ms\n", (long long)n_reads,
                      (long long)n_words, t_1 - t_0, t_2 - t_1, (double)n_words * 12.0 / ((t_2 - t_1) * 1e6), now() - t_2);
    rb_batch &b = B.b;
    b.device = s->device; b.n_reads = n_reads; b.n_words = n_words;
    b.max_len = s->h_stats[0];
    b.wpr_uniform = (n_reads && s->h_stats[1] == s->h_stats[2]) ? s->h_stats[1] : 0u;
    b.n_bases = (int64_t)(((uint64_t)s->h_stats[5] << 32) | s->h_stats[4]);
    b.codes = B.codes.as<uint64_t>(); b.valid = B.valid.as<uint32_t>(); b.word_read = B.word_read.as<uint32_t>();
    b.woff = B.woff.as<uint32_t>(); b.len = B.len.as<uint32_t>(); b.rnz = nullptr;
    b.device_bytes = nw * 16 + (nr + 1) * 4 + nr * 4;
}
}  // namespace

extern "C" {

int rb_host_alloc(size_t bytes, void **out) {
    return guarded([&] {
        RB_REQUIRE(out, "rb_host_alloc: null argument");
        RB_HIP(hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault));
    });
}
int rb_host_free(void *p) {
    if (p) (void)hipHostFree(p);
    return RB_OK;
}

int rb_batch_download_packed(const rb_batch *b, int64_t first, int64_t n, uint64_t *codes, uint32_t *valid, uint32_t *len, int64_t *n_words) {
    return guarded([&] {
        RB_REQUIRE(b && n_words && first >= 0 && n >= 0 && first + n <= b->n_reads, "rb_batch_download_packed: bad range");
        RB_HIP(hipSetDevice(b->device));
        const int64_t w0 = n ? (int64_t)b->h_woff[(size_t)first] : 0, w1 = n ? (int64_t)b->h_woff[(size_t)(first + n)] : 0;
        *n_words = w1 - w0;
        if (!codes && !valid && !len) return;                 // size query
        RB_REQUIRE(codes && valid && len, "rb_batch_download_packed: null output array");
        if (n) RB_HIP(hipMemcpy(len, b->len + first, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (w1 > w0) {
            RB_HIP(hipMemcpy(codes, b->codes + w0, (size_t)(w1 - w0) * 8, hipMemcpyDeviceToHost));
            RB_HIP(hipMemcpy(valid, b->valid + w0, (size_t)(w1 - w0) * 4, hipMemcpyDeviceToHost));
        }
    });
}

int rb_packed_stream_create(int device, int64_t max_reads, int64_t max_words, rb_packed_stream **out) {
    rb_packed_stream *s = nullptr;
    int rc = guarded([&] {
        RB_REQUIRE(out && max_reads >= 0 && max_words >= 0 && max_words < 0xFFFFFFF0ll && max_reads < 0xFFFFFFF0ll, "rb_packed_stream_create: bad argument");
        int ndev = 0;
        RB_HIP(hipGetDeviceCount(&ndev));
        RB_REQUIRE(device >= 0 && device < ndev, "rb_packed_stream_create: device %d not present", device);
        RB_HIP(hipSetDevice(device));
        s = new rb_packed_stream();
        s->device = device; s->max_reads = max_reads; s->max_words = max_words;
        RB_HIP(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
        RB_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_stats), 64, hipHostMallocDefault));
        for (auto &B : s->buf) {                   // both buffers at their full size now: growing one later frees memory, and hipFree waits for the device
            const size_t nw = (size_t)std::max<int64_t>(max_words, 1), nr = (size_t)std::max<int64_t>(max_reads, 1);
            B.codes.reserve(nw * 8); B.valid.reserve(nw * 4); B.word_read.reserve(nw * 4);
            B.woff.reserve((nr + 1) * 4); B.len.reserve(nr * 4); B.wc.reserve((nr + 1) * 4); B.stats.reserve(64);
            B.temp.reserve(scan_temp_bytes(nr + 1));
            RB_HIP(hipHostMalloc(reinterpret_cast<void **>(&B.h_woff), (nr + 1) * 4, hipHostMallocDefault));
            B.h_woff_cap = max_reads;
        }
        *out = s;
    });
    if (rc != RB_OK && s) rb_packed_stream_destroy(s);
    return rc;
}

int rb_packed_stream_begin(rb_packed_stream *s, const uint64_t *codes, const uint32_t *valid, const uint32_t *len, int64_t n_reads, int64_t n_words) {
    return guarded([&] {
        RB_REQUIRE(s && n_reads >= 0 && n_words >= 0 && (n_reads == 0 || len) && (n_words == 0 || (codes && valid)), "rb_packed_stream_begin: bad argument");
        RB_REQUIRE(!s->pending, "rb_packed_stream_begin: the chunk begun before has not been finished");
        RB_REQUIRE(n_words < 0xFFFFFFF0ll && n_reads < 0xFFFFFFF0ll, "rb_packed_stream_begin: chunk too large (> 2^32 words)");
        const int slot = s->fill;
        s->pending = true;
        s->worker_rc = RB_OK; s->worker_err.clear();
        if (!s->worker.joinable()) s->worker = std::thread([s] { s->run(); });
        s->post([=] {
            s->worker_rc = guarded([&] { upload_chunk(s, slot, codes, valid, len, n_reads, n_words); });
            if (s->worker_rc != RB_OK) s->worker_err = rb_last_error();       // (the error text is thread-local)
        });
    });
}

int rb_packed_stream_finish(rb_packed_stream *s, const rb_batch **out) {
    return guarded([&] {
        RB_REQUIRE(s && out, "rb_packed_stream_finish: null argument");
        RB_REQUIRE(s->pending, "rb_packed_stream_finish: no chunk was begun");
        s->wait_done();
        s->pending = false;
        if (s->worker_rc != RB_OK) { set_error("%s", s->worker_err.c_str()); throw HipError{s->worker_rc}; }
        *out = &s->buf[s->fill].b;             // valid until the begin() after next reuses this buffer
        s->fill ^= 1;
    });
}

int rb_packed_stream_destroy(rb_packed_stream *s) {
    if (!s) return RB_OK;
    s->stop();
    (void)hipSetDevice(s->device);
    if (s->st) { (void)hipStreamSynchronize(s->st); (void)hipStreamDestroy(s->st); }
    for (auto &B : s->buf) {
        for (DevBuf *d : {&B.codes, &B.valid, &B.word_read, &B.woff, &B.len, &B.wc, &B.temp, &B.stats}) d->release();
        if (B.h_woff) (void)hipHostFree(B.h_woff);
        B.b.codes = nullptr; B.b.valid = nullptr; B.b.word_read = nullptr; B.b.woff = nullptr; B.b.len = nullptr;
    }
    if (s->h_stats) (void)hipHostFree(s->h_stats);
    delete s;
    return RB_OK;
}

}  // extern "C"

namespace {
// Everything an insert from packed host memory sends, enqueued on the handle's copy stream without waiting for any of it: the lengths, the kernels
// that turn them into word offsets and word owners, the offsets' way back into pinned memory (event ev_woff), then codes and valid in pieces cut
// at WORD boundaries (no offset is needed to cut them), an event behind each.
void ingest_begin(rb_graph *g, rb_graph::PackedIngest &K, const uint64_t *codes, const uint32_t *valid, const uint32_t *len, int64_t n_reads, int64_t n_words, int64_t piece_reads) {
    if (!g->pk_stream) RB_HIP(hipStreamCreateWithFlags(&g->pk_stream, hipStreamNonBlocking));
    hipStream_t st = g->pk_stream;
    if (!K.h_stats) RB_HIP(hipHostMalloc(reinterpret_cast<void **>(&K.h_stats), 64, hipHostMallocDefault));
    if (!K.ev_woff) RB_HIP(hipEventCreateWithFlags(&K.ev_woff, hipEventDisableTiming));
    const size_t nw = (size_t)std::max<int64_t>(n_words, 1), nr = (size_t)n_reads;
    K.codes.reserve(nw * 8); K.valid.reserve(nw * 4); K.word_read.reserve(nw * 4);
    K.woff.reserve((nr + 1) * 4); K.len.reserve(nr * 4); K.wc.reserve((nr + 1) * 4); K.stats.reserve(64); K.temp.reserve(scan_temp_bytes(nr + 1));
    if (K.h_woff_cap < nr + 1) {
        if (K.h_woff) (void)hipHostFree(K.h_woff);
        K.h_woff = nullptr; K.h_woff_cap = 0;
        const size_t want = nr + 1 + (nr >> 3);
        RB_HIP(hipHostMalloc(reinterpret_cast<void **>(&K.h_woff), want * 4, hipHostMallocDefault));
        K.h_woff_cap = want;
    }
    // the caller's arrays: registered for the upload unless they are pinned already (best effort; pageable memory makes every copy below a blocking one)
    for (auto pr : {std::make_pair((const void *)codes, (size_t)n_words * 8), std::make_pair((const void *)valid, (size_t)n_words * 4), std::make_pair((const void *)len, nr * 4)})
        if (pr.first && pr.second > ((size_t)16 << 20) && !getenv("RB_NO_PIN") && !HostPin::pinned_already(pr.first)) {
            if (hipHostRegister(const_cast<void *>(pr.first), pr.second, hipHostRegisterDefault) == hipSuccess) K.pins.push_back(const_cast<void *>(pr.first));
            else (void)hipGetLastError();
        }
    K.src = codes; K.n_reads = n_reads; K.n_words = n_words; K.inflight = true;
    uint32_t init[8] = {0u, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u};
    memcpy(K.h_stats + 8, init, sizeof init);
    RB_HIP(hipMemcpyAsync(K.stats.p, K.h_stats + 8, sizeof init, hipMemcpyHostToDevice, st));
    RB_HIP(hipMemcpyAsync(K.len.p, len, nr * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_packed_words, dim3(std::min<unsigned>(blocks_for(n_reads + 1), 2048u)), dim3(TPB), 0, st, K.len.as<uint32_t>(), n_reads, K.wc.as<uint32_t>(), K.stats.as<uint32_t>());
    exclusive_scan_u32(K.temp.p, K.temp.cap, K.wc.as<uint32_t>(), K.woff.as<uint32_t>(), nr + 1, st);
    RB_HIP(hipMemcpyAsync(K.h_stats, K.stats.p, 32, hipMemcpyDeviceToHost, st));
    RB_HIP(hipMemcpyAsync(K.h_woff, K.woff.p, (nr + 1) * 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_packed_word_read, dim3(blocks_for(n_reads)), dim3(TPB), 0, st, K.woff.as<uint32_t>(), n_reads, K.word_read.as<uint32_t>());
    RB_HIP(hipGetLastError());
    RB_HIP(hipEventRecord(K.ev_woff, st));
    // pieces of words: piece_reads reads' worth each (0: 2^22 words — 50 MB, a millisecond of link — doubling up to 2^25)
    K.wend.clear();
    {
        const int64_t wpr = std::max<int64_t>(1, (n_words + n_reads - 1) / std::max<int64_t>(n_reads, 1));
        int64_t w = 0, step = piece_reads ? std::max<int64_t>(1, piece_reads * wpr) : ((int64_t)1 << 22);
        while (w < n_words) {
            w = std::min(n_words, w + step);
            K.wend.push_back(w);
            if (!piece_reads) step = std::min<int64_t>(step * 2, (int64_t)1 << 25);
        }
    }
    while (K.ev.size() < K.wend.size()) { hipEvent_t e; RB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); K.ev.push_back(e); }
    int64_t w0 = 0;
    for (size_t p = 0; p < K.wend.size(); ++p) {
        const int64_t w1 = K.wend[p];
        RB_HIP(hipMemcpyAsync(K.codes.as<uint64_t>() + w0, codes + w0, (size_t)(w1 - w0) * 8, hipMemcpyHostToDevice, st));
        RB_HIP(hipMemcpyAsync(K.valid.as<uint32_t>() + w0, valid + w0, (size_t)(w1 - w0) * 4, hipMemcpyHostToDevice, st));
        RB_HIP(hipEventRecord(K.ev[p], st));
        w0 = w1;
    }
}
// the upload of a slot is over (or abandoned): wait for it, give the caller's arrays back
void ingest_end(rb_graph *g, rb_graph::PackedIngest &K) {
    if (!K.inflight) return;
    if (K.feeder.joinable()) {             // (a streamed ASCII ingest: an abandoned one stops feeding at the next piece)
        { std::lock_guard<std::mutex> lk(K.fm); K.cancel = true; }
        K.feeder.join();
    }
    const size_t fed = K.enqueued;
    if (!K.rend.empty()) {                   // a streamed ASCII ingest: whatever was fed is waited for (the stream is in order: its last operation)
        if (g->pk_stream) (void)hipStreamSynchronize(g->pk_stream);
        (void)fed;
    } else if (!K.wend.empty()) (void)hipEventSynchronize(K.ev[K.wend.size() - 1]);
    else if (K.ev_woff) (void)hipEventSynchronize(K.ev_woff);
    K.rend.clear(); K.enqueued = 0;
    for (void *p : K.pins) (void)hipHostUnregister(p);
    K.pins.clear();
    K.inflight = false; K.src = nullptr;
}
}  // namespace

namespace {
// Everything a streamed ASCII ingest sends, enqueued on the handle's copy stream without waiting: the caller's base offsets, the kernels that turn
// them into lengths, word offsets (one scan) and their way back into pinned memory (event ev_woff), then piece after piece the bases and
// qualities of <= piece_bases bases into one of two staging buffers and the encode kernel of that piece (k_encode_ascii_t<true>, rb_batch.hip)
// behind it, an event behind each.  The stream is in order: the copy into a staging buffer queues behind the encode that last read it.
void ingest_begin_ascii(rb_graph *g, rb_graph::PackedIngest &K, const char *seq, const char *qual, const int64_t *offsets, int64_t n_reads, int min_q, int64_t piece_bases) {
    if (!g->pk_stream) RB_HIP(hipStreamCreateWithFlags(&g->pk_stream, hipStreamNonBlocking));
    hipStream_t st = g->pk_stream;
    if (!K.h_stats) RB_HIP(hipHostMalloc(reinterpret_cast<void **>(&K.h_stats), 64, hipHostMallocDefault));
    if (!K.ev_woff) RB_HIP(hipEventCreateWithFlags(&K.ev_woff, hipEventDisableTiming));
    const size_t nr = (size_t)n_reads;
    const int64_t base0 = offsets[0], nbases = offsets[n_reads] - base0;
    RB_REQUIRE(nbases >= 0, "rb_graph_add_reads: the offsets run backwards");
    const size_t nw_ub = (size_t)nbases / 32 + nr + 1;                       // every read adds at most one partly filled word
    RB_REQUIRE(nw_ub < 0xFFFFFFF0ull, "rb_graph_add_reads: batch too large (> 2^32 words)");
    K.codes.reserve(nw_ub * 8); K.valid.reserve(nw_ub * 4); K.word_read.reserve(nw_ub * 4);
    K.woff.reserve((nr + 1) * 4); K.len.reserve(nr * 4); K.wc.reserve((nr + 1) * 4); K.stats.reserve(64); K.temp.reserve(scan_temp_bytes(nr + 1));
    K.off.reserve((nr + 1) * 8);
    if (K.h_woff_cap < nr + 1) {
        if (K.h_woff) (void)hipHostFree(K.h_woff);
        K.h_woff = nullptr; K.h_woff_cap = 0;
        const size_t want = nr + 1 + (nr >> 3);
        RB_HIP(hipHostMalloc(reinterpret_cast<void **>(&K.h_woff), want * 4, hipHostMallocDefault));
        K.h_woff_cap = want;
    }
    // the pieces: reads [rend[p - 1], rend[p]) of at most piece_bases bases (at least one read)
    K.rend.clear(); K.wend.clear();
    int64_t max_piece = 0;
    for (int64_t a = 0; a < n_reads;) {
        int64_t lo = a + 1, hi = n_reads;
        while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (offsets[mid] - offsets[a] <= piece_bases) lo = mid; else hi = mid - 1; }
        RB_REQUIRE(offsets[lo] >= offsets[a], "rb_graph_add_reads: the offsets run backwards");
        max_piece = std::max(max_piece, offsets[lo] - offsets[a]);
        K.rend.push_back(lo);
        a = lo;
    }
    for (int q = 0; q < 2; ++q) { K.stage_seq[q].reserve((size_t)std::max<int64_t>(max_piece, 1)); if (qual) K.stage_qual[q].reserve((size_t)std::max<int64_t>(max_piece, 1)); }
    while (K.ev.size() < K.rend.size()) { hipEvent_t e; RB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); K.ev.push_back(e); }
    // the caller's offsets are registered here (they go up first); bases and qualities are registered slab by slab by the feeder below, ahead of the copies
    if ((nr + 1) * 8 > ((size_t)16 << 20) && !getenv("RB_NO_PIN") && !HostPin::pinned_already(offsets)) {
        if (hipHostRegister(const_cast<int64_t *>(offsets), (nr + 1) * 8, hipHostRegisterDefault) == hipSuccess) K.pins.push_back(const_cast<int64_t *>(offsets));
        else (void)hipGetLastError();
    }
    K.src = seq; K.n_reads = n_reads; K.n_words = -1; K.inflight = true;
    uint32_t init[8] = {0u, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u};
    memcpy(K.h_stats + 8, init, sizeof init);
    RB_HIP(hipMemcpyAsync(K.stats.p, K.h_stats + 8, sizeof init, hipMemcpyHostToDevice, st));
    RB_HIP(hipMemcpyAsync(K.off.p, offsets, (nr + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_ascii_lens, dim3(std::min<unsigned>(blocks_for(n_reads), 2048u)), dim3(TPB), 0, st, K.off.as<int64_t>(), n_reads, K.len.as<uint32_t>(), K.stats.as<uint32_t>());
    hipLaunchKernelGGL(k_packed_words, dim3(std::min<unsigned>(blocks_for(n_reads + 1), 2048u)), dim3(TPB), 0, st, K.len.as<uint32_t>(), n_reads, K.wc.as<uint32_t>(), K.stats.as<uint32_t>());
    exclusive_scan_u32(K.temp.p, K.temp.cap, K.wc.as<uint32_t>(), K.woff.as<uint32_t>(), nr + 1, st);
    RB_HIP(hipMemcpyAsync(K.h_stats, K.stats.p, 32, hipMemcpyDeviceToHost, st));
    RB_HIP(hipMemcpyAsync(K.h_woff, K.woff.p, (nr + 1) * 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_packed_word_read, dim3(blocks_for(n_reads)), dim3(TPB), 0, st, K.woff.as<uint32_t>(), n_reads, K.word_read.as<uint32_t>());      // (the pieces' encode kernels look their words' reads up here)
    RB_HIP(hipGetLastError());
    RB_HIP(hipEventRecord(K.ev_woff, st));
    // the feeder: registers what a piece's copies read (slabs that end on 256 MiB boundaries of the address: neighbours never share a page), then
    // enqueues the piece and tells the insert (rb_graph::await_words waits for `enqueued` before it makes a stream wait for the piece's event)
    { std::lock_guard<std::mutex> lk(K.fm); K.enqueued = 0; K.cancel = false; K.feeder_rc = RB_OK; K.feeder_err.clear(); }
    rb_graph::PackedIngest *Kp = &K;
    const int device = g->p.device;
    K.feeder = std::thread([=] {
        rb_graph::PackedIngest &K = *Kp;
        const int rc = guarded([&] {
            RB_HIP(hipSetDevice(device));
            const bool pin = !getenv("RB_NO_PIN");
            constexpr uintptr_t SLAB = (uintptr_t)256 << 20;       // slab boundaries are multiples of it: page-aligned, so neighbouring slabs never share a page
            struct Cursor { const char *base; uintptr_t lo, hi, done; bool on; };
            auto cursor = [&](const char *p) {
                Cursor c{p, 0, 0, 0, false};
                if (!p || !pin || (size_t)nbases <= ((size_t)16 << 20) || HostPin::pinned_already(p + base0)) return c;
                c.lo = reinterpret_cast<uintptr_t>(p + base0); c.hi = c.lo + (uintptr_t)nbases; c.done = c.lo; c.on = true;      // (from the array's own first byte: what lies below it may not be mapped)
                return c;
            };
            Cursor cs = cursor(seq), cq = cursor(qual);
            const bool tdbg = getenv("RB_HOST_TIMING") != nullptr;
            auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
            double t_pin = 0, t_enq = 0; const double t_f0 = now();
            auto pin_to = [&](Cursor &c, uintptr_t need) {             // register slabs until [lo, need) is covered
                while (c.on && c.done < need) {
                    const uintptr_t a = c.done, e = std::min(c.hi, (a & ~(SLAB - 1)) + SLAB);
                    if (hipHostRegister(reinterpret_cast<void *>(a), (size_t)(e - a), hipHostRegisterDefault) == hipSuccess) K.pins.push_back(reinterpret_cast<void *>(a));
                    else { (void)hipGetLastError(); c.on = false; }       // (best effort: the copies still work from pageable memory, slowly)
                    c.done = e;
                }
            };
            int64_t ra = 0;
            for (size_t p = 0; p < K.rend.size(); ++p) {
                { std::lock_guard<std::mutex> lk(K.fm); if (K.cancel) break; }
                const int64_t rb = K.rend[p], bytes = offsets[rb] - offsets[ra];
                const int slot = (int)(p & 1u);
                double tp1 = 0;
                if (bytes) {
                    const double tp0 = now();
                    pin_to(cs, reinterpret_cast<uintptr_t>(seq + offsets[rb]));
                    if (qual) pin_to(cq, reinterpret_cast<uintptr_t>(qual + offsets[rb]));
                    tp1 = now(); t_pin += tp1 - tp0;
                    // (a copy must stay inside ONE registration: a source range over two of them is refused as an invalid argument)
                    auto copy_by_slab = [&](void *dst, const char *src, size_t n) {
                        for (size_t o = 0; o < n;) {
                            const uintptr_t at = reinterpret_cast<uintptr_t>(src + o);
                            const size_t m = std::min<size_t>(n - o, (size_t)(((at & ~(SLAB - 1)) + SLAB) - at));
                            RB_HIP(hipMemcpyAsync(static_cast<char *>(dst) + o, src + o, m, hipMemcpyHostToDevice, st));
                            o += m;
                        }
                    };
                    copy_by_slab(K.stage_seq[slot].p, seq + offsets[ra], (size_t)bytes);
                    if (qual) copy_by_slab(K.stage_qual[slot].p, qual + offsets[ra], (size_t)bytes);
                    launch_encode_ascii_piece(K.stage_seq[slot].as<uint8_t>(), qual ? K.stage_qual[slot].as<uint8_t>() : (const uint8_t *)nullptr, K.off.as<int64_t>(),
                                              K.woff.as<uint32_t>(), ra, rb - ra, offsets[ra], bytes / 32 + (rb - ra) + 1, min_q, K.codes.as<uint64_t>(),
                                              K.valid.as<uint32_t>(), K.word_read.as<uint32_t>(), st);
                }
                RB_HIP(hipGetLastError());
                RB_HIP(hipEventRecord(K.ev[p], st));
                if (bytes) t_enq += now() - tp1;
                { std::lock_guard<std::mutex> lk(K.fm); K.enqueued = p + 1; }
                K.fcv.notify_all();
                ra = rb;
            }
            if (tdbg) fprintf(stderr, "[rb] add_reads feeder: %zu pieces in %.1f ms: registering %.1f ms, enqueueing %.1f ms\n", K.rend.size(), now() - t_f0, t_pin, t_enq);
        });
        { std::lock_guard<std::mutex> lk(K.fm); K.feeder_rc = rc; if (rc != RB_OK) K.feeder_err = rb_last_error(); K.enqueued = K.rend.size(); }   // (an error wakes every waiter)
        K.fcv.notify_all();
    });
}
}  // namespace

// rb_graph_add_reads over host ASCII reads of more than one piece: FastqToGraphWorker's loop (R/RNABloom.java:526-643) as ONE insert.  The device batch
// is sized for the whole call; lengths and word offsets are computed on the GPU from the caller's base offsets (the offsets come back into pinned
// memory for the sub-batch plan), bases and qualities follow piece by piece through two staging buffers and the 2-bit encode kernel, an event
// behind each piece; add_range makes its producer stream wait for the pieces a sub-batch's words lie in (rb_graph::await_words) — one pipeline over
// the whole call with the upload beside it, where rounds 1-5 made one insert call per 256 M-base chunk (no overlap between a chunk's stages: 276 ms
// of insert for 50 M reads that take 144 ms as one batch).  Same filters as the chunked path and as rb_graph_add_batch of the same reads.
void rb::add_reads_streamed(rb_graph *g, const char *seq, const char *qual, const int64_t *offsets, int64_t n_reads, int min_q, int64_t piece_bases,
                            unsigned flags, rb_add_stats *stats) {
    RB_REQUIRE(min_q >= 0 && min_q < 94, "rb_batch_create_ascii: min_base_qual out of range");
    rb_graph::PackedIngest *K = nullptr;
    struct Done { rb_graph *g; rb_graph::PackedIngest **K; ~Done() {
        g->await_words = nullptr;
        std::lock_guard<std::mutex> lk(g->pk_mutex);
        if (*K) ingest_end(g, **K);          // (also on failure: the caller's arrays are free again when the call returns)
        g->pk_busy = -1;
    } } done{g, &K};
    RB_HIP(hipSetDevice(g->p.device));
    {
        std::lock_guard<std::mutex> lk(g->pk_mutex);
        const int q = !g->pk[0].inflight ? 0 : !g->pk[1].inflight ? 1 : 0;
        if (g->pk[q].inflight) ingest_end(g, g->pk[q]);
        K = &g->pk[q]; g->pk_busy = q;
        ingest_begin_ascii(g, *K, seq, qual, offsets, n_reads, min_q, piece_bases);
    }
    const size_t nr = (size_t)n_reads;
    const bool tdbg = getenv("RB_HOST_TIMING") != nullptr;
    auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
    const double t_m0 = now();
    RB_HIP(hipEventSynchronize(K->ev_woff));
    const double t_m1 = now();
    RB_REQUIRE(!K->h_stats[6], "rb_batch_create_ascii: a read has an invalid length (offsets that run backwards, or 2^30 bases and more)");
    K->wend.resize(K->rend.size());
    for (size_t p = 0; p < K->rend.size(); ++p) K->wend[p] = (int64_t)K->h_woff[(size_t)K->rend[p]];
    rb_batch b;
    b.device = g->p.device; b.n_reads = n_reads; b.n_words = (int64_t)K->h_woff[nr];
    b.max_len = K->h_stats[0];
    b.wpr_uniform = K->h_stats[1] == K->h_stats[2] ? K->h_stats[1] : 0u;
    b.n_bases = (int64_t)(((uint64_t)K->h_stats[5] << 32) | K->h_stats[4]);
    b.codes = K->codes.as<uint64_t>(); b.valid = K->valid.as<uint32_t>(); b.word_read = K->word_read.as<uint32_t>();
    b.woff = K->woff.as<uint32_t>(); b.len = K->len.as<uint32_t>(); b.rnz = nullptr;
    b.device_bytes = (size_t)std::max<int64_t>(b.n_words, 1) * 16 + (nr + 1) * 4 + nr * 4;
    b.h_woff.borrow(K->h_woff, nr + 1);
    size_t waited = 0;
    const std::vector<int64_t> &wend = K->wend;
    rb_graph::PackedIngest *Kc = K;
    g->await_words = [&waited, &wend, Kc](int64_t w_end, hipStream_t on) {
        size_t need = 0;
        while (need < wend.size() && (need == 0 ? 0 : wend[need - 1]) < w_end) ++need;
        if (need > waited) {
            {   // the feeder must have RECORDED the piece's event before a stream can be told to wait for it
                std::unique_lock<std::mutex> lk(Kc->fm);
                Kc->fcv.wait(lk, [&] { return Kc->enqueued >= need || Kc->feeder_rc != RB_OK; });
                if (Kc->feeder_rc != RB_OK) { set_error("%s", Kc->feeder_err.c_str()); throw HipError{Kc->feeder_rc}; }
            }
            RB_HIP(hipStreamWaitEvent(on, Kc->ev[need - 1], 0));
            waited = need;
        }
    };
    add_range(g, &b, 0, n_reads, flags, stats);
    if (tdbg) fprintf(stderr, "[rb] add_reads streamed: offsets back after %.1f ms, insert %.1f ms\n", t_m1 - t_m0, now() - t_m1);
    b.codes = nullptr; b.valid = nullptr; b.word_read = nullptr; b.woff = nullptr; b.len = nullptr;
    {   // a feeder that failed after the last piece the insert asked for still failed the call
        std::unique_lock<std::mutex> lk(K->fm);
        K->fcv.wait(lk, [&] { return K->enqueued >= K->rend.size() || K->feeder_rc != RB_OK; });
        if (K->feeder_rc != RB_OK) { set_error("%s", K->feeder_err.c_str()); throw HipError{K->feeder_rc}; }
    }
}

extern "C" {

// Start the upload of a packed batch that a later rb_graph_add_packed call WITH THE SAME ARRAYS AND SIZES will insert, and return at once: the
// reader side of the reference's stage 1 fetches the next file while the workers still insert this one (R/RNABloom.java:7123-7188).  The arrays
// must stay as they are until that call returns.  One batch may be ahead; a second prefetch, or an add of other arrays, first finishes it.
int rb_graph_prefetch_packed(rb_graph *g, const uint64_t *codes, const uint32_t *valid, const uint32_t *len, int64_t n_reads, int64_t n_words, int64_t piece_reads) {
    if (!g) { set_error("rb_graph_prefetch_packed: null graph"); return RB_ERR_INVALID; }
    return guarded([&] {
        RB_REQUIRE(n_reads > 0 && n_words > 0 && piece_reads >= 0 && len && codes && valid, "rb_graph_prefetch_packed: bad argument");
        RB_REQUIRE(n_words < 0xFFFFFFF0ll && n_reads < 0xFFFFFFF0ll, "rb_graph_prefetch_packed: batch too large (> 2^32 words)");
        std::lock_guard<std::mutex> lk(g->pk_mutex);
        RB_HIP(hipSetDevice(g->p.device));
        rb_graph::PackedIngest *K = !g->pk[0].inflight ? &g->pk[0] : !g->pk[1].inflight ? &g->pk[1] : nullptr;
        if (!K) { K = &g->pk[g->pk_busy == 0 ? 1 : 0]; ingest_end(g, *K); }     // both taken: the one no insert is reading gives way
        ingest_begin(g, *K, codes, valid, len, n_reads, n_words, piece_reads);
    });
}

// FastqToGraphWorker.run over reads that are already packed in host memory, as ONE insert call: the device batch is sized for all n_reads reads,
// the lengths go up first (offsets and word owners are computed on the GPU, the offsets come back into pinned memory for the sub-batch plan),
// codes and valid follow on the handle's copy stream in pieces — piece_reads reads' worth each (0: 2^22 words, doubling up to 2^25: the first
// piece is all an insert has to wait for) — with an event behind every piece; add_range makes its producer stream wait for the pieces a
// sub-batch's words lie in (rb_graph::await_words) and otherwise runs exactly as it does on a resident batch: one pipeline over the whole file,
// the upload beside it.  A batch that rb_graph_prefetch_packed already started (same arrays, same sizes) is picked up where it is.
// Same filters as rb_graph_add_batch of the same reads.
int rb_graph_add_packed(rb_graph *g, const uint64_t *codes, const uint32_t *valid, const uint32_t *len, int64_t n_reads, int64_t n_words,
                        int64_t piece_reads, unsigned flags, rb_add_stats *stats) {
    if (stats) memset(stats, 0, sizeof *stats);
    if (!g) { set_error("rb_graph_add_packed: null graph"); return RB_ERR_INVALID; }
    WriteLock wl(g->rw);
    rb_graph::PackedIngest *K = nullptr;
    struct Done { rb_graph *g; rb_graph::PackedIngest **K; ~Done() {
        g->await_words = nullptr;
        std::lock_guard<std::mutex> lk(g->pk_mutex);
        if (*K) ingest_end(g, **K);          // (also on failure: the caller's arrays are free again when the call returns)
        g->pk_busy = -1;
    } } done{g, &K};
    return guarded([&] {
        RB_REQUIRE(n_reads >= 0 && n_words >= 0 && piece_reads >= 0 && (n_reads == 0 || len) && (n_words == 0 || (codes && valid)), "rb_graph_add_packed: bad argument");
        RB_REQUIRE(n_words < 0xFFFFFFF0ll && n_reads < 0xFFFFFFF0ll, "rb_graph_add_packed: batch too large (> 2^32 words)");
        if (!n_reads) return;
        RB_HIP(hipSetDevice(g->p.device));
        {
            std::lock_guard<std::mutex> lk(g->pk_mutex);
            for (int q = 0; q < 2 && !K; ++q)
                if (g->pk[q].inflight && g->pk[q].src == codes && g->pk[q].n_reads == n_reads && g->pk[q].n_words == n_words) { K = &g->pk[q]; g->pk_busy = q; }
            if (!K) {
                const int q = !g->pk[0].inflight ? 0 : !g->pk[1].inflight ? 1 : 0;
                if (g->pk[q].inflight) ingest_end(g, g->pk[q]);
                K = &g->pk[q]; g->pk_busy = q;
                ingest_begin(g, *K, codes, valid, len, n_reads, n_words, piece_reads);
            }
        }
        const size_t nr = (size_t)n_reads;
        RB_HIP(hipEventSynchronize(K->ev_woff));                        // lengths up, offsets back: ~7 ms per 10^8 reads, the pieces keep going meanwhile
        RB_REQUIRE((int64_t)K->h_woff[nr] == n_words, "rb_graph_add_packed: the lengths of %lld reads add up to %u words, %lld were passed", (long long)n_reads, K->h_woff[nr], (long long)n_words);
        rb_batch b;
        b.device = g->p.device; b.n_reads = n_reads; b.n_words = n_words;
        b.max_len = K->h_stats[0];
        b.wpr_uniform = K->h_stats[1] == K->h_stats[2] ? K->h_stats[1] : 0u;
        b.n_bases = (int64_t)(((uint64_t)K->h_stats[5] << 32) | K->h_stats[4]);
        b.codes = K->codes.as<uint64_t>(); b.valid = K->valid.as<uint32_t>(); b.word_read = K->word_read.as<uint32_t>();
        b.woff = K->woff.as<uint32_t>(); b.len = K->len.as<uint32_t>(); b.rnz = nullptr;
        b.device_bytes = (size_t)std::max<int64_t>(n_words, 1) * 16 + (nr + 1) * 4 + nr * 4;
        b.h_woff.borrow(K->h_woff, nr + 1);
        size_t waited = 0;                                   // the producer stream has been told to wait for pieces [0, waited)
        const std::vector<int64_t> &wend = K->wend;
        rb_graph::PackedIngest *Kc = K;
        g->await_words = [&waited, &wend, Kc](int64_t w_end, hipStream_t on) {
            size_t need = 0;
            while (need < wend.size() && (need == 0 ? 0 : wend[need - 1]) < w_end) ++need;      // pieces 0 .. need - 1 cover words [0, w_end)
            // (the producer stream asks in increasing order; waiting for the LAST piece it needs is enough: the copy stream is in order)
            if (need > waited) { RB_HIP(hipStreamWaitEvent(on, Kc->ev[need - 1], 0)); waited = need; }
        };
        add_range(g, &b, 0, n_reads, flags, stats);
        b.codes = nullptr; b.valid = nullptr; b.word_read = nullptr; b.woff = nullptr; b.len = nullptr;
    });
}

}  // extern "C"
