// rb_capi.hip — the C ABI around the insert pipeline: life cycle of a graph handle, the stage-1 insert entry points (batches, host
// reads, FASTQ / FASTA text and files), array-of-hashes ops, and the filters as objects (sizes, popcounts, FPRs, export / import, the
// digest).  The pipeline itself is rb_graph.hip, queries and traversals rb_query.hip.  (Split out of rb_graph.hip in round 5.)
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

namespace {

// ---- popcounts (UnsafeByteBuffer.bitPopCount :131-150 / popCount :121-129) ----
__global__ void k_popcount_bits(const uint32_t *__restrict__ w, size_t n_words, unsigned long long *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n_words; i += (size_t)gridDim.x * blockDim.x) c += __popc(w[i]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
__global__ void k_count_nonzero_bytes(const uint32_t *__restrict__ w, size_t n_words, unsigned long long *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = w[i];
        c += ((x & 0xFFu) != 0) + ((x & 0xFF00u) != 0) + ((x & 0xFF0000u) != 0) + ((x & 0xFF000000u) != 0);
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
// order-independent 64-bit digest of a filter's bytes: the wrapping sum over the non-zero 32-bit words of
// mix(global word number, word).  Sums of the digests of index ranges = digest of the whole filter, so the
// shards of a distributed filter and a single-GPU filter compare without exporting 150 GB (rb_filter_fold).
__device__ __forceinline__ uint64_t fold_mix(uint64_t gw, uint32_t x) {
    uint64_t z = gw * 0x9E3779B97F4A7C15ull + (uint64_t)x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void k_fold_words(const uint32_t *__restrict__ w, size_t n_words, uint64_t gw0, unsigned long long *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t x = w[i];
        if (x) c += fold_mix(gw0 + i, x);
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
// CountingBloomFilter.getBloomFilter(minCov) R/bloom/CountingBloomFilter.java:328-338: bit i of the new filter is set iff
// MiniFloat.toFloat(counts[i]) >= minCov.  One thread per 32 counters = one output word.
__global__ void k_cbf_to_bits(const uint8_t *__restrict__ cbf, int64_t n, float min_cov, uint32_t *__restrict__ bits) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w * 32 >= n) return;
    uint32_t out = 0;
    for (int b = 0; b < 32; ++b) {
        const int64_t i = w * 32 + b;
        if (i < n && minifloat_to_float((uint32_t)cbf[i] & 0x7Fu) >= min_cov) out |= 1u << b;
    }
    bits[w] = out;
}
// CountingBloomFilter.incrementAndGet(long[]) R/bloom/CountingBloomFilter.java:196-222, one call after the other in array
// order (the subsampler's loops are sequential by nature: every result decides what happens next).  One lane walks the
// array; op i draws from op ordinal ordinal0 + i, position 0 (as a per-hash API call does).
__global__ void k_increment_and_get(FilterView fv, const uint64_t *__restrict__ h0, size_t n, float *__restrict__ out) {
    if (blockIdx.x || threadIdx.x) return;
    for (size_t i = 0; i < n; ++i) {
        uint64_t idx[RB_MAX_HASH];
        uint32_t mn = 0;
        for (int j = 0; j < fv.cbf_h; ++j) {
            idx[j] = index_of(multi_hash(h0[i], (uint32_t)j, fv.kmul), fv.cbf_mod);
            const uint32_t c = fv.cbf[idx[j]];
            mn = (j == 0 || c < mn) ? c : mn;
        }
        const uint32_t up = minifloat_inc(mn, rng31(fv.seed, fv.ordinal0 + (uint64_t)i, 0u));
        if (up != mn)
            for (int j = 0; j < fv.cbf_h; ++j) if (fv.cbf[idx[j]] == mn) fv.cbf[idx[j]] = (uint8_t)up;
        out[i] = minifloat_to_float(up);
    }
}
__global__ void k_iota(uint32_t *v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

}  // namespace

// ----------------------------------------------------------------------- C ABI ----
extern "C" {

int rb_version(void) { return 1; }
const char *rb_last_error(void) { return rb::last_error_text(); }

int rb_graph_create(const rb_graph_params *p, rb_graph **out) {
    rb_graph *g = nullptr;
    int rc = guarded([&] {
        RB_REQUIRE(p && out, "rb_graph_create: null argument");
        RB_REQUIRE(p->k >= 1 && p->k <= RB_MAX_K, "rb_graph_create: k=%d out of range [1,%d]", p->k, RB_MAX_K);
        RB_REQUIRE(p->dbgbf_bits > 0 && p->cbf_bytes > 0, "rb_graph_create: filter sizes must be positive");
        RB_REQUIRE(p->dbgbf_num_hash >= 1 && p->dbgbf_num_hash <= RB_MAX_HASH && p->cbf_num_hash >= 1 &&
                   p->cbf_num_hash <= RB_MAX_HASH, "rb_graph_create: numHash out of range [1,%d]", RB_MAX_HASH);
        if (p->use_read_paired_kmers)
            RB_REQUIRE(p->pkbf_bits > 0 && p->pkbf_num_hash >= 1 && p->pkbf_num_hash <= RB_MAX_HASH,
                       "rb_graph_create: pair filter parameters invalid");
        int ndev = 0;
        RB_HIP(hipGetDeviceCount(&ndev));
        RB_REQUIRE(p->device >= 0 && p->device < ndev, "rb_graph_create: device %d not present (%d devices)", p->device, ndev);
        RB_HIP(hipSetDevice(p->device));
        g = new rb_graph();
        g->p = *p;
        g->k = p->k;
        g->stranded = p->stranded != 0;
        g->H = std::max(p->dbgbf_num_hash, p->cbf_num_hash);
        g->max_batch_kmers = p->max_batch_kmers > 0 ? p->max_batch_kmers : ((int64_t)1 << 30);
        RB_REQUIRE(p->group_bits >= 0 && p->group_bits <= 64, "rb_graph_create: group_bits out of range [0,64]");
        if (p->group_bits) g->sort_begin_bit = 64 - p->group_bits;
        if (const char *e = getenv("RB_LIGHT_OPS")) g->light_ops = (uint32_t)std::max(1, atoi(e));
        if (const char *e = getenv("RB_SMALL_COMPONENT_OPS")) g->small_ops = (uint32_t)std::max(1, atoi(e));
        if (const char *e = getenv("RB_SORT_BEGIN_BIT")) g->sort_begin_bit = std::max(0, std::min(63, atoi(e)));
        RB_REQUIRE(g->max_batch_kmers <= ((int64_t)1 << 31), "rb_graph_create: max_batch_kmers above 2^31");
        if (const char *e = getenv("RB_CONSUMER_PRIORITY")) RB_HIP(hipStreamCreateWithPriority(&g->stream, hipStreamNonBlocking, atoi(e)));
        else RB_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
        {   // the producer (hash + sort of the next sub-batch) is the critical path: give it priority
            int lo_p = 0, hi_p = 0;
            RB_HIP(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
            int pr = hi_p;
            if (const char *e = getenv("RB_PRODUCER_PRIORITY")) pr = atoi(e);
            RB_HIP(hipStreamCreateWithPriority(&g->stream2, hipStreamNonBlocking, pr));
            RB_HIP(hipStreamCreateWithPriority(&g->stream3, hipStreamNonBlocking, pr));
        }
        // the packed ingest's copy stream (rb_packed.hip) is created HERE, not at the first upload: HIP hands its streams to a few hardware queues in the
        // order they are made, and a copy stream made after a query context's stream (rb_filter_fold before the first rb_graph_add_packed was enough)
        // shared a queue with the insert's kernels — its pieces queued behind them, 10 ms of every 300 ms step (tools/host_gap.py, HISTORY "Round 6")
        RB_HIP(hipStreamCreateWithFlags(&g->pk_stream, hipStreamNonBlocking));
        RB_HIP(hipEventCreate(&g->ev0));
        RB_HIP(hipEventCreate(&g->ev1));
        RB_HIP(hipEventCreateWithFlags(&g->ev2, hipEventDisableTiming));
        RB_HIP(hipEventCreateWithFlags(&g->ev3, hipEventDisableTiming));
        alloc_bits(g->dbg, p->dbgbf_bits, p->dbgbf_num_hash, 0, p->dbgbf_bits);
        g->cbf_size = p->cbf_bytes; g->cbf_lo = 0; g->cbf_hi = p->cbf_bytes;
        g->cbf_alloc = (((size_t)p->cbf_bytes + 3) / 4 + 1) * 4;
        g->cbf_h = p->cbf_num_hash;
        g->cbf_mod = make_mod((uint64_t)p->cbf_bytes);
        g->cbf = static_cast<uint8_t *>(rb::alloc_best_placed(g->cbf_alloc, "cbf"));
        if (p->use_read_paired_kmers) { alloc_bits(g->rpk, p->pkbf_bits, p->pkbf_num_hash, 0, p->pkbf_bits); alloc_pair_seen(g->rpk); }
        {   // no-op prefilter cache: one 8-byte entry per ~64 counters, 2^16..2^28 entries (8-way buckets fill well: 2^27 entries hold the 64 M hot k-mers of config 2 as completely as 2^28)
            const char *e = getenv("RB_NPF");
            uint32_t l2 = log2_ceil((uint64_t)std::max<int64_t>(p->cbf_bytes / 64, 1));
            l2 = std::max(16u, std::min(28u, l2));
            if (e) l2 = (uint32_t)atoi(e);
            if (l2 >= 8 && l2 <= 30) {
                g->npf.reserve(sizeof(uint64_t) << l2);
                RB_HIP(hipMemset(g->npf.p, 0, sizeof(uint64_t) << l2));
                g->npf_log2 = l2;
            }
        }
        {   // minimizer-bucketed variant for the k <= 31 insert path: 16 slots (128 B) per bucket + 1 overflow bit
            const char *e = getenv("RB_MPF");
            uint32_t lb = log2_ceil((uint64_t)std::max<int64_t>(p->cbf_bytes / 256, 1));
            lb = std::max(12u, std::min(25u, lb));
            if (e) lb = (uint32_t)atoi(e);
            if (lb >= 8 && lb <= 28 && p->k <= RB_MPF_WIDE_MAX_K && p->k >= 8) {       // (32 <= k <= 63: used by the read-per-lane prefilter only, add_range decides per batch)
                g->mpf.reserve((size_t)128 << lb);
                RB_HIP(hipMemset(g->mpf.p, 0, (size_t)128 << lb));
                g->mpf_log2b = lb;
                g->mpf_m = (uint32_t)std::min(getenv("RB_MPF_M") ? std::max(4, std::min(16, atoi(getenv("RB_MPF_M")))) : 16, p->k);
            }
        }
        RB_HIP(hipDeviceSynchronize());
        *out = g;
    });
    if (rc != RB_OK && g) { rb_graph_destroy(g); }
    return rc;
}

int rb_graph_destroy(rb_graph *g) {
    if (!g) return RB_OK;
    (void)hipSetDevice(g->p.device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->stream2) (void)hipStreamSynchronize(g->stream2);
    if (g->stream3) (void)hipStreamSynchronize(g->stream3);
    rb::trav_free(g);
    rb::shard_free(g);
    free_bits(g->dbg); free_bits(g->rpk); free_bits(g->fpk);
    if (g->cbf) (void)hipFree(g->cbf);
    DevBuf *bufs[] = {&g->chunk_cnt, &g->chunk_off, &g->keys0,  &g->vals0,   
                       &g->status, &g->nops, &g->temp, &g->ftable, &g->ctable, &g->heavy, &g->confk,
                      &g->conf_sizes, &g->conf_off, &g->opk0, &g->opk1, &g->opv0, &g->opv1, &g->label, &g->kk0, &g->kk1, &g->biglist, &g->cvals, &g->foreign,  &g->devctr, &g->cwriters, &g->cshared, &g->comm_keep, &g->comm_dreply, &g->comm_creply, &g->qbuf0,
                      &g->qbuf1, &g->qbuf2, &g->qbuf3};
    for (auto *b : bufs) b->release();
    for (rb_query_ctx *c : g->qfree) {
        c->b0.release(); c->b1.release(); c->b2.release(); c->b3.release();
        if (c->st) (void)hipStreamDestroy(c->st);
        delete c;
    }
    g->qfree.clear();
    if (g->pk_stream) (void)hipStreamSynchronize(g->pk_stream);
    for (auto &K : g->pk) {
        for (DevBuf *d : {&K.codes, &K.valid, &K.word_read, &K.woff, &K.len, &K.wc, &K.temp, &K.stats}) d->release();
        if (K.h_woff) (void)hipHostFree(K.h_woff);
        if (K.h_stats) (void)hipHostFree(K.h_stats);
        for (auto e : K.ev) (void)hipEventDestroy(e);
        if (K.ev_woff) (void)hipEventDestroy(K.ev_woff);
        for (void *p : K.pins) (void)hipHostUnregister(p);
    }
    if (g->pk_stream) (void)hipStreamDestroy(g->pk_stream);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->ev2) (void)hipEventDestroy(g->ev2);
    if (g->ev3) (void)hipEventDestroy(g->ev3);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    if (g->stream2) (void)hipStreamDestroy(g->stream2);
    if (g->stream3) (void)hipStreamDestroy(g->stream3);
    for (auto e : g->prof_pool) (void)hipEventDestroy(e);
    for (auto &sl : g->slots) { sl.keys1.release(); sl.valsT.release(); sl.vals1.release(); sl.tz.release(); sl.uniq.release(); sl.counts.release(); sl.starts.release(); }
    g->temp2.release(); g->devctr2.release(); g->pairs_ctr.release(); g->npf.release(); g->mpf.release(); g->chunk_mask.release(); g->npf_tot.release(); g->wstate.release();
    delete g;
    return RB_OK;
}

int rb_graph_clear(rb_graph *g, unsigned which_mask) {
    return guarded([&] {
        RB_REQUIRE(g, "rb_graph_clear: null graph");
        WriteLock wl(g->rw);
        RB_HIP(hipSetDevice(g->p.device));
        if ((which_mask & 1u) && g->dbg.bits) fast_zero(g->dbg.bits, g->dbg.alloc, g->stream);
        if ((which_mask & 2u) && g->cbf) fast_zero(g->cbf, g->cbf_alloc, g->stream);
        if ((which_mask & 4u) && g->rpk.bits) { fast_zero(g->rpk.bits, g->rpk.alloc, g->stream); seen_reset(g->rpk, g->stream); }
        if ((which_mask & 4u) && g->shard) rb::shard_clear_pairs_acc(g);
        if ((which_mask & 8u) && g->fpk.bits) fast_zero(g->fpk.bits, g->fpk.alloc, g->stream);
        if ((which_mask & 3u) && g->npf_log2) fast_zero(g->npf.p, sizeof(uint64_t) << g->npf_log2, g->stream);   // cache entries speak about dbgbf + cbf
        if ((which_mask & 3u) && g->mpf_log2b) fast_zero(g->mpf.p, (size_t)128 << g->mpf_log2b, g->stream);
        if ((which_mask & 3u) == 3u) { g->ordinal = 0; g->pf_streak = 0; g->pf_skip_left = 0; g->last_present_frac = 0.0f; }
        RB_HIP(hipStreamSynchronize(g->stream));
    });
}

int rb_graph_set_read_paired_kmer_distance(rb_graph *g, int d) {
    if (!g) { set_error("null graph"); return RB_ERR_INVALID; }
    g->read_d = d; return RB_OK;
}
int rb_graph_set_frag_paired_kmer_distance(rb_graph *g, int d) {
    if (!g) { set_error("null graph"); return RB_ERR_INVALID; }
    g->frag_d = d; return RB_OK;
}
int rb_graph_init_fragment_pairs(rb_graph *g, int64_t pkbf_bits, int pkbf_num_hash) {
    return guarded([&] {
        RB_REQUIRE(g && pkbf_bits > 0 && pkbf_num_hash >= 1 && pkbf_num_hash <= RB_MAX_HASH, "rb_graph_init_fragment_pairs: bad argument");
        WriteLock wl(g->rw);
        RB_HIP(hipSetDevice(g->p.device));
        if (!g->fpk.bits) alloc_bits(g->fpk, pkbf_bits, pkbf_num_hash, 0, pkbf_bits);   // :352-359: create once, else empty()
        else { RB_HIP(hipMemset(g->fpk.bits, 0, g->fpk.alloc)); RB_HIP(hipDeviceSynchronize()); }
    });
}
int rb_graph_get_op_ordinal(rb_graph *g, uint64_t *out) {
    if (!g || !out) { set_error("null argument"); return RB_ERR_INVALID; }
    *out = g->ordinal; return RB_OK;
}
int rb_graph_set_op_ordinal(rb_graph *g, uint64_t v) {
    if (!g) { set_error("null graph"); return RB_ERR_INVALID; }
    g->ordinal = v; return RB_OK;
}

int rb_graph_add_batch_range(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, unsigned flags, rb_add_stats *stats) {
    return guarded([&] {
        RB_REQUIRE(g && b, "rb_graph_add_batch: null argument");
        WriteLock wl(g->rw);
        if (stats) memset(stats, 0, sizeof *stats);
        add_range(g, b, first, n, flags, stats);
    });
}
// PairedKmersToGraphWorker (R/RNABloom.java:436-524): paired k-mers only, optionally only where both k-mers are in dbgbf
int rb_graph_add_pairs(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, int which, unsigned flags, rb_add_stats *stats) {
    return guarded([&] {
        RB_REQUIRE(g && b, "rb_graph_add_pairs: null argument");
        WriteLock wl(g->rw);
        RB_REQUIRE(!g->shard, "rb_graph_add_pairs: not available on a shard handle");
        RB_REQUIRE(b->device == g->p.device, "batch lives on device %d, graph on %d", b->device, g->p.device);
        RB_REQUIRE(first >= 0 && n >= 0 && first + n <= b->n_reads, "rb_graph_add_pairs: bad read range");
        RB_REQUIRE(which == RB_RPKBF || which == RB_FPKBF, "rb_graph_add_pairs: which must be RB_RPKBF or RB_FPKBF");
        RB_REQUIRE((flags & ~(RB_ADD_REVCOMP | RB_ADD_PAIRS_IF_PRESENT)) == 0u, "rb_graph_add_pairs: unknown flag");
        BitFilter &f = which == RB_RPKBF ? g->rpk : g->fpk;
        const int dist = which == RB_RPKBF ? g->read_d : g->frag_d;
        if (!f.bits || dist <= 0) { set_error("rb_graph_add_pairs: pair filter %d not initialised or its k-mer distance not set", which); throw HipError{RB_ERR_STATE}; }
        if (stats) memset(stats, 0, sizeof *stats);
        if (!n) return;
        RB_HIP(hipSetDevice(g->p.device));
        const int64_t w0 = (int64_t)b->h_woff[(size_t)first], nw = (int64_t)b->h_woff[(size_t)(first + n)] - w0;
        const int mode_hash = g->stranded ? ((flags & RB_ADD_REVCOMP) ? 2 : 0) : 1;
        g->devctr.reserve(DEVCTR_BYTES);
        unsigned long long *pc = reinterpret_cast<unsigned long long *>(g->devctr.as<uint32_t>() + 12);
        RB_HIP(hipMemsetAsync(pc, 0, 8, g->stream));
        if (nw > 0) launch_pairs_reads(g, b, w0, nw, mode_hash, f, dist, 0u, (flags & RB_ADD_PAIRS_IF_PRESENT) != 0, nullptr, nullptr, pc, g->stream);
        RB_HIP(hipGetLastError());
        unsigned long long np = 0;
        RB_HIP(hipMemcpyAsync(&np, pc, 8, hipMemcpyDeviceToHost, g->stream));
        RB_HIP(hipStreamSynchronize(g->stream));
        if (stats) { stats->reads = n; stats->pairs = (int64_t)np; }
    });
}

// FragmentsToGraphWorker (R/RNABloom.java:1463-1539): every k-mer of a fragment into dbgbf only; with loadPairedKmers also
// its read-paired k-mers into rpkbf and — where those could start — its fragment-paired k-mers into fpkbf.  All pure ORs.
int rb_graph_add_fragments(rb_graph *g, const rb_batch *b, int64_t first, int64_t n, int load_paired_kmers, rb_add_stats *stats) {
    return guarded([&] {
        RB_REQUIRE(g && b, "rb_graph_add_fragments: null argument");
        WriteLock wl(g->rw);
        RB_REQUIRE(!g->shard, "rb_graph_add_fragments: not available on a shard handle");
        RB_REQUIRE(b->device == g->p.device, "batch lives on device %d, graph on %d", b->device, g->p.device);
        RB_REQUIRE(first >= 0 && n >= 0 && first + n <= b->n_reads, "rb_graph_add_fragments: bad read range");
        if (load_paired_kmers && !(g->rpk.bits && g->read_d > 0 && g->fpk.bits && g->frag_d > 0)) {
            set_error("rb_graph_add_fragments: loadPairedKmers needs rpkbf + fpkbf and both paired k-mer distances"); throw HipError{RB_ERR_STATE};
        }
        if (stats) memset(stats, 0, sizeof *stats);
        if (!n) return;
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        const int64_t w0 = (int64_t)b->h_woff[(size_t)first], nw = (int64_t)b->h_woff[(size_t)(first + n)] - w0;
        if (nw <= 0) return;
        const int mode_hash = g->stranded ? 0 : 1;               // graph.getHashIterator: forward or canonical
        // k-mers -> dbgbf (addDbgOnly, order independent): count, scan, hash, set bits
        g->chunk_cnt.reserve(((size_t)nw + 1) * 4); g->chunk_off.reserve(((size_t)nw + 1) * 4);
        g->temp.reserve(scan_temp_bytes((size_t)nw + 1));
        RB_HIP(hipMemsetAsync(g->chunk_cnt.p, 0, ((size_t)nw + 1) * 4, s));
        launch_count_windows(b, w0, nw, g->k, g->chunk_cnt.as<uint32_t>(), s);
        exclusive_scan_u32(g->temp.p, g->temp.cap, g->chunk_cnt.as<uint32_t>(), g->chunk_off.as<uint32_t>(), (size_t)nw + 1, s);
        uint32_t total = 0;
        RB_HIP(hipMemcpyAsync(&total, g->chunk_off.as<uint32_t>() + nw, 4, hipMemcpyDeviceToHost, s));
        RB_HIP(hipStreamSynchronize(s));
        if (total) {
            g->keys0.reserve((size_t)total * 8);
            launch_hash_windows(b, w0, nw, g->k, mode_hash, g->chunk_off.as<uint32_t>(), 0, 0, g->keys0.as<uint64_t>(), nullptr, nullptr, nullptr, s);
            hipLaunchKernelGGL(k_bits_add, dim3(blocks_for((int64_t)total)), dim3(TPB), 0, s, g->dbg.bits, g->dbg.mod, g->dbg.num_hash, kmul_of(g->k),
                               g->keys0.as<uint64_t>(), (size_t)total);
        }
        unsigned long long np = 0;
        if (load_paired_kmers) {
            g->devctr.reserve(DEVCTR_BYTES);
            unsigned long long *pc = reinterpret_cast<unsigned long long *>(g->devctr.as<uint32_t>() + 12);
            RB_HIP(hipMemsetAsync(pc, 0, 8, s));
            launch_pairs_reads(g, b, w0, nw, mode_hash, g->rpk, g->read_d, 0u, false, nullptr, nullptr, pc, s);
            launch_pairs_reads(g, b, w0, nw, mode_hash, g->fpk, g->frag_d, (uint32_t)(g->k + g->read_d), false, nullptr, nullptr, pc, s);
            RB_HIP(hipMemcpyAsync(&np, pc, 8, hipMemcpyDeviceToHost, s));
        }
        RB_HIP(hipGetLastError());
        RB_HIP(hipStreamSynchronize(s));
        // the prefilter cache speaks about dbgbf + cbf: new dbgbf bits do not falsify an entry (entries only ever
        // understate), so it stays as it is
        if (stats) { stats->reads = n; stats->kmers = total; stats->pairs = (int64_t)np; }
    });
}

int rb_graph_add_batch(rb_graph *g, const rb_batch *b, unsigned flags, rb_add_stats *stats) {
    if (!b) { set_error("rb_graph_add_batch: null batch"); return RB_ERR_INVALID; }
    return rb_graph_add_batch_range(g, b, 0, b->n_reads, flags, stats);
}
// Host ASCII reads: the caller's buffers are pinned for the duration of the call (hipHostRegister costs ~8 ms per
// GB and lets the DMA engines run at link speed, ~57 GB/s measured, instead of ~15 GB/s from pageable pages),
// cut into chunks of <= 256 M bases, and chunk c+1 is uploaded + 2-bit encoded on its own stream while the
// insert pipeline works on chunk c.
int rb_graph_add_reads(rb_graph *g, const char *seq, const char *qual, const int64_t *offsets, int64_t n_reads,
                       int min_base_qual, unsigned flags, rb_add_stats *stats) {
    if (!g) { set_error("rb_graph_add_reads: null graph"); return RB_ERR_INVALID; }
    if (!offsets || n_reads < 0) { set_error("rb_graph_add_reads: null argument"); return RB_ERR_INVALID; }
    rb::AsciiUpload up;
    if (!getenv("RB_NO_INGEST_POOL")) up.pool = &g->ingest_pool;
    hipStream_t st = nullptr;
    bool own_stream = false;
    const char *pin_seq = nullptr, *pin_qual = nullptr;
    WriteLock wl(g->rw);
    int rc = guarded([&] {
        RB_HIP(hipSetDevice(g->p.device));
        const int64_t base0 = n_reads ? offsets[0] : 0, nbases = n_reads ? offsets[n_reads] - base0 : 0;
        {   // more than one piece: one insert over a batch that is still being uploaded and encoded (rb_packed.hip); RB_ASCII_PIECE=<bases> sets the piece
            // (and with it the size from which a call is streamed), RB_ASCII_CHUNKED=1 keeps the chunk-by-chunk path below
            const int64_t piece = getenv("RB_ASCII_PIECE") ? std::max<int64_t>(64, atoll(getenv("RB_ASCII_PIECE"))) : ((int64_t)256 << 20);
            if (seq && n_reads > 1 && nbases > piece && !getenv("RB_ASCII_CHUNKED") && !g->shard) {
                rb::add_reads_streamed(g, seq, qual, offsets, n_reads, min_base_qual, piece, flags, stats);
                return;
            }
        }
        const bool tdbg0 = getenv("RB_HOST_TIMING") != nullptr;
        auto now0 = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
        const double t_pin0 = now0();
        if (nbases > (16 << 20) && !getenv("RB_NO_PIN")) {        // pinning is best effort (foreign mappings may refuse)
            if (seq && hipHostRegister(const_cast<char *>(seq + base0), (size_t)nbases, hipHostRegisterDefault) == hipSuccess) pin_seq = seq + base0;
            if (qual && hipHostRegister(const_cast<char *>(qual + base0), (size_t)nbases, hipHostRegisterDefault) == hipSuccess) pin_qual = qual + base0;
            (void)hipGetLastError();
        }
        if (tdbg0) fprintf(stderr, "[rb] add_reads: pinning %.1f ms (seq %s, qual %s)\n", now0() - t_pin0, pin_seq ? "registered" : "not registered", pin_qual ? "registered" : "not registered");
        st = getenv("RB_INGEST_OWN_STREAM") ? nullptr : g->pk_stream;      // the handle's copy stream (made with the graph: a hardware queue of its own, rb_graph_create)
        if (!st) { RB_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); own_stream = true; }
        const int64_t chunk_bases = (int64_t)256 << 20;
        auto chunk_end = [&](int64_t a) {   // largest e > a with bases(a..e) <= chunk_bases (at least one read)
            int64_t lo = a + 1, hi = n_reads;
            while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (offsets[mid] - offsets[a] <= chunk_bases) lo = mid; else hi = mid - 1; }
            return std::min(lo, n_reads);
        };
        int64_t a = 0, e = n_reads ? chunk_end(0) : 0;
        const double t_b0 = now0();
        int turn = 0;
        if (up.pool) up.hs = &g->ingest_host[turn];
        rb::ascii_batch_begin(up, g->p.device, seq, qual, offsets, a, e - a, min_base_qual, st);
        if (tdbg0) fprintf(stderr, "[rb] add_reads: first chunk begun in %.1f ms\n", now0() - t_b0);
        const bool tdbg = getenv("RB_HOST_TIMING") != nullptr;
        auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
        double t_fin = 0, t_beg = 0, t_add = 0, t_des = 0, t_join = 0;
        for (;;) {
            double t0 = now();
            rb_batch *b = rb::ascii_batch_finish(up);
            double t1 = now(); t_fin += t1 - t0;
            a = e;
            // the next chunk's host-side preparation (offset tables, allocations, enqueueing copies + encode) runs on a
            // helper thread while this thread drives the insert pipeline of the current chunk
            std::thread prep;
            int prep_rc = RB_OK;
            std::string prep_err;
            if (a < n_reads) {
                e = chunk_end(a);
                const int64_t ca = a, cn = e - a;
                turn ^= 1;
                if (up.pool) up.hs = &g->ingest_host[turn];
                prep = std::thread([&, ca, cn] {
                    prep_rc = guarded([&] { rb::ascii_batch_begin(up, g->p.device, seq, qual, offsets, ca, cn, min_base_qual, st); });
                    if (prep_rc != RB_OK) prep_err = rb_last_error();      // the error text is thread-local
                });
            }
            double t2 = now(); t_beg += t2 - t1;
            int add_rc = RB_OK;
            {
                struct G { rb_batch *b; ~G() { rb_batch_destroy(b); } } guard{b};
                add_rc = guarded([&] { add_range(g, b, 0, b->n_reads, flags, stats); });
                t_add += now() - t2;
                t2 = now();
            }
            t_des += now() - t2;
            { const double tj = now(); if (prep.joinable()) prep.join(); t_join += now() - tj; }
            if (add_rc != RB_OK) throw HipError{add_rc};
            if (prep_rc != RB_OK) { set_error("%s", prep_err.c_str()); throw HipError{prep_rc}; }
            if (a >= n_reads) break;
        }
        if (tdbg) fprintf(stderr, "[rb] add_reads: %.1f ms since entry, %.1f ms of them waiting for the next chunk's preparation; ", now0() - t_pin0, t_join);
        if (tdbg) fprintf(stderr, "[rb] add_reads: wait upload %.1f ms, begin next %.1f ms, insert %.1f ms, destroy %.1f ms\n", t_fin, t_beg, t_add, t_des);
    });
    if (rc != RB_OK) rb::ascii_batch_abort(up);
    if (st && !own_stream && rc != RB_OK) (void)hipStreamSynchronize(st);      // (a failed call leaves nothing of its own in flight on the shared stream)
    {
        timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
        if (pin_seq) (void)hipHostUnregister(const_cast<char *>(pin_seq));
        if (pin_qual) (void)hipHostUnregister(const_cast<char *>(pin_qual));
        timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        if (getenv("RB_HOST_TIMING")) fprintf(stderr, "[rb] add_reads: unpinning %.1f ms\n", (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
    }
    if (st && own_stream) (void)hipStreamDestroy(st);
    return rc;
}

int rb_graph_add_fastq(rb_graph *g, const char *text, size_t len, int min_base_qual, unsigned flags, rb_add_stats *stats, int64_t *n_records) {
    if (!g) { set_error("rb_graph_add_fastq: null graph"); return RB_ERR_INVALID; }
    if (!text && len) { set_error("rb_graph_add_fastq: null text"); return RB_ERR_INVALID; }
    hipStream_t st = nullptr;
    bool own_stream = false;
    SlabPin slabs;
    WriteLock wl(g->rw);
    int rc = guarded([&] {
        RB_HIP(hipSetDevice(g->p.device));
        slabs.begin(text, len);                                      // registered slab by slab as the pieces advance (best effort)
        st = getenv("RB_INGEST_OWN_STREAM") ? nullptr : g->pk_stream;      // the handle's copy stream (made with the graph: a hardware queue of its own, rb_graph_create)
        if (!st) { RB_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); own_stream = true; }
        // pieces of 1 GiB of text; a piece starts where the complete records of the one before ended.  The next piece is
        // uploaded and parsed (helper thread, own stream) while the insert pipeline works on the current one.
        const size_t piece_bytes = getenv("RB_FASTQ_PIECE") ? (size_t)std::max(64, atoi(getenv("RB_FASTQ_PIECE"))) : (size_t)1 << 30;
        auto piece = [&](size_t a) {
            const size_t e = std::min(len, a + piece_bytes);
            slabs.pin_to(text + e);
            return rb::fastq_batch_create(g->p.device, text + a, e - a, e == len, min_base_qual, true, st, getenv("RB_NO_INGEST_POOL") ? nullptr : &g->ingest_pool);
        };
        size_t a = 0;
        int64_t recs = 0;
        rb::FastqChunk cur = piece(0);
        for (;;) {
            struct G { rb_batch *b; ~G() { if (b) rb_batch_destroy(b); } } guard{cur.b};
            recs += cur.records;
            const bool last = a + piece_bytes >= len;
            const size_t next = a + cur.consumed;
            RB_REQUIRE(last || cur.consumed > 0, "rb_graph_add_fastq: a record longer than %zu bytes", piece_bytes);
            rb::FastqChunk nxt;
            std::thread prep;
            int prep_rc = RB_OK;
            std::string prep_err;
            if (!last) prep = std::thread([&] {
                prep_rc = guarded([&] { nxt = piece(next); });
                if (prep_rc != RB_OK) prep_err = rb_last_error();                      // the error text is thread-local
            });
            const int add_rc = guarded([&] { add_range(g, cur.b, 0, cur.b->n_reads, flags, stats); });
            if (prep.joinable()) prep.join();
            if (add_rc != RB_OK) { if (nxt.b) rb_batch_destroy(nxt.b); throw HipError{add_rc}; }
            if (prep_rc != RB_OK) { set_error("%s", prep_err.c_str()); throw HipError{prep_rc}; }
            if (last) break;
            a = next; cur = nxt;
        }
        if (n_records) *n_records = recs;
    });
    if (st && !own_stream && rc != RB_OK) (void)hipStreamSynchronize(st);      // (a failed call leaves nothing of its own in flight on the shared stream)
    slabs.end();
    if (st && own_stream) (void)hipStreamDestroy(st);
    return rc;
}

int rb_graph_add_fasta(rb_graph *g, const char *text, size_t len, unsigned flags, rb_add_stats *stats, int64_t *n_records) {
    if (!g) { set_error("rb_graph_add_fasta: null graph"); return RB_ERR_INVALID; }
    if (!text && len) { set_error("rb_graph_add_fasta: null text"); return RB_ERR_INVALID; }
    hipStream_t st = nullptr;
    bool own_stream = false;
    SlabPin slabs;
    WriteLock wl(g->rw);
    int rc = guarded([&] {
        RB_HIP(hipSetDevice(g->p.device));
        slabs.begin(text, len);                                      // registered slab by slab as the pieces advance (best effort)
        st = getenv("RB_INGEST_OWN_STREAM") ? nullptr : g->pk_stream;      // the handle's copy stream (made with the graph: a hardware queue of its own, rb_graph_create)
        if (!st) { RB_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); own_stream = true; }
        // pieces of 1 GiB of text; a piece starts where the complete records of the one before ended.  The next piece is
        // uploaded and parsed (helper thread, own stream) while the insert pipeline works on the current one.
        const size_t piece_bytes = getenv("RB_FASTQ_PIECE") ? (size_t)std::max(64, atoi(getenv("RB_FASTQ_PIECE"))) : (size_t)1 << 30;
        bool ended = false;                                       // FastaReader.next() returned null at an empty header line
        auto piece = [&](size_t a) {
            const size_t e = std::min(len, a + piece_bytes);
            slabs.pin_to(text + e);
            bool end_here = false;
            rb::FastqChunk c = rb::fasta_batch_create(g->p.device, text + a, e - a, e == len, st, &end_here, getenv("RB_NO_INGEST_POOL") ? nullptr : &g->ingest_pool);
            if (end_here) ended = true;
            return c;
        };
        size_t a = 0;
        int64_t recs = 0;
        rb::FastqChunk cur = piece(0);
        for (;;) {
            struct G { rb_batch *b; ~G() { if (b) rb_batch_destroy(b); } } guard{cur.b};
            recs += cur.records;
            const bool last = a + piece_bytes >= len || ended;
            const size_t next = a + cur.consumed;
            RB_REQUIRE(last || cur.consumed > 0, "rb_graph_add_fasta: a record longer than %zu bytes", piece_bytes);
            rb::FastqChunk nxt;
            std::thread prep;
            int prep_rc = RB_OK;
            std::string prep_err;
            if (!last) prep = std::thread([&] {
                prep_rc = guarded([&] { nxt = piece(next); });
                if (prep_rc != RB_OK) prep_err = rb_last_error();                      // the error text is thread-local
            });
            const int add_rc = guarded([&] { add_range(g, cur.b, 0, cur.b->n_reads, flags, stats); });
            if (prep.joinable()) prep.join();
            if (add_rc != RB_OK) { if (nxt.b) rb_batch_destroy(nxt.b); throw HipError{add_rc}; }
            if (prep_rc != RB_OK) { set_error("%s", prep_err.c_str()); throw HipError{prep_rc}; }
            if (last) break;
            a = next; cur = nxt;
        }
        if (n_records) *n_records = recs;
    });
    if (st && !own_stream && rc != RB_OK) (void)hipStreamSynchronize(st);      // (a failed call leaves nothing of its own in flight on the shared stream)
    slabs.end();
    if (st && own_stream) (void)hipStreamDestroy(st);
    return rc;
}

}  // extern "C"

// ---- streaming ingest: a FASTQ / FASTA FILE (plain or .gz) goes through the stage-1 worker's loop piece by piece ----
// FastqReader / FastaReader stream their file (R/io/FastqReader.java:140-186 over FileUtils.getTextFileReader, R/util/FileUtils.java:50-57:
// a GZIPInputStream for ".gz"); rb_graph_add_fastq wants the whole text in memory.  Here a reader thread reads — and for gzip
// input inflates — the next piece while the GPU inserts the current one: the file never exists as one buffer, and the inflate
// of piece c + 1 hides behind the insert of piece c.  gzip members are inflated as they come, any number of them; whatever follows a
// member and is not another gzip header ends the stream, as in GZIPInputStream; BGZF input takes the same path (its blocks are small members).
namespace {
struct TextSource {
    int fd = -1;
    bool gz = false, eof = false;
    z_stream z;
    bool z_open = false, z_member_done = true;
    std::vector<unsigned char> cbuf;           // compressed input window
    size_t cpos = 0, cend = 0;
    explicit TextSource(const char *path) {
        fd = open(path, O_RDONLY);
        RB_REQUIRE(fd >= 0, "cannot open %s", path);
        unsigned char magic[2] = {0, 0};
        const ssize_t got = pread(fd, magic, 2, 0);
        gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (gz) { cbuf.resize((size_t)4 << 20); memset(&z, 0, sizeof z); }
    }
    ~TextSource() { if (z_open) inflateEnd(&z); if (fd >= 0) close(fd); }
    bool refill() {                             // more compressed bytes; false at the end of the file
        if (cpos < cend) return true;
        const ssize_t got = read(fd, cbuf.data(), cbuf.size());
        RB_REQUIRE(got >= 0, "read error on the input file");
        cpos = 0; cend = (size_t)got;
        return got > 0;
    }
    // up to cap bytes of text into dst; returns the number written (0 only at the end of the input)
    size_t fill(char *dst, size_t cap) {
        size_t out = 0;
        if (!gz) {
            while (out < cap && !eof) {
                const ssize_t got = read(fd, dst + out, std::min(cap - out, (size_t)1 << 30));
                RB_REQUIRE(got >= 0, "read error on the input file");
                if (got == 0) eof = true;
                out += (size_t)got;
            }
            return out;
        }
        while (out < cap && !eof) {
            if (z_member_done) {                // between members: GZIPInputStream takes anything that is not another gzip header for the end of the stream
                if (!refill()) { eof = true; break; }
                if (z_open) {
                    if (cend - cpos < 2) {          // the header's two magic bytes may straddle the window: pull one more byte in
                        unsigned char b0 = cbuf[cpos], b1 = 0;
                        const ssize_t got = read(fd, &b1, 1);
                        if (got == 1) { cbuf[0] = b0; cbuf[1] = b1; cpos = 0; cend = 2; }
                    }
                    if (!(cend - cpos >= 2 && cbuf[cpos] == 0x1f && cbuf[cpos + 1] == 0x8b)) { eof = true; break; }
                }
                if (z_open) inflateReset(&z);
                else { RB_REQUIRE(inflateInit2(&z, 15 + 16) == Z_OK, "inflateInit2 failed"); z_open = true; }
                z_member_done = false;
            }
            if (!refill()) { set_error("unexpected end of the gzip data"); throw HipError{RB_ERR_INVALID}; }
            z.next_in = cbuf.data() + cpos; z.avail_in = (uInt)(cend - cpos);
            z.next_out = reinterpret_cast<unsigned char *>(dst + out); z.avail_out = (uInt)std::min(cap - out, (size_t)1 << 30);
            const uInt out0 = z.avail_out;
            const int rc = inflate(&z, Z_NO_FLUSH);
            cpos = cend - z.avail_in; out += out0 - z.avail_out;
            if (rc == Z_STREAM_END) z_member_done = true;
            else if (rc != Z_OK && rc != Z_BUF_ERROR) { set_error("not in gzip format / corrupt data (zlib %d)", rc); throw HipError{RB_ERR_INVALID}; }
        }
        return out;
    }
};

// the loop of rb_graph_add_fastq / _fasta over pieces that come from a TextSource: piece c + 1 is read (inflated), uploaded and
// parsed on a helper thread while piece c is inserted; what a piece leaves unparsed (an incomplete last record) is carried over
int add_text_file(rb_graph *g, const char *path, bool fasta, int min_base_qual, unsigned flags, rb_add_stats *stats, int64_t *n_records) {
    if (!g || !path) { set_error("rb_graph_add_%s_file: null argument", fasta ? "fasta" : "fastq"); return RB_ERR_INVALID; }
    hipStream_t st = nullptr;
    bool own_stream = false;
    char *buf[2] = {nullptr, nullptr};
    WriteLock wl(g->rw);
    int rc = guarded([&] {
        RB_HIP(hipSetDevice(g->p.device));
        st = getenv("RB_INGEST_OWN_STREAM") ? nullptr : g->pk_stream;
        if (!st) { RB_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); own_stream = true; }
        TextSource src(path);
        const size_t piece_bytes = getenv("RB_FASTQ_PIECE") ? (size_t)std::max(64, atoi(getenv("RB_FASTQ_PIECE"))) : (size_t)256 << 20;
        for (auto &b : buf) RB_HIP(hipHostMalloc(reinterpret_cast<void **>(&b), piece_bytes, hipHostMallocDefault));
        bool ended = false, src_done = false;
        // piece into buf[w]: `carry` bytes of the piece before (already at the front of buf[w]) + fresh text
        auto make = [&](int w, size_t carry, size_t *len_out) {
            const size_t got = src.fill(buf[w] + carry, piece_bytes - carry);
            const size_t len = carry + got;
            const bool final = got < piece_bytes - carry;      // the source ran dry: this is the last piece
            if (final) src_done = true;
            *len_out = len;
            bool end_here = false;
            DevPool *pool = getenv("RB_NO_INGEST_POOL") ? nullptr : &g->ingest_pool;
            rb::FastqChunk c = fasta ? rb::fasta_batch_create(g->p.device, buf[w], len, final, st, &end_here, pool)
                                     : rb::fastq_batch_create(g->p.device, buf[w], len, final, min_base_qual, true, st, pool);
            if (end_here) ended = true;
            return c;
        };
        int w = 0;
        size_t len = 0;
        int64_t recs = 0;
        rb::FastqChunk cur = make(0, 0, &len);
        for (;;) {
            struct G { rb_batch *b; ~G() { if (b) rb_batch_destroy(b); } } guard{cur.b};
            recs += cur.records;
            const bool last = src_done || ended;
            RB_REQUIRE(last || cur.consumed > 0, "a record longer than %zu bytes", piece_bytes);
            rb::FastqChunk nxt;
            size_t nlen = 0;
            std::thread prep;
            int prep_rc = RB_OK;
            std::string prep_err;
            if (!last) {
                const size_t carry = len - cur.consumed;
                memcpy(buf[1 - w], buf[w] + cur.consumed, carry);
                prep = std::thread([&, carry] {
                    prep_rc = guarded([&] { nxt = make(1 - w, carry, &nlen); });
                    if (prep_rc != RB_OK) prep_err = rb_last_error();
                });
            }
            const int add_rc = guarded([&] { if (cur.b && cur.b->n_reads) add_range(g, cur.b, 0, cur.b->n_reads, flags, stats); });
            if (prep.joinable()) prep.join();
            if (add_rc != RB_OK) { if (nxt.b) rb_batch_destroy(nxt.b); throw HipError{add_rc}; }
            if (prep_rc != RB_OK) { set_error("%s", prep_err.c_str()); throw HipError{prep_rc}; }
            if (last) break;
            w = 1 - w; len = nlen; cur = nxt;
        }
        if (n_records) *n_records = recs;
    });
    if (st && !own_stream && rc != RB_OK) (void)hipStreamSynchronize(st);      // (a failed call leaves nothing of its own in flight on the shared stream)
    for (auto b : buf) if (b) (void)hipHostFree(b);
    if (st && own_stream) (void)hipStreamDestroy(st);
    return rc;
}
}  // namespace

extern "C" {
int rb_graph_add_fastq_file(rb_graph *g, const char *path, int min_base_qual, unsigned flags, rb_add_stats *stats, int64_t *n_records) {
    return add_text_file(g, path, false, min_base_qual, flags, stats, n_records);
}
int rb_graph_add_fasta_file(rb_graph *g, const char *path, unsigned flags, rb_add_stats *stats, int64_t *n_records) {
    return add_text_file(g, path, true, 0, flags, stats, n_records);
}

int rb_graph_apply(rb_graph *g, int op, const uint64_t *h0, size_t n) {
    return guarded([&] {
        RB_REQUIRE(g && (h0 || n == 0), "rb_graph_apply: null argument");
        WriteLock wl(g->rw);
        RB_REQUIRE(op >= RB_OP_ADD && op <= RB_OP_ADD_FRAG_PAIR, "rb_graph_apply: unknown op %d", op);
        RB_REQUIRE(!g->shard, "rb_graph_apply: not available on a shard handle");
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        if (op == RB_OP_ADD_DBG_ONLY || op == RB_OP_ADD_READ_PAIR || op == RB_OP_ADD_FRAG_PAIR) {
            BitFilter *f = op == RB_OP_ADD_DBG_ONLY ? &g->dbg : op == RB_OP_ADD_READ_PAIR ? &g->rpk : &g->fpk;
            if (!f->bits) { set_error("rb_graph_apply: filter not initialised"); throw HipError{RB_ERR_STATE}; }
            if (n) {
                uint64_t *d = upload_h0(g, g->qbuf0, h0, n);
                hipLaunchKernelGGL(k_bits_add, dim3(blocks_for((int64_t)n)), dim3(TPB), 0, s, f->bits, f->mod, f->num_hash,
                                   kmul_of(g->k), d, n);
                RB_HIP(hipGetLastError());
            }
            RB_HIP(hipStreamSynchronize(s));
            return;
        }
        const int mode = op == RB_OP_ADD ? M_ADD : op == RB_OP_ADD_IF_ABSENT ? M_ADD_IF_ABSENT
                       : op == RB_OP_ADD_COUNT_IF_PRESENT ? M_COUNT_IF_PRESENT : M_COUNT_ONLY;
        size_t done = 0;
        const size_t chunk = (size_t)g->max_batch_kmers;
        while (done < n) {
            size_t m = std::min(chunk, n - done);
            g->keys0.reserve(m * 8); g->vals0.reserve(m * 4);
            RB_HIP(hipMemcpyAsync(g->keys0.p, h0 + done, m * 8, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_iota, dim3(blocks_for((int64_t)m)), dim3(TPB), 0, s, g->vals0.as<uint32_t>(), m);
            run_pipeline(g, m, mode, g->ordinal, 0, nullptr);
            g->ordinal += m;
            done += m;
        }
        RB_HIP(hipStreamSynchronize(s));
    });
}

}  // extern "C"
// development (tools/alloc_lottery.py): time n random returning atomics on the counting filter as it lies in memory — every word is
// XORed twice with the same value, so the contents are what they were.  mode 0: atomicOr with 0 (reads), 1: XOR pairs (read-modify-write)
namespace {
__global__ void k_debug_probe(uint32_t *words, uint64_t n_words, uint32_t per_thread, int mode, uint64_t salt, unsigned long long *sink) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = (t + salt) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_thread; ++i) {
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        uint32_t *w = &words[(uint64_t)(((unsigned __int128)x * n_words) >> 64)];
        acc |= mode ? atomicXor(w, 0x80808080u) : atomicOr(w, 0u);
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}
}  // namespace
extern "C" {
int rb_debug_probe_cbf(rb_graph *g, int mode, float *ms_out) {
    return guarded([&] {
        RB_REQUIRE(g && g->cbf && ms_out, "rb_debug_probe_cbf: bad argument");
        rb::WriteLock wl(g->rw);
        RB_HIP(hipSetDevice(g->p.device));
        hipStream_t s = g->stream;
        g->devctr.reserve(DEVCTR_BYTES);
        hipEvent_t e0, e1;
        RB_HIP(hipEventCreate(&e0)); RB_HIP(hipEventCreate(&e1));
        RB_HIP(hipStreamSynchronize(s));
        RB_HIP(hipEventRecord(e0, s));
        // the same places twice: the second pass undoes the first (mode 1)
        for (int pass = 0; pass < 2; ++pass)
            hipLaunchKernelGGL(k_debug_probe, dim3(65536), dim3(256), 0, s, reinterpret_cast<uint32_t *>(g->cbf), (uint64_t)(g->cbf_alloc / 4), 16u, mode, 0ull, g->devctr.as<unsigned long long>());
        RB_HIP(hipEventRecord(e1, s));
        RB_HIP(hipEventSynchronize(e1));
        RB_HIP(hipEventElapsedTime(ms_out, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    });
}

int rb_debug_scan_u32(int device, const uint32_t *in, size_t n, uint32_t *out, int misalign) {
    DevBuf a, b, t;
    struct Rel { DevBuf &a, &b, &t; ~Rel() { a.release(); b.release(); t.release(); } } rel{a, b, t};
    return guarded([&] {
        RB_REQUIRE((in && out) || n == 0, "rb_debug_scan_u32: null array");
        RB_REQUIRE(misalign >= 0 && misalign < 16, "rb_debug_scan_u32: misalign in 0..15");
        RB_HIP(hipSetDevice(device));
        a.reserve((n + 8) * 4); b.reserve((n + 8) * 4); t.reserve(scan_temp_bytes(n));
        uint32_t *di = a.as<uint32_t>() + (misalign & 3), *dout = b.as<uint32_t>() + ((misalign >> 2) & 3);   // words past a 16-byte boundary: input, output
        if (n) RB_HIP(hipMemcpy(di, in, n * 4, hipMemcpyHostToDevice));
        exclusive_scan_u32(t.p, t.cap, di, dout, n, nullptr);
        RB_HIP(hipGetLastError());
        RB_HIP(hipDeviceSynchronize());
        if (n) RB_HIP(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    });
}
int rb_debug_sort_pairs(int device, uint64_t *keys, void *vals, int vals64, size_t n, int lo_begin, int lo_end, int hi_begin, int hi_end) {
    DevBuf k0, k1, v0, v1, t;
    struct Rel { DevBuf &a, &b, &c, &d, &t; ~Rel() { a.release(); b.release(); c.release(); d.release(); t.release(); } } rel{k0, k1, v0, v1, t};
    return guarded([&] {
        RB_REQUIRE(keys || n == 0, "rb_debug_sort_pairs: null keys");
        RB_REQUIRE(!(vals64 && hi_begin >= 0), "rb_debug_sort_pairs: two ranges with 32-bit values only");
        RB_HIP(hipSetDevice(device));
        if (n == 0) return;
        const size_t vb = vals64 ? 8 : 4;
        k0.reserve(n * 8); k1.reserve(n * 8);
        if (vals) { v0.reserve(n * vb); v1.reserve(n * vb); RB_HIP(hipMemcpy(v0.p, vals, n * vb, hipMemcpyHostToDevice)); }
        RB_HIP(hipMemcpy(k0.p, keys, n * 8, hipMemcpyHostToDevice));
        t.reserve(vals64 ? sort_pairs32_temp_bytes(n) : sort_pairs_temp_bytes(n));
        if (vals64) sort_pairs_u64_u64(t.p, t.cap, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint64_t>(), v1.as<uint64_t>(), n, lo_begin, lo_end, nullptr);
        else if (hi_begin >= 0) sort_pairs_u64_u32_2r(t.p, t.cap, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint32_t>(), v1.as<uint32_t>(), n, lo_begin, lo_end, hi_begin, hi_end, nullptr);
        else if (vals) sort_pairs_u64_u32(t.p, t.cap, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint32_t>(), v1.as<uint32_t>(), n, lo_begin, lo_end, nullptr);
        else sort_keys_u64(t.p, t.cap, k0.as<uint64_t>(), k1.as<uint64_t>(), n, lo_begin, lo_end, nullptr);
        RB_HIP(hipGetLastError());
        RB_HIP(hipDeviceSynchronize());
        RB_HIP(hipMemcpy(keys, k1.p, n * 8, hipMemcpyDeviceToHost));
        if (vals) RB_HIP(hipMemcpy(vals, v1.p, n * vb, hipMemcpyDeviceToHost));
    });
}
int rb_filter_size(rb_graph *g, int which, int64_t *size, int64_t *nbytes, int *num_hash) {
    if (!g) { set_error("null graph"); return RB_ERR_INVALID; }
    if (which == RB_CBF) {
        if (!g->cbf) { set_error("rb_filter_size: filter %d not initialised", which); return RB_ERR_STATE; }
        if (size) *size = g->cbf_size;
        if (nbytes) *nbytes = g->cbf_hi - g->cbf_lo;
        if (num_hash) *num_hash = g->cbf_h;
        return RB_OK;
    }
    BitFilter *f = bit_filter(g, which);
    if (!f) { set_error("rb_filter_size: unknown filter %d", which); return RB_ERR_INVALID; }
    if (!f->bits) { set_error("rb_filter_size: filter %d not initialised", which); return RB_ERR_STATE; }
    if (size) *size = f->size;
    if (nbytes) *nbytes = f->nbytes;
    if (num_hash) *num_hash = f->num_hash;
    return RB_OK;
}

// popcount (fold == false) or digest (fold == true) of the locally held part of a filter; a read-only call: shared lock + a
// leased query context, so concurrent queries do not wait behind it
static int filter_reduce(rb_graph *g, int which, bool fold, unsigned long long *out) {
    return guarded([&] {
        RB_REQUIRE(g && out, "rb_filter_popcount / rb_filter_fold: null argument");
        QueryLease q(g);
        RB_HIP(hipStreamSynchronize(g->stream));   // shard phases return with work in flight on the handle's stream
        q.c->b0.reserve(64);
        unsigned long long *acc = q.c->b0.as<unsigned long long>();
        RB_HIP(hipMemsetAsync(acc, 0, 8, q.c->st));
        const uint32_t *words; size_t nw; uint64_t gw0;
        if (which == RB_CBF) {
            if (!g->cbf) { set_error("rb_filter_popcount: filter %d not initialised", which); throw HipError{RB_ERR_STATE}; }
            words = reinterpret_cast<const uint32_t *>(g->cbf); nw = g->cbf_alloc / 4;   // padding bytes are zero
            gw0 = (uint64_t)g->cbf_lo / 4u;
        } else {
            BitFilter *f = bit_filter(g, which);
            RB_REQUIRE(f, "rb_filter_popcount: unknown filter %d", which);
            if (!f->bits) { set_error("rb_filter_popcount: filter %d not initialised", which); throw HipError{RB_ERR_STATE}; }
            words = f->bits; nw = f->alloc / 4;
            gw0 = (uint64_t)f->lo / 32u;
        }
        const dim3 grid(std::min<unsigned>(blocks_for((int64_t)nw), 8192u));
        if (fold) hipLaunchKernelGGL(k_fold_words, grid, dim3(TPB), 0, q.c->st, words, nw, gw0, acc);
        else if (which == RB_CBF) hipLaunchKernelGGL(k_count_nonzero_bytes, grid, dim3(TPB), 0, q.c->st, words, nw, acc);
        else hipLaunchKernelGGL(k_popcount_bits, grid, dim3(TPB), 0, q.c->st, words, nw, acc);
        RB_HIP(hipGetLastError());
        unsigned long long v = 0;
        RB_HIP(hipMemcpyAsync(&v, acc, 8, hipMemcpyDeviceToHost, q.c->st));
        RB_HIP(hipStreamSynchronize(q.c->st));
        *out = v;
    });
}
int rb_filter_popcount(rb_graph *g, int which, int64_t *out) {
    unsigned long long v = 0;
    if (!out) { set_error("rb_filter_popcount: null argument"); return RB_ERR_INVALID; }
    int rc = filter_reduce(g, which, false, &v);
    if (rc == RB_OK) *out = (int64_t)v;
    return rc;
}
int rb_filter_fold(rb_graph *g, int which, uint64_t *out) {
    unsigned long long v = 0;
    if (!out) { set_error("rb_filter_fold: null argument"); return RB_ERR_INVALID; }
    int rc = filter_reduce(g, which, true, &v);
    if (rc == RB_OK) *out = (uint64_t)v;
    return rc;
}

int rb_filter_fpr(rb_graph *g, int which, float *out) {
    int64_t pop = 0, size = 0; int h = 0;
    int rc = rb_filter_popcount(g, which, &pop);
    if (rc != RB_OK) return rc;
    rc = rb_filter_size(g, which, &size, nullptr, &h);
    if (rc != RB_OK) return rc;
    if (!out) { set_error("null argument"); return RB_ERR_INVALID; }
    *out = (float)pow((double)pop / (double)size, h);   // BloomFilter.getFPR :185-194
    return RB_OK;
}

int rb_filter_export(rb_graph *g, int which, void *dst, size_t nbytes) {
    return guarded([&] {
        RB_REQUIRE(g && dst, "rb_filter_export: null argument");
        QueryLease q(g);                      // read-only: shared lock (mutators finish their work before they release the handle)
        const void *src; size_t have;
        if (which == RB_CBF) { src = g->cbf; have = (size_t)(g->cbf_hi - g->cbf_lo); }
        else {
            BitFilter *f = bit_filter(g, which);
            RB_REQUIRE(f, "rb_filter_export: unknown filter %d", which);
            if (!f->bits) { set_error("rb_filter_export: filter %d not initialised", which); throw HipError{RB_ERR_STATE}; }
            src = f->bits; have = (size_t)f->nbytes;
        }
        RB_REQUIRE(nbytes == have, "rb_filter_export: buffer is %zu bytes, filter has %zu", nbytes, have);
        RB_HIP(hipStreamSynchronize(g->stream));
        RB_HIP(hipMemcpyAsync(dst, src, have, hipMemcpyDeviceToHost, q.c->st));
        RB_HIP(hipStreamSynchronize(q.c->st));
    });
}

int rb_filter_import(rb_graph *g, int which, const void *srcp, size_t nbytes) {
    return guarded([&] {
        RB_REQUIRE(g && srcp, "rb_filter_import: null argument");
        WriteLock wl(g->rw);
        RB_HIP(hipSetDevice(g->p.device));
        void *dst; size_t have, alloc;
        if (which == RB_CBF) { dst = g->cbf; have = (size_t)(g->cbf_hi - g->cbf_lo); alloc = g->cbf_alloc; }
        else {
            BitFilter *f = bit_filter(g, which);
            RB_REQUIRE(f, "rb_filter_import: unknown filter %d", which);
            if (!f->bits) { set_error("rb_filter_import: filter %d not initialised", which); throw HipError{RB_ERR_STATE}; }
            dst = f->bits; have = (size_t)f->nbytes; alloc = f->alloc;
        }
        RB_REQUIRE(nbytes == have, "rb_filter_import: buffer is %zu bytes, filter has %zu", nbytes, have);
        if (which == RB_CBF) {   // counters are MiniFloat bytes 0..127; bit 7 is the library's transient claim mark
            const uint8_t *b = static_cast<const uint8_t *>(srcp);
            for (size_t i = 0; i < nbytes; ++i)
                RB_REQUIRE(!(b[i] & 0x80u), "rb_filter_import: counter byte %zu is %u (> 127, not a MiniFloat count)", i, (unsigned)b[i]);
        }
        RB_HIP(hipStreamSynchronize(g->stream));
        if (g->npf_log2 && (which == RB_CBF || which == RB_DBGBF)) RB_HIP(hipMemset(g->npf.p, 0, sizeof(uint64_t) << g->npf_log2));
        if (g->mpf_log2b && (which == RB_CBF || which == RB_DBGBF)) RB_HIP(hipMemset(g->mpf.p, 0, (size_t)128 << g->mpf_log2b));
        if (which != RB_CBF) seen_reset(*bit_filter(g, which), g->stream);      // the bits are replaced: what the seen-pair cache knew is void
        if (which == RB_RPKBF && g->shard) rb::shard_clear_pairs_acc(g);        // (and what this rank's accumulation copy still holds must not come back)
        RB_HIP(hipStreamSynchronize(g->stream));
        RB_HIP(hipMemset(dst, 0, alloc));
        RB_HIP(hipMemcpy(dst, srcp, have, hipMemcpyHostToDevice));
        RB_HIP(hipDeviceSynchronize());
    });
}

int64_t rb_expected_size(int64_t n, float fpr, int num_hash) {   // BloomFilter.getExpectedSize :196-199
    double r = (double)(-num_hash) / log(1 - exp(log((double)fpr) / (double)num_hash));
    return (int64_t)ceil((double)n * r);
}

int rb_graph_profile_enable(rb_graph *g, int on) {
    if (!g) { set_error("null graph"); return RB_ERR_INVALID; }
    g->prof_on = on != 0; return RB_OK;
}
int rb_graph_profile_get(rb_graph *g, rb_profile *out, int reset) {
    if (!g || !out) { set_error("null argument"); return RB_ERR_INVALID; }
    g->prof_collect();
    out->n = 0;
    for (auto &e : g->prof) {
        if (out->n >= RB_PROF_MAX) break;
        out->name[out->n] = e.name; out->ms[out->n] = e.ms; out->launches[out->n] = e.launches; out->n++;
    }
    if (reset) g->prof.clear();
    return RB_OK;
}

/* BloomFilterDeBruijnGraph.destroyDbgbf / destroyCbf / destroyRpkbf / destroyFpkbf :249-275: the memory goes back to the device */
int rb_graph_destroy_filter(rb_graph *g, int which) {
    return guarded([&] {
        RB_REQUIRE(g && !g->shard, "rb_graph_destroy_filter: null or shard handle");
        WriteLock wl(g->rw);
        RB_HIP(hipSetDevice(g->p.device));
        RB_HIP(hipStreamSynchronize(g->stream)); RB_HIP(hipStreamSynchronize(g->stream2));
        if (which == RB_CBF) {
            if (g->cbf) RB_HIP(hipFree(g->cbf));
            g->cbf = nullptr; g->cbf_size = 0; g->cbf_lo = g->cbf_hi = 0; g->cbf_alloc = 0;
        } else {
            BitFilter *f = bit_filter(g, which);
            RB_REQUIRE(f, "rb_graph_destroy_filter: unknown filter %d", which);
            free_bits(*f);
        }
    });
}

int rb_filter_increment_and_get(rb_graph *g, const uint64_t *h0, size_t n, float *out) {
    return guarded([&] {
        RB_REQUIRE(g && !g->shard && g->cbf && (n == 0 || (h0 && out)), "rb_filter_increment_and_get: bad argument or no counting filter");
        WriteLock wl(g->rw);
        if (!n) return;
        RB_HIP(hipSetDevice(g->p.device));
        uint64_t *d = upload_h0(g, g->qbuf0, h0, n);
        g->qbuf1.reserve(n * 4);
        hipLaunchKernelGGL(k_increment_and_get, dim3(1), dim3(64), 0, g->stream, g->view(g->ordinal, 0, false), d, n, g->qbuf1.as<float>());
        RB_HIP(hipGetLastError());
        g->ordinal += n;
        RB_HIP(hipMemcpyAsync(out, g->qbuf1.p, n * 4, hipMemcpyDeviceToHost, g->stream));
        RB_HIP(hipStreamSynchronize(g->stream));
        // the counters moved: what the prefilter caches assert stays true (counters only grow)
    });
}

int rb_cbf_to_bloom(rb_graph *src, float min_cov, rb_graph *dst, int which) {
    return guarded([&] {
        RB_REQUIRE(src && dst && !src->shard && !dst->shard, "rb_cbf_to_bloom: bad handles");
        WriteLock l1(src < dst ? src->rw : dst->rw, std::defer_lock), l2(src < dst ? dst->rw : src->rw, std::defer_lock);
        l1.lock(); if (src != dst) l2.lock();
        RB_REQUIRE(src->cbf, "rb_cbf_to_bloom: the source has no counting filter");
        BitFilter *f = bit_filter(dst, which);
        RB_REQUIRE(f && f->bits, "rb_cbf_to_bloom: destination filter %d not initialised", which);
        RB_REQUIRE(f->size == src->cbf_size && src->p.device == dst->p.device, "rb_cbf_to_bloom: size (%lld vs %lld) or device mismatch",
                   (long long)f->size, (long long)src->cbf_size);
        RB_HIP(hipSetDevice(src->p.device));
        RB_HIP(hipStreamSynchronize(dst->stream));
        const int64_t words = (src->cbf_size + 31) / 32;
        hipLaunchKernelGGL(k_cbf_to_bits, dim3(blocks_for(words)), dim3(TPB), 0, src->stream, src->cbf, src->cbf_size, min_cov, f->bits);
        RB_HIP(hipGetLastError());
        seen_reset(*f, src->stream);                 // every word of the filter was rewritten
        RB_HIP(hipStreamSynchronize(src->stream));
        if (which == RB_DBGBF) {
            if (dst->npf_log2) RB_HIP(hipMemset(dst->npf.p, 0, sizeof(uint64_t) << dst->npf_log2));
            if (dst->mpf_log2b) RB_HIP(hipMemset(dst->mpf.p, 0, (size_t)128 << dst->mpf_log2b));
            RB_HIP(hipDeviceSynchronize());
        }
    });
}

}  // extern "C"
