// rb_internal.hpp — host-side internals shared by the translation units of librb_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/rb_capi.h"
#include "rb_device.hpp"

namespace rb {

// Best-effort pinning of a caller's host buffer for the duration of a query call: pageable pages go over the link at
// ~10-15 GB/s through the runtime's staging buffers, registered ones at ~57 GB/s (hipHostRegister itself: ~8 ms per GB).
// Buffers that cannot be registered (foreign mappings, already registered) are copied the slow way.
struct HostPin {
    void *p = nullptr;
    // memory that is pinned already (hipHostMalloc'ed, or registered by the caller) is left alone: registering a piece of it again
    // either fails or pins and unpins the pages once more — 8 ms per GB each way, which halved the rate of the packed stream's uploads
    static bool pinned_already(const void *ptr) {     // (public: rb_packed.hip registers arrays that outlive a scope)
        hipPointerAttribute_t a;
        const bool yes = hipPointerGetAttributes(&a, ptr) == hipSuccess && a.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        return yes;
    }
    HostPin(const void *ptr, size_t bytes) {
        if (ptr && bytes > ((size_t)16 << 20) && !getenv("RB_NO_PIN") && !pinned_already(ptr) &&
            hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterDefault) == hipSuccess) p = const_cast<void *>(ptr);
        else (void)hipGetLastError();
    }
    ~HostPin() { if (p) (void)hipHostUnregister(p); }
    HostPin(const HostPin &) = delete;
    HostPin &operator=(const HostPin &) = delete;
};


void set_error(const char *fmt, ...);

struct HipError {
    int code;
};

#define RB_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            ::rb::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            throw ::rb::HipError{_e == hipErrorOutOfMemory ? RB_ERR_NOMEM : RB_ERR_HIP};    \
        }                                                                                   \
    } while (0)

#define RB_REQUIRE(cond, ...)                          \
    do {                                               \
        if (!(cond)) {                                 \
            ::rb::set_error(__VA_ARGS__);              \
            throw ::rb::HipError{RB_ERR_INVALID};      \
        }                                              \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) RB_HIP(hipFree(p));
        p = nullptr; cap = 0;
        size_t want = bytes + (bytes >> 3) + 256;
        RB_HIP(hipMalloc(&p, want));
        cap = want;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace rb

// ---- device-resident read batch (packed) ----
// Read r occupies words [woff[r], woff[r+1]) ; word w holds 32 bases: 2-bit codes in codes[w]
// (base i at bits 2i..2i+1, LSB first) and one usable-bit per base in valid[w].
// host copy of a batch's word offsets: the batch's own vector, or — batches of a packed stream (rb_packed.hip) — pinned memory of the
// stream the offsets were copied into at link speed (a pageable destination costs 10 ms per 50 MB chunk on the upload's critical path)
struct HostWoff {
    std::vector<uint32_t> own;
    uint32_t *ext = nullptr;
    size_t ext_n = 0;
    const uint32_t *data() const { return ext ? ext : own.data(); }
    uint32_t *data() { return ext ? ext : own.data(); }
    size_t size() const { return ext ? ext_n : own.size(); }
    bool empty() const { return size() == 0; }
    const uint32_t &operator[](size_t i) const { return data()[i]; }
    uint32_t &operator[](size_t i) { return data()[i]; }
    const uint32_t *begin() const { return data(); }
    const uint32_t *end() const { return data() + size(); }
    void assign(size_t n, uint32_t v) { ext = nullptr; own.assign(n, v); }
    void resize(size_t n) { ext = nullptr; own.resize(n); }
    HostWoff &operator=(const std::vector<uint32_t> &v) { ext = nullptr; own = v; return *this; }
    void borrow(uint32_t *p, size_t n) { own.clear(); ext = p; ext_n = n; }
};

// A caller's large host array registered slab by slab as an upload advances through it, instead of as a whole before the first copy (hipHostRegister
// takes 8 ms per GB: 126 ms for a 15.75 GB FASTQ text with the link idle meanwhile).  Slabs end on 256 MiB boundaries of the ADDRESS, so neighbours
// never share a page; a copy must stay inside one registration (HIP refuses a source range over two as an invalid argument): copy() splits there.
// Best effort: where registration fails the copies still work, from pageable memory.  One user at a time.
struct SlabPin {
    static constexpr uintptr_t SLAB = (uintptr_t)256 << 20;
    uintptr_t hi = 0, done = 0;
    bool on = false;
    std::vector<void *> pins;
    void begin(const void *p, size_t bytes) {
        on = false;
        if (!p || bytes <= ((size_t)16 << 20) || getenv("RB_NO_PIN")) return;
        hipPointerAttribute_t a;
        const bool pinned = hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        if (pinned) return;
        done = reinterpret_cast<uintptr_t>(p); hi = done + bytes; on = true;
    }
    void pin_to(const void *upto) {                    // register slabs until everything below `upto` is covered
        const uintptr_t need = std::min(reinterpret_cast<uintptr_t>(upto), hi);
        while (on && done < need) {
            const uintptr_t a = done, e = std::min(hi, (a & ~(SLAB - 1)) + SLAB);
            if (hipHostRegister(reinterpret_cast<void *>(a), (size_t)(e - a), hipHostRegisterDefault) == hipSuccess) pins.push_back(reinterpret_cast<void *>(a));
            else { (void)hipGetLastError(); on = false; }
            done = e;
        }
    }
    void end() { for (void *q : pins) (void)hipHostUnregister(q); pins.clear(); on = false; }
    ~SlabPin() { end(); }
    static hipError_t copy(void *dst, const char *src, size_t n, hipStream_t st) {
        for (size_t o = 0; o < n;) {
            const uintptr_t at = reinterpret_cast<uintptr_t>(src + o);
            const size_t m = std::min<size_t>(n - o, (size_t)(((at & ~(SLAB - 1)) + SLAB) - at));
            const hipError_t e = hipMemcpyAsync(static_cast<char *>(dst) + o, src + o, m, hipMemcpyHostToDevice, st);
            if (e != hipSuccess) return e;
            o += m;
        }
        return hipSuccess;
    }
};

// Device blocks that chunked ingests hand back and take again (rb_graph_add_reads cuts a call into chunks of 256 M bases, each a batch of six
// arrays + three staging arrays: hipMalloc / hipFree of those cost 6 ms a chunk — a third of the call — and hipFree waits for the device).
// get() takes the smallest cached block that fits and is at most twice the size, else allocates; put() keeps up to `limit` bytes.
struct DevPool {
    std::mutex m;
    std::vector<std::pair<void *, size_t>> free_;
    std::unordered_map<void *, size_t> out;      // blocks handed out -> capacity
    size_t held = 0, limit = (size_t)12 << 30;
    void *get(size_t bytes) {
        bytes = std::max<size_t>((bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1), (size_t)2 << 20);
        {
            std::lock_guard<std::mutex> lk(m);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].second >= bytes && free_[i].second <= 2 * bytes && (best == free_.size() || free_[i].second < free_[best].second)) best = i;
            if (best != free_.size()) {
                void *p = free_[best].first; const size_t cap = free_[best].second;
                free_.erase(free_.begin() + (std::ptrdiff_t)best);
                held -= cap; out[p] = cap;
                return p;
            }
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { clear(); e = hipMalloc(&p, bytes); }        // (out of memory with blocks cached: give them back first)
        if (e != hipSuccess) { (void)hipGetLastError(); rb::set_error("hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e)); throw rb::HipError{RB_ERR_NOMEM}; }
        std::lock_guard<std::mutex> lk(m);
        out[p] = bytes;
        return p;
    }
    void put(void *p) {
        if (!p) return;
        size_t cap = 0;
        {
            std::lock_guard<std::mutex> lk(m);
            auto it = out.find(p);
            if (it != out.end()) { cap = it->second; out.erase(it); }
            if (cap && held + cap <= limit) { free_.push_back({p, cap}); held += cap; return; }
        }
        (void)hipFree(p);
    }
    void clear() {
        std::vector<std::pair<void *, size_t>> f;
        { std::lock_guard<std::mutex> lk(m); f.swap(free_); held = 0; }
        for (auto &b : f) (void)hipFree(b.first);
    }
    ~DevPool() { clear(); }
};

// Pinned host scratch of one chunk's preparation (word offsets, lengths, base offsets relative to the chunk): grow-only, two of them take turns on a
// handle — chunk c + 1 is prepared in one while the insert of chunk c still reads the offsets in the other.  Pinned: the three copies to the
// device are real asynchronous copies (from a std::vector they are staged, and the preparing thread waits for them).
struct IngestHost {
    uint32_t *woff = nullptr, *len = nullptr;
    int64_t *rel = nullptr;
    size_t cap = 0;
    void reserve(size_t n) {
        if (n <= cap) return;
        release();
        const size_t want = n + (n >> 2) + 1024;
        void *p = nullptr;
        if (hipHostMalloc(&p, want * 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); rb::set_error("hipHostMalloc of %zu bytes failed", want * 16); throw rb::HipError{RB_ERR_NOMEM}; }
        rel = static_cast<int64_t *>(p); woff = reinterpret_cast<uint32_t *>(rel + want); len = woff + want;
        cap = want;
    }
    void release() { if (rel) (void)hipHostFree(rel); rel = nullptr; woff = len = nullptr; cap = 0; }
    ~IngestHost() { release(); }
};

struct rb_batch {
    int device = 0;
    DevPool *pool = nullptr;       // the arrays below came out of this pool and go back to it (rb_batch_destroy); nullptr: hipMalloc / hipFree
    int64_t n_reads = 0, n_bases = 0, n_words = 0;
    uint32_t max_len = 0;
    uint32_t wpr_uniform = 0;      // words per read when every read has the same word count, else 0
    uint64_t *codes = nullptr;     // [n_words]
    uint32_t *valid = nullptr;     // [n_words]
    uint32_t *rnz = nullptr;       // [n_words] or nullptr: bit = the base's REVERSE-strand seed is non-zero.  The reference looks the complement's
                                   // seed up by `ch & 7` (R/bloom/hash/NTHash.java:30, 133-166), so some non-ACGTU letters (K M S W Y I E L O Q D ...)
                                   // hash as a base on the reverse strand and as nothing on the forward strand; for such a base the code bits hold
                                   // the code whose complement's seed that is.  Only batches made for all-window hashing (getKmers, sketches) carry it.
    uint32_t *word_read = nullptr; // [n_words] owning read of each word
    uint32_t *woff = nullptr;      // [n_reads+1]
    uint32_t *len = nullptr;       // [n_reads]
    size_t device_bytes = 0;
    HostWoff h_woff;               // host copy of woff (sub-batch splitting)
};

namespace rb {

// device-wide sort / scan / selection primitives (hand-written: scans and selection in rb_sort.hip, the LSD radix sorts in
// rb_group.hip on the grouping stage's stable partition passes).  The sorts clobber their INPUT arrays.
size_t sort_pairs_temp_bytes(size_t n);
void sort_pairs_u64_u32(void *temp, size_t temp_bytes, uint64_t *keys_in, uint64_t *keys_out,
                        uint32_t *vals_in, uint32_t *vals_out, size_t n, int begin_bit, int end_bit,
                        hipStream_t s);
// the same on two bit ranges, the lower first (the bits in between are equal in all keys)
void sort_pairs_u64_u32_2r(void *temp, size_t temp_bytes, uint64_t *keys_in, uint64_t *keys_out, uint32_t *vals_in, uint32_t *vals_out, size_t n,
                           int lo_begin, int lo_end, int hi_begin, int hi_end, hipStream_t s);
size_t sort_pairs32_temp_bytes(size_t n);
void sort_pairs_u64_u64(void *temp, size_t temp_bytes, uint64_t *keys_in, uint64_t *keys_out,
                        uint64_t *vals_in, uint64_t *vals_out, size_t n, int begin_bit, int end_bit,
                        hipStream_t s);
size_t sort_keys_temp_bytes(size_t n);
void sort_keys_u64(void *temp, size_t temp_bytes, uint64_t *keys_in, uint64_t *keys_out, size_t n, int begin_bit,
                   int end_bit, hipStream_t s);
size_t scan_temp_bytes(size_t n);
void exclusive_scan_u32(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, size_t n,
                        hipStream_t s);
// indices i in [0,n) with (status[i] & mask) != 0, in order; count written to *count_dev
size_t select_temp_bytes(size_t n);
void select_flagged(void *temp, size_t temp_bytes, const uint32_t *status, uint32_t mask, size_t n, uint32_t *out,
                    uint32_t *count_dev, hipStream_t s);
// the same for two masks at once, out_a / out_b in index order, counts to count_dev[0..1]
size_t select2_temp_bytes(size_t n);
void select_flagged2(void *temp, size_t temp_bytes, const uint32_t *status, size_t n, uint32_t mask_a, uint32_t *out_a,
                     uint32_t mask_b, uint32_t *out_b, uint32_t *count_dev, hipStream_t s);

// hand-written grouping stage of the insert pipeline (rb_group.hip): N (h0, occurrence) records -> occurrences in
// grouped order, their draw strengths, runs (hash, count, start) and the run count, all on `st`, nothing synchronised.
// keys0/vals0 are clobbered; keys_tmp/vals_tmp are scratch of the same size.
constexpr int GR_FLAG_DEAD = 1;     // the records may include ones the emit pass cancelled (key and occurrence id all ones): dropped by the first partition pass
// Index-keyed partition passes (round 4): the digits of the MSD partition are taken from the k-mer's first filter index (idx_0 =
// (h0 >>> 1) % size, mapped onto 2^T equal index ranges of [lo, lo + span): a fine bucket IS an index range) instead of from the hash's top
// bits.  Any function of the hash is a valid grouping key (equal hashes stay together); with this one the fine buckets — and the runs
// the bucket kernel emits, in ticket order — sweep the filters by index range, so the first Bloom bit and the first counter of
// consecutive runs fall into a moving window of a few hundred KB instead of all over the array.
struct GrIdx {
    Mod mod;                    // the filter's size (index_of)
    uint64_t lo = 0, span = 0;  // this handle's index range (a shard's; the whole filter otherwise).  span == 0: off
};
// what a grouping can leave behind for the swept Bloom-bit stage below: per fine bucket the first run slot and the number of runs (the
// bucket's runs are contiguous; an oversized bucket has none there), and the number of runs before the oversized buckets' (appended last)
struct GroupExport { uint32_t *brun = nullptr, *bnr = nullptr, *n_main = nullptr; };
void group_debug_big(const void *temp, size_t N, int group_bits, int bucket_target, uint32_t *n_big_out, uint64_t *records_out, uint32_t *largest_out, int flags = 0);
size_t group_temp_bytes(size_t N, int group_bits, int bucket_target = 0, int flags = 0);
const uint32_t *group_live_count(const void *temp, size_t N, int group_bits, int bucket_target, int flags);
void group_records_device(uint64_t *keys0, uint32_t *vals0, uint64_t *keys_tmp, uint32_t *vals_tmp, size_t N, int group_bits,
                          uint64_t seed, uint64_t ordinal0, uint32_t pos_bits, void *temp, size_t temp_bytes,
                          uint32_t *vals_out, uint8_t *tz_out, uint64_t *uniq, uint32_t *counts, uint32_t *starts, uint32_t *n_runs_dev,
                          hipStream_t st, struct rb_graph *prof = nullptr /* per-kernel HIP-event timing into this handle's profile */,
                          int bucket_target = 0 /* average fine-bucket size aimed at (0: the default, 3072) */, int flags = 0 /* GR_FLAG_* */,
                          GrIdx idx = GrIdx{Mod{1, 0, 0}, 0, 0} /* span != 0: first partition digit from the first filter index */,
                          GroupExport ex = GroupExport{});
// T, the log2 of the number of index ranges (fine buckets) such a grouping partitions by; 0: it would not be index-keyed
uint32_t group_index_buckets(size_t N, int group_bits, int bucket_target, int flags, GrIdx idx);
// Swept Bloom-bit stage (rb_group.hip): both Bloom bits of the D runs of an index-keyed grouping with 2^T buckets are tested and set range by
// range through LDS; st0[d] / st1[d] = what probe 0 / 1 of run d found: 0 it set the bit, 1 set before the sub-batch, 2 set by another probe of
// the sub-batch.  keys_* / vals_* are scratch for D records each.  The filter must be whole (idx.lo = 0 is the filter's first bit).
size_t sweep_temp_bytes(size_t D, uint32_t T);
void sweep_bits_device(uint32_t *words, GrIdx idx, uint32_t T, uint64_t kmul, const uint64_t *uniq, uint32_t D, uint32_t n_main, const uint32_t *brun,
                       const uint32_t *bnr, uint64_t *keys_a, uint32_t *vals_a, uint64_t *keys_b, uint32_t *vals_b, void *temp, size_t temp_bytes,
                       uint8_t *st0, uint8_t *st1, hipStream_t st);

}  // namespace rb
