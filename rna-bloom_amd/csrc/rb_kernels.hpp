// rb_kernels.hpp — launchers shared between translation units
#pragma once
#include "rb_internal.hpp"

namespace rb {

// number of windows of `span` usable bases starting in each 32-base word of reads' words [w0,w0+nw)
void launch_count_windows(const rb_batch *b, int64_t w0, int64_t nw, int span, uint32_t *cnt, hipStream_t s);
// base hash (hVals[0]) of every usable k-mer window, written densely in read order
void launch_hash_windows(const rb_batch *b, int64_t w0, int64_t nw, int k, int mode,
                         const uint32_t *chunk_off, uint32_t first_read, uint32_t pos_bits,
                         uint64_t *keys, uint32_t *vals, uint32_t *out_read, uint32_t *out_pos,
                         hipStream_t s);

// fast path with the no-op prefilter (k <= 31): per word the number of KEPT windows and their bit mask;
// total usable windows are accumulated into total_spread[16*q], q < 32.  own: only
// windows whose k-mer this rank owns (own_mine: first counter index inside [own.lo, own.hi)) are considered at all
// (sharded engine; mask 0 = every window).  cache.tab == nullptr: no prefilter, ownership only.
void launch_filter_windows(const rb_batch *b, int64_t w0, int64_t nw, int k, int mode, uint32_t first_read,
                           uint32_t pos_bits, uint64_t seed, uint64_t ordinal0, Npf cache, uint32_t *cnt, uint32_t *keepmask,
                           uint32_t *total_spread, hipStream_t s, OwnRange own = OwnRange{Mod{1, 0, 0}, 0, 0},
                           Mpf mcache = Mpf{nullptr, 0, 0},    // mcache.tab != nullptr: minimizer-bucketed cache instead of `cache`
                           void *wstate = nullptr);            // filter_saves_state(): 16 B per word for the resuming emit pass
bool filter_wide_mpf_ok(const rb_batch *b, int64_t nw, int k);   // 32 <= k <= 63: may the minimizer-bucketed cache be used for this batch?
bool filter_saves_state(const rb_batch *b, int64_t nw, int k);
bool filter_saves_state_wide(const rb_batch *b, int64_t nw, int k);
void launch_hash_windows_masked(const rb_batch *b, int64_t w0, int64_t nw, int k, int mode, const uint32_t *chunk_off,
                                const uint32_t *keepmask, uint32_t first_read, uint32_t pos_bits, uint64_t *keys, uint32_t *vals,
                                hipStream_t s, const void *wstate = nullptr);

// ASCII reads -> packed batch in two halves (rb_batch.hip): begin() allocates and enqueues the copies + the
// encode kernel on `st` and returns; finish() waits for them and frees the staging buffers
struct AsciiUpload {
    rb_batch *b = nullptr;
    uint8_t *d_seq = nullptr, *d_qual = nullptr;
    int64_t *d_off = nullptr;
    std::vector<uint32_t> len;
    std::vector<int64_t> rel;
    hipStream_t st = nullptr;
    DevPool *pool = nullptr;         // chunked ingests: staging and batch arrays are taken from / handed back to the handle's pool
    IngestHost *hs = nullptr;        // ... and this begin() prepares its host-side tables in this pinned scratch (else in vectors)
    template <class T> T *dev_alloc(size_t bytes) {
        if (pool) return static_cast<T *>(pool->get(bytes));
        void *p = nullptr;
        RB_HIP(hipMalloc(&p, bytes));
        return static_cast<T *>(p);
    }
    void drop() {
        for (void *p : {(void *)d_seq, (void *)d_qual, (void *)d_off}) {
            if (!p) continue;
            if (pool) pool->put(p); else (void)hipFree(p);
        }
        d_seq = d_qual = nullptr; d_off = nullptr;
    }
};
void ascii_batch_begin(AsciiUpload &u, int device, const char *seq, const char *qual, const int64_t *offsets, int64_t first, int64_t n_reads,
                       int min_base_qual, hipStream_t st, bool want_rnz = false /* also build rb_batch::rnz (all-window hashing of raw strings) */);
rb_batch *ascii_batch_finish(AsciiUpload &u);
void ascii_batch_abort(AsciiUpload &u);
void launch_encode_ascii_piece(const uint8_t *seq, const uint8_t *qual, const int64_t *off_all, const uint32_t *woff_all, int64_t r_first, int64_t n_reads,
                               int64_t seq_base, int64_t words_ub, int min_q, uint64_t *codes, uint32_t *valid, uint32_t *word_read, hipStream_t st);

// FASTQ text -> packed batch, parsed on the GPU (rb_io.hip): uploads text[0, n) on `st`, finds the lines and the records
// there and 2-bit encodes the sequence lines in place of a host-side split.  `final`: the text ends the input (a last line
// without an end of line counts; otherwise it belongs to the caller's next piece).  consumed = bytes of the complete
// records (where the next piece starts).  Synchronises `st`.
struct FastqChunk { rb_batch *b = nullptr; size_t consumed = 0; int64_t records = 0; };
FastqChunk fastq_batch_create(int device, const char *text, size_t n, bool final, int min_base_qual, bool use_qual, hipStream_t st, DevPool *pool = nullptr);
// FASTA text (FastaReader.next semantics) the same way; *ended: an empty line in header position ended the iteration
FastqChunk fasta_batch_create(int device, const char *text, size_t n, bool final, hipStream_t st, bool *ended, DevPool *pool = nullptr);
}  // namespace rb
