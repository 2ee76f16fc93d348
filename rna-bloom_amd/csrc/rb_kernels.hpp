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
// total usable windows are accumulated into total_spread[16*q], q < 32
void launch_filter_windows(const rb_batch *b, int64_t w0, int64_t nw, int k, int mode, uint32_t first_read,
                           uint32_t pos_bits, uint64_t seed, uint64_t ordinal0, Npf cache, uint32_t *cnt, uint32_t *keepmask,
                           uint32_t *total_spread, hipStream_t s);
void launch_hash_windows_masked(const rb_batch *b, int64_t w0, int64_t nw, int k, int mode, const uint32_t *chunk_off,
                                const uint32_t *keepmask, uint32_t first_read, uint32_t pos_bits, uint64_t *keys, uint32_t *vals,
                                hipStream_t s);

}  // namespace rb
