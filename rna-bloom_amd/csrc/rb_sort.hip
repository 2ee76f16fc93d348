// rb_sort.hip — device-wide sort / scan / run-length primitives (rocPRIM), isolated in their own
// translation unit because the templates are slow to compile.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "rb_internal.hpp"

namespace rb {

// rocPRIM (ROCm 7.2) routes 4K < n <= 1M through a merge-sort sub-algorithm that returns WRONGLY
// ORDERED output when begin_bit > 0 (verified on gfx950: bad order + instability at n = 250000,
// begin_bit = 32; onesweep and the single-block sort are correct and stable).  MergeSortLimit = 0
// disables that sub-algorithm.
using sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                               rocprim::default_config, 0>;

size_t sort_pairs_temp_bytes(size_t n) {
    size_t bytes = 0;
    RB_HIP(rocprim::radix_sort_pairs<sort_config>(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                     (uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 64));
    return bytes;
}
void sort_pairs_u64_u32(void *temp, size_t temp_bytes, uint64_t *keys_in, uint64_t *keys_out,
                        uint32_t *vals_in, uint32_t *vals_out, size_t n, int begin_bit, int end_bit,
                        hipStream_t s) {
    RB_HIP(rocprim::radix_sort_pairs<sort_config>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n,
                                     (unsigned)begin_bit, (unsigned)end_bit, s));
}
size_t sort_pairs32_temp_bytes(size_t n) {
    size_t bytes = 0;
    RB_HIP(rocprim::radix_sort_pairs<sort_config>(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                     (uint64_t *)nullptr, (uint64_t *)nullptr, n, 0, 64));
    return bytes;
}
void sort_pairs_u64_u64(void *temp, size_t temp_bytes, uint64_t *keys_in, uint64_t *keys_out,
                        uint64_t *vals_in, uint64_t *vals_out, size_t n, int begin_bit, int end_bit,
                        hipStream_t s) {
    RB_HIP(rocprim::radix_sort_pairs<sort_config>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n,
                                     (unsigned)begin_bit, (unsigned)end_bit, s));
}
size_t sort_keys_temp_bytes(size_t n) {
    size_t bytes = 0;
    RB_HIP(rocprim::radix_sort_keys<sort_config>(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, n, 0, 64));
    return bytes;
}
void sort_keys_u64(void *temp, size_t temp_bytes, uint64_t *keys_in, uint64_t *keys_out, size_t n, int begin_bit,
                   int end_bit, hipStream_t s) {
    RB_HIP(rocprim::radix_sort_keys<sort_config>(temp, temp_bytes, keys_in, keys_out, n, (unsigned)begin_bit, (unsigned)end_bit, s));
}
size_t scan_temp_bytes(size_t n) {
    size_t bytes = 0;
    RB_HIP(rocprim::exclusive_scan(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                   0u, n, rocprim::plus<uint32_t>()));
    return bytes;
}
void exclusive_scan_u32(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, size_t n,
                        hipStream_t s) {
    RB_HIP(rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s));
}
struct MaskFlag {
    uint32_t mask;
    __host__ __device__ bool operator()(uint32_t st) const { return (st & mask) != 0u; }
};
size_t select_temp_bytes(size_t n) {
    size_t bytes = 0;
    auto flags = rocprim::make_transform_iterator((const uint32_t *)nullptr, MaskFlag{1u});
    RB_HIP(rocprim::select(nullptr, bytes, rocprim::counting_iterator<uint32_t>(0), flags, (uint32_t *)nullptr,
                           (uint32_t *)nullptr, n));
    return bytes;
}
void select_flagged(void *temp, size_t temp_bytes, const uint32_t *status, uint32_t mask, size_t n, uint32_t *out,
                    uint32_t *count_dev, hipStream_t s) {
    auto flags = rocprim::make_transform_iterator(status, MaskFlag{mask});
    RB_HIP(rocprim::select(temp, temp_bytes, rocprim::counting_iterator<uint32_t>(0), flags, out, count_dev, n, s));
}
size_t rle_temp_bytes(size_t n) {
    size_t bytes = 0;
    RB_HIP(rocprim::run_length_encode(nullptr, bytes, (const uint64_t *)nullptr, n,
                                      (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr));
    return bytes;
}
void run_length_encode_u64(void *temp, size_t temp_bytes, const uint64_t *keys, size_t n,
                           uint64_t *uniq, uint32_t *counts, uint32_t *n_runs_dev, hipStream_t s) {
    RB_HIP(rocprim::run_length_encode(temp, temp_bytes, keys, n, uniq, counts, n_runs_dev, s));
}

}  // namespace rb
