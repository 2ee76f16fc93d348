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
// Two ordered index lists from one status array in two light passes (count per block, scan, write): the heavy and the
// conflicting runs of a sub-batch are a few percent of the runs, and rocprim::select through a transform iterator
// spent 0.75 ms per list on 36 M status words (190 GB/s) — this reads the words twice at full bandwidth instead.
namespace {
constexpr uint32_t S2_TPB = 256, S2_ITEMS = 8, S2_TILE = S2_TPB * S2_ITEMS;
__global__ void __launch_bounds__(S2_TPB) k_select2_count(const uint32_t *__restrict__ status, size_t n, uint32_t mask_a, uint32_t mask_b,
                                                          uint32_t nblk, uint32_t *__restrict__ counts) {
    __shared__ uint32_t s_a[S2_TPB / 64], s_b[S2_TPB / 64];
    const size_t base = (size_t)blockIdx.x * S2_TILE;
    uint32_t ca = 0, cb = 0;
    for (uint32_t i = 0; i < S2_ITEMS; ++i) {
        const size_t x = base + (size_t)i * S2_TPB + threadIdx.x;
        const uint32_t st = x < n ? status[x] : 0u;
        ca += (st & mask_a) != 0u; cb += (st & mask_b) != 0u;
    }
    for (int o = 32; o > 0; o >>= 1) { ca += __shfl_down(ca, o, 64); cb += __shfl_down(cb, o, 64); }
    if ((threadIdx.x & 63u) == 0) { s_a[threadIdx.x >> 6] = ca; s_b[threadIdx.x >> 6] = cb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ta = 0, tb = 0;
        for (uint32_t w = 0; w < S2_TPB / 64; ++w) { ta += s_a[w]; tb += s_b[w]; }
        counts[blockIdx.x] = ta; counts[nblk + blockIdx.x] = tb;
        if (blockIdx.x == 0) counts[2u * nblk] = 0u;
    }
}
__global__ void __launch_bounds__(S2_TPB) k_select2_write(const uint32_t *__restrict__ status, size_t n, uint32_t mask_a, uint32_t mask_b,
                                                          uint32_t nblk, const uint32_t *__restrict__ offs, uint32_t *__restrict__ out_a,
                                                          uint32_t *__restrict__ out_b, uint32_t *__restrict__ count_dev) {
    constexpr uint32_t NW = S2_TPB / 64, NS = S2_ITEMS * NW;       // (item row, wavefront) segments in output order
    __shared__ uint32_t s_a[NS], s_b[NS];
    const size_t base = (size_t)blockIdx.x * S2_TILE;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t st[S2_ITEMS];
    unsigned long long ba[S2_ITEMS], bb[S2_ITEMS];
    for (uint32_t i = 0; i < S2_ITEMS; ++i) {
        const size_t x = base + (size_t)i * S2_TPB + threadIdx.x;
        st[i] = x < n ? status[x] : 0u;
        ba[i] = __builtin_amdgcn_ballot_w64((st[i] & mask_a) != 0u);
        bb[i] = __builtin_amdgcn_ballot_w64((st[i] & mask_b) != 0u);
        if (lane == 0) { s_a[i * NW + wave] = (uint32_t)__popcll(ba[i]); s_b[i * NW + wave] = (uint32_t)__popcll(bb[i]); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                         // exclusive prefix over the NS segments
        uint32_t ra = 0, rb = 0;
        for (uint32_t q = 0; q < NS; ++q) { const uint32_t a = s_a[q], b = s_b[q]; s_a[q] = ra; s_b[q] = rb; ra += a; rb += b; }
    }
    __syncthreads();
    const uint32_t oa = offs[blockIdx.x], tot_a = offs[nblk], ob = offs[nblk + blockIdx.x] - tot_a;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t i = 0; i < S2_ITEMS; ++i) {
        const uint32_t x = (uint32_t)(base + (size_t)i * S2_TPB + threadIdx.x);
        if (st[i] & mask_a) out_a[oa + s_a[i * NW + wave] + (uint32_t)__popcll(ba[i] & below)] = x;
        if (st[i] & mask_b) out_b[ob + s_b[i * NW + wave] + (uint32_t)__popcll(bb[i] & below)] = x;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { count_dev[0] = tot_a; count_dev[1] = offs[2u * nblk] - tot_a; }
}
}  // namespace
size_t select2_temp_bytes(size_t n) {
    const size_t nblk = (n + S2_TILE - 1) / S2_TILE;
    return 2 * ((2 * nblk + 1) * 4 + 256) + scan_temp_bytes(2 * nblk + 1);
}
void select_flagged2(void *temp, size_t temp_bytes, const uint32_t *status, size_t n, uint32_t mask_a, uint32_t *out_a,
                     uint32_t mask_b, uint32_t *out_b, uint32_t *count_dev, hipStream_t s) {
    const uint32_t nblk = (uint32_t)((n + S2_TILE - 1) / S2_TILE);
    if (nblk == 0) { RB_HIP(hipMemsetAsync(count_dev, 0, 8, s)); return; }
    const size_t arr = (((size_t)2 * nblk + 1) * 4 + 255) / 256 * 256;
    uint32_t *counts = reinterpret_cast<uint32_t *>(temp), *offs = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(temp) + arr);
    void *scan_tmp = reinterpret_cast<char *>(temp) + 2 * arr;
    hipLaunchKernelGGL(k_select2_count, dim3(nblk), dim3(S2_TPB), 0, s, status, n, mask_a, mask_b, nblk, counts);
    exclusive_scan_u32(scan_tmp, temp_bytes - 2 * arr, counts, offs, (size_t)2 * nblk + 1, s);
    hipLaunchKernelGGL(k_select2_write, dim3(nblk), dim3(S2_TPB), 0, s, status, n, mask_a, mask_b, nblk, offs, out_a, out_b, count_dev);
}
}  // namespace rb
