// rb_sort.hip — device-wide scan / selection primitives, written for gfx950 (no library code on the insert path).
//
// Rounds 1-3 called rocPRIM here.  Its decoupled look-back scan spins: a block waits for the published prefix of its
// predecessors, and when the device is shared with another stream's kernels (the producer's window walk runs beside every
// scan of the consumer) the predecessors may not even be resident — scans of 20 k-75 k threads took 1-3.5 ms each, 27.8 ms of
// kernel time per step (profiles/r03_kernel_stats.csv, gpurun_out/r03_scans.txt).  The scans below never wait for another
// workgroup: reduce per tile -> one workgroup scans the tile sums -> apply (the input is read twice instead of once, which
// the arrays in question — at most a few hundred MB, mostly a few MB — do not notice), and arrays of up to 64 K entries are
// scanned by a single workgroup in one launch (16 K entries per round of four barriers).  The radix sorts of the conflict path are LSD passes of the grouping stage's
// own stable partition kernels (rb_group.hip: lsd_sort_*).
#include <cstring>

#include "rb_internal.hpp"

namespace rb {

namespace {
constexpr uint32_t SC_TPB = 256, SC_VEC = 4, SC_SUB = SC_TPB * SC_VEC, SC_SUBS = 16;   // 16384 items per tile.  (16 items per thread and
// four rounds instead of sixteen measured TWICE as slow, 233 against 116 us for the 23 M-word chunk scan: a thread's 64 consecutive bytes put the lanes' uint4 reads
// on the same LDS banks; at 4 items the blocked reads are conflict-free)
constexpr uint32_t SC_ONE_TPB = 512, SC_ONE_VEC = 16, SC_SMALL_VEC = 16, SC_ONE_MAX = 65536;   // single-workgroup path: 8 K items per round of 4 barriers.
// (512 threads and 35 KB of LDS, not 1024 and 70: a single-workgroup kernel of the consumer stream has to find room on a CU beside the producer's
// persistent bucket kernel — three workgroups of 512 threads and 40 KB per CU for milliseconds; what does not fit beside them waits for that kernel to
// END, 0.2-0.6 ms per call on average where the two overlap)

// One sub-tile of TPB x VEC items at in[base ...): exclusive scan with `carry` added, written to out; returns carry + the
// sub-tile's sum.  Striped (coalesced) global accesses, blocked scan through LDS; no alignment assumptions; in == out is fine.
// LDS index of logical item i: four words of padding after every 64, so that the lanes' blocked 16-byte accesses (a thread's VEC consecutive
// items) fall on different banks for any VEC (unpadded, 16 items per thread put every fourth lane on the same banks: the scan ran twice as long)
__device__ __forceinline__ uint32_t sc_pad(uint32_t i) { return i + ((i >> 6) << 2); }
constexpr uint32_t sc_padded(uint32_t n) { return n + ((n >> 6) << 2) + 4u; }
template <uint32_t TPB, uint32_t VEC>
__device__ __forceinline__ uint32_t sc_sub_scan(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t base, size_t n, uint32_t carry,
                                                uint32_t *s_data /* [sc_padded(TPB * VEC)], 16-byte aligned */, uint32_t *s_wsum /* [TPB / 64] */) {
    static_assert(VEC % 4 == 0, "whole uint4 per thread");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
#pragma unroll
    for (uint32_t i = 0; i < VEC; ++i) {
        const size_t x = base + (size_t)i * TPB + tid;
        s_data[sc_pad(i * TPB + tid)] = x < n ? in[x] : 0u;
    }
    __syncthreads();
    uint32_t e[VEC];                                   // exclusive prefix inside the thread's VEC consecutive items
    uint32_t tot = 0;
#pragma unroll
    for (uint32_t q = 0; q < VEC / 4; ++q) {
        const uint4 v = *reinterpret_cast<const uint4 *>(s_data + sc_pad(tid * VEC + 4u * q));
        e[4 * q] = tot; tot += v.x; e[4 * q + 1] = tot; tot += v.y; e[4 * q + 2] = tot; tot += v.z; e[4 * q + 3] = tot; tot += v.w;
    }
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += t;
    }
    if (lane == 63u) s_wsum[w] = inc;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (uint32_t i = 0; i < TPB / 64; ++i) {
        const uint32_t t = s_wsum[i];
        if (i < w) wbase += t;
        total += t;
    }
    const uint32_t b0 = carry + wbase + inc - tot;
#pragma unroll
    for (uint32_t q = 0; q < VEC / 4; ++q)
        *reinterpret_cast<uint4 *>(s_data + sc_pad(tid * VEC + 4u * q)) = make_uint4(b0 + e[4 * q], b0 + e[4 * q + 1], b0 + e[4 * q + 2], b0 + e[4 * q + 3]);
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < VEC; ++i) {
        const size_t x = base + (size_t)i * TPB + tid;
        if (x < n) out[x] = s_data[sc_pad(i * TPB + tid)];
    }
    __syncthreads();                                   // s_data / s_wsum are reused by the caller's next sub-tile
    return carry + total;
}

__global__ void __launch_bounds__(SC_ONE_TPB) k_scan_one(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n) {
    __shared__ __attribute__((aligned(16))) uint32_t s_data[sc_padded(SC_ONE_TPB * SC_ONE_VEC)];
    __shared__ uint32_t s_wsum[SC_ONE_TPB / 64];
    uint32_t carry = 0;
    for (size_t base = 0; base < n; base += SC_ONE_TPB * SC_ONE_VEC) carry = sc_sub_scan<SC_ONE_TPB, SC_ONE_VEC>(in, out, base, n, carry, s_data, s_wsum);
}
// the same with 256 threads for arrays of a few thousand entries (a 1024-thread workgroup is mostly barrier there)
__global__ void __launch_bounds__(SC_TPB) k_scan_one_small(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n) {
    __shared__ __attribute__((aligned(16))) uint32_t s_data[sc_padded(SC_TPB * SC_SMALL_VEC)];
    __shared__ uint32_t s_wsum[SC_TPB / 64];
    uint32_t carry = 0;
    for (size_t base = 0; base < n; base += SC_TPB * SC_SMALL_VEC) carry = sc_sub_scan<SC_TPB, SC_SMALL_VEC>(in, out, base, n, carry, s_data, s_wsum);
}

__global__ void __launch_bounds__(SC_TPB) k_scan_reduce(const uint32_t *__restrict__ in, size_t n, uint32_t *__restrict__ tile_sums, uint32_t tile) {
    __shared__ uint32_t s_w[SC_TPB / 64];
    const size_t base = (size_t)blockIdx.x * tile;
    uint32_t sum = 0;
#pragma unroll 4
    for (uint32_t j = 0; j < tile / SC_TPB; ++j) {
        const size_t x = base + (size_t)j * SC_TPB + threadIdx.x;
        sum += x < n ? in[x] : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t q = 0; q < SC_TPB / 64; ++q) t += s_w[q];
        tile_sums[blockIdx.x] = t;
    }
}

// in and out 16-byte aligned (the pipeline's per-word arrays are): a thread's four items are ONE 16-byte access each way, nothing is staged
// in LDS but the wavefront sums — half the instructions and two barriers less per 1024 items
__global__ void __launch_bounds__(SC_TPB) k_scan_apply_v4(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n,
                                                          const uint32_t *__restrict__ tile_offs, uint32_t tile) {
    __shared__ uint32_t s_wsum[2][SC_TPB / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const size_t base = (size_t)blockIdx.x * tile;
    uint32_t carry = tile_offs[blockIdx.x];
    for (uint32_t q = 0; q < tile / (SC_TPB * 4u); ++q) {
        const size_t x = base + (size_t)q * (SC_TPB * 4u) + (size_t)tid * 4u;
        if (base + (size_t)q * (SC_TPB * 4u) >= n) break;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (x + 3 < n) v = *reinterpret_cast<const uint4 *>(in + x);
        else if (x < n) { v.x = in[x]; if (x + 1 < n) v.y = in[x + 1]; if (x + 2 < n) v.z = in[x + 2]; }
        const uint32_t e1 = v.x, e2 = e1 + v.y, e3 = e2 + v.z, tot = e3 + v.w;
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((int)lane >= o) inc += t; }
        if (lane == 63u) s_wsum[q & 1u][w] = inc;
        __syncthreads();                               // (double-buffered sums: one barrier per round)
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (uint32_t i = 0; i < SC_TPB / 64; ++i) { const uint32_t t = s_wsum[q & 1u][i]; if (i < w) wbase += t; total += t; }
        const uint32_t b0 = carry + wbase + inc - tot;
        if (x + 3 < n) *reinterpret_cast<uint4 *>(out + x) = make_uint4(b0, b0 + e1, b0 + e2, b0 + e3);
        else if (x < n) { out[x] = b0; if (x + 1 < n) out[x + 1] = b0 + e1; if (x + 2 < n) out[x + 2] = b0 + e2; }
        carry += total;
    }
}
__global__ void __launch_bounds__(SC_TPB) k_scan_apply(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n,
                                                       const uint32_t *__restrict__ tile_offs, uint32_t tile) {
    __shared__ __attribute__((aligned(16))) uint32_t s_data[sc_padded(SC_SUB)];
    __shared__ uint32_t s_wsum[SC_TPB / 64];
    const size_t base = (size_t)blockIdx.x * tile;
    uint32_t carry = tile_offs[blockIdx.x];
    for (uint32_t q = 0; q < tile / SC_SUB; ++q) {
        const size_t b = base + (size_t)q * SC_SUB;
        if (b >= n) break;
        carry = sc_sub_scan<SC_TPB, SC_VEC>(in, out, b, n, carry, s_data, s_wsum);
    }
}
void scan_one(const uint32_t *in, uint32_t *out, size_t n, hipStream_t s) {
    if (n <= 2 * SC_TPB * SC_SMALL_VEC) hipLaunchKernelGGL(k_scan_one_small, dim3(1), dim3(SC_TPB), 0, s, in, out, n);
    else hipLaunchKernelGGL(k_scan_one, dim3(1), dim3(SC_ONE_TPB), 0, s, in, out, n);
}
}  // namespace

size_t scan_temp_bytes(size_t n) {           // tile sums of the finest tiling (1024 items) + the same again for the level above them
    const size_t l1 = (n + SC_SUB - 1) / SC_SUB + 128, l2 = (l1 + SC_SUB - 1) / SC_SUB + 128;
    return (l1 + l2 + 4096) * 4;
}

// out[i] = in[0] + ... + in[i-1] (wrapping u32); in == out allowed.  Never waits for another workgroup.
void exclusive_scan_u32(void *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, size_t n, hipStream_t s) {
    if (n == 0) return;
    if (n <= SC_ONE_MAX) { scan_one(in, out, n, s); return; }
    // a tile is 1 ... 16 rounds of 1024 items: mid-sized arrays (a few million entries) get enough workgroups to fill the device instead of a
    // hundred workgroups walking sixteen rounds each (178 us for 2 M entries, the latency of the rounds, not bandwidth)
    const uint32_t rounds = (uint32_t)std::max<size_t>(1, std::min<size_t>(SC_SUBS, n / ((size_t)2048 * SC_SUB)));
    const uint32_t tile = rounds * SC_SUB;
    const size_t nt = (n + tile - 1) / tile;
    RB_REQUIRE(temp && temp_bytes >= scan_temp_bytes(n), "exclusive_scan_u32: temp too small");
    RB_REQUIRE(nt < (1ull << 31), "exclusive_scan_u32: too many items");
    uint32_t *sums = reinterpret_cast<uint32_t *>(temp);
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nt), dim3(SC_TPB), 0, s, in, n, sums, tile);
    const size_t used = ((nt + 63) / 64) * 64 + 64;
    exclusive_scan_u32(sums + used, temp_bytes - used * 4, sums, sums, nt, s);                     // (one workgroup up to 64 K tile sums; else one more level)
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0)
        hipLaunchKernelGGL(k_scan_apply_v4, dim3((unsigned)nt), dim3(SC_TPB), 0, s, in, out, n, sums, tile);
    else
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nt), dim3(SC_TPB), 0, s, in, out, n, sums, tile);
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
    if (blockIdx.x == 0 && threadIdx.x == 0) { count_dev[0] = tot_a; if (mask_b) count_dev[1] = offs[2u * nblk] - tot_a; }
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
// one list: indices i with (status[i] & mask) != 0, in order; count to *count_dev
size_t select_temp_bytes(size_t n) { return select2_temp_bytes(n); }
void select_flagged(void *temp, size_t temp_bytes, const uint32_t *status, uint32_t mask, size_t n, uint32_t *out,
                    uint32_t *count_dev, hipStream_t s) {
    if (n == 0) { RB_HIP(hipMemsetAsync(count_dev, 0, 4, s)); return; }
    select_flagged2(temp, temp_bytes, status, n, mask, out, 0u, out, count_dev, s);
}
}  // namespace rb
