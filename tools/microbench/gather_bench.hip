// gather_bench.hip — random-access ceiling of one MI355X: G lookups/s of independent random 8-byte /
// 64-byte-line reads as a function of table size and of the number of independent loads a thread keeps
// in flight.  Build: hipcc -O3 --offload-arch=gfx950 gather_bench.hip -o gather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
// K independent 8-byte loads per iteration, ITER iterations per thread; LINE=1: read the whole 64 B line (4 x 16 B)
template <int K, int LINE>
__global__ void k_gather(const uint64_t *__restrict__ tab, uint64_t mask, int iters, uint64_t *out) {
    uint64_t s = mix((uint64_t)blockIdx.x * blockDim.x + threadIdx.x), acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t idx[K];
#pragma unroll
        for (int q = 0; q < K; ++q) { s = mix(s); idx[q] = s & mask; }
#pragma unroll
        for (int q = 0; q < K; ++q) {
            if (LINE) {
                const ulonglong2 *b = reinterpret_cast<const ulonglong2 *>(tab + (idx[q] & ~7ull));
                ulonglong2 a0 = b[0], a1 = b[1], a2 = b[2], a3 = b[3];
                acc += a0.x ^ a0.y ^ a1.x ^ a1.y ^ a2.x ^ a2.y ^ a3.x ^ a3.y;
            } else acc += tab[idx[q]];
        }
    }
    if (acc == 0x1234567ull) out[0] = acc;
}
template <int K, int LINE> double run(const uint64_t *tab, uint64_t n, uint64_t *out, int blocks, int tpb, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<K, LINE>), dim3(blocks), dim3(tpb), 0, 0, tab, n - 1, 2, out);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<K, LINE>), dim3(blocks), dim3(tpb), 0, 0, tab, n - 1, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)blocks * tpb * iters * K / (ms * 1e-3) / 1e9;
}
// what FETCH_SIZE tallies per random request (run under rocprofv3 --pmc FETCH_SIZE: `gather_bench calib`): one lane reads 8 B,
// a whole 64-byte line (4 x 16 B) or a whole 128-byte bucket (8 x 16 B, the minimizer-bucketed prefilter cache's access) at a
// random aligned place of a 4 GB table; every kernel issues exactly 2^28 such requests
template <int BYTES>
__global__ void k_calib(const uint64_t *__restrict__ tab, uint64_t mask, int iters, uint64_t *out) {
    uint64_t s = mix((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 77), acc = 0;
    for (int it = 0; it < iters; ++it) {
        s = mix(s);
        const uint64_t idx = s & mask;
        if (BYTES == 8) acc += tab[idx];
        else {
            const ulonglong2 *b = reinterpret_cast<const ulonglong2 *>(tab + (idx & ~(uint64_t)(BYTES / 8 - 1)));
#pragma unroll
            for (int q = 0; q < BYTES / 16; ++q) { const ulonglong2 a = b[q]; acc += a.x ^ a.y; }
        }
    }
    if (acc == 0x1234567ull) out[0] = acc;
}

// ---- FETCH_SIZE on a GAPPED streaming read (round 5: what k_hash_windows_resume does) ----
// A wavefront takes 512 consecutive elements of an array, lists the ones it keeps (a fixed pseudo-random DENS percent, as the emit pass lists
// the words that keep a window) and reads those, one element of BYTES bytes per lane, 64 listed elements per load instruction.  The host
// counts what a sector- and a line-granular memory system must move for exactly that set: 64-byte sectors touched, 128-byte lines touched.
// `gather_bench gapped` under rocprofv3 --pmc FETCH_SIZE says which of them the counter follows (profiles/r05_pmc_calibration.txt).
__host__ __device__ __forceinline__ bool gap_keep(uint64_t i, uint32_t dens_pct) {
    uint64_t z = i * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    return (uint32_t)(z % 100u) < dens_pct;
}
template <int BYTES, int DENS>
__global__ void __launch_bounds__(64) k_gapped(const uint32_t *__restrict__ a, uint64_t n, uint64_t *out) {
    __shared__ uint16_t s_list[512];
    const uint64_t base = (uint64_t)blockIdx.x * 512u;
    const uint32_t lane = threadIdx.x;
    uint32_t n_list = 0;
    for (uint32_t q = 0; q < 512u; q += 64u) {
        const uint64_t i = base + q + lane;
        const bool ne = i < n && gap_keep(i, DENS);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ne);
        if (ne) s_list[n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)(q + lane);
        n_list += (uint32_t)__popcll(m);
    }
    __syncthreads();
    uint64_t acc = 0;
    for (uint32_t e0 = 0; e0 < n_list; e0 += 64u) {
        if (e0 + lane < n_list) {
            const uint64_t i = base + s_list[e0 + lane];
            const uint32_t *p = a + i * (BYTES / 4);
            if (BYTES == 16) { const uint4 v = *reinterpret_cast<const uint4 *>(p); acc += v.x ^ v.y ^ v.z ^ v.w; }
            else if (BYTES == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(p); acc += v.x ^ v.y; }
            else acc += p[0];
        }
    }
    if (acc == 0x1234567ull) out[0] = acc;
}
template <int BYTES, int DENS> static void gapped_one(const uint32_t *a, uint64_t n, uint64_t *out) {
    hipLaunchKernelGGL((k_gapped<BYTES, DENS>), dim3((unsigned)((n + 511) / 512)), dim3(64), 0, 0, a, n, out);
    CK(hipDeviceSynchronize());
    uint64_t kept = 0, sect = 0, lines = 0, last_s = ~0ull, last_l = ~0ull;
    for (uint64_t i = 0; i < n; ++i) {
        if (!gap_keep(i, DENS)) continue;
        ++kept;
        const uint64_t s64 = i * BYTES / 64, l128 = i * BYTES / 128;
        if (s64 != last_s) { ++sect; last_s = s64; }
        if (l128 != last_l) { ++lines; last_l = l128; }
    }
    printf("k_gapped<%d, %d>: %llu of %llu elements read = %.3f GB useful; %.3f GB in 64-byte sectors touched; %.3f GB in 128-byte lines touched (x 64: %.3f GB)\n", BYTES, DENS,
           (unsigned long long)kept, (unsigned long long)n, kept * (double)BYTES / 1e9, sect * 64.0 / 1e9, lines * 128.0 / 1e9, lines * 64.0 / 1e9);
}
static int gapped() {
    const uint64_t n = 1ull << 27;                      // 2 GB of 16-byte elements: past the Infinity Cache
    uint32_t *a; uint64_t *out;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 1, n * 16));
    gapped_one<16, 100>(a, n, out); gapped_one<16, 55>(a, n, out); gapped_one<16, 25>(a, n, out); gapped_one<16, 10>(a, n, out);
    gapped_one<8, 100>(a, n, out);  gapped_one<8, 55>(a, n, out);  gapped_one<8, 25>(a, n, out);
    gapped_one<4, 100>(a, n, out);  gapped_one<4, 55>(a, n, out);  gapped_one<4, 25>(a, n, out);
    return 0;
}
static int calib() {
    uint64_t *out, *tab; const uint64_t n = 1ull << 29;            // 4 GB
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&tab, n * 8)); CK(hipMemset(tab, 1, n * 8));
    const int blocks = 1 << 14, tpb = 256, iters = 64;             // 2^14 * 2^8 * 2^6 = 2^28 requests per kernel
    hipLaunchKernelGGL((k_calib<8>), dim3(blocks), dim3(tpb), 0, 0, tab, n - 1, iters, out);
    hipLaunchKernelGGL((k_calib<64>), dim3(blocks), dim3(tpb), 0, 0, tab, n - 1, iters, out);
    hipLaunchKernelGGL((k_calib<128>), dim3(blocks), dim3(tpb), 0, 0, tab, n - 1, iters, out);
    CK(hipDeviceSynchronize());
    printf("issued %llu requests per kernel (k_calib<8>, <64>, <128>)\n", (unsigned long long)blocks * tpb * iters);
    return 0;
}
// `gather_bench big`: one 8 GB table (the size of config 2's counting filter), random 8-byte reads and random byte-granular atomicOr:
// run it several times — each process gets its own placement of the table — to see how much of the run-to-run spread of the
// insert pipeline's probe / resolve stages is the placement
__global__ void k_scatter_or(uint32_t *__restrict__ tab, uint64_t mask, int iters, uint32_t *out) {
    uint64_t s = mix((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 5), acc = 0;
    for (int it = 0; it < iters; ++it) {
        s = mix(s);
        acc += atomicOr(&tab[s & mask], 0x80u);
    }
    if (acc == 0x1234567ull) out[0] = (uint32_t)acc;
}
// the same addresses first read, then OR-ed by a second kernel (what k_probe_h2 + k_set_bits do to the Bloom bits), and the
// OR without a use of the old value
template <int MODE /* 0 read, 1 returning atomicOr, 2 atomicOr whose result is not used */>
__global__ void k_touch(uint32_t *__restrict__ tab, uint64_t mask, int iters, uint32_t *out) {
    uint64_t s = mix((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 5), acc = 0;
    for (int it = 0; it < iters; ++it) {
        s = mix(s);
        if (MODE == 0) acc += tab[s & mask];
        else if (MODE == 1) acc += atomicOr(&tab[s & mask], 0x80u);
        else atomicOr(&tab[s & mask], 0x40u);
    }
    if (acc == 0x1234567ull) out[0] = (uint32_t)acc;
}
template <int MODE> static double touch(uint32_t *tab, uint64_t words, uint32_t *out, int blocks, int tpb, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_touch<MODE>, dim3(blocks), dim3(tpb), 0, 0, tab, words - 1, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)blocks * tpb * iters / (ms * 1e-3) / 1e9;
}
// n touches at SORTED random places (touch i lies in the i-th of n equal slices of the table, so consecutive lanes go to
// increasing addresses a few lines apart) against the same number at unsorted places: what ordering the runs of a sub-batch by
// counter index would buy the claims
template <int MODE /* 0 read, 1 returning atomicOr */, int SORTED>
__global__ void k_touch_n(uint32_t *__restrict__ tab, uint64_t words, uint64_t n, uint32_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t slice = words / n, r = mix(i + 99);
    const uint64_t idx = SORTED ? i * slice + r % slice : r % words;
    const uint32_t v = MODE == 0 ? tab[idx] : atomicOr(&tab[idx], 0x80u);
    if (v == 0x12345u) out[0] = v;
}
template <int MODE, int SORTED> static double touch_n(uint32_t *tab, uint64_t words, uint64_t n, uint32_t *out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_touch_n<MODE, SORTED>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, tab, words, n, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)n / (ms * 1e-3) / 1e9;
}
static int big(int log2_gb, int alloc_flag = 0) {
    uint64_t *out, *tab; const uint64_t n = 1ull << (27 + log2_gb);            // 8 GB by default
    CK(hipMalloc(&out, 64));
    // `big <log2 GB> <flag>`: 1 = hipDeviceMallocFinegrained, 3 = hipDeviceMallocUncached, 4 = hipDeviceMallocContiguous
    if (alloc_flag) { CK(hipExtMallocWithFlags((void **)&tab, n * 8, (unsigned)alloc_flag)); printf("(allocation flag %d) ", alloc_flag); }
    else CK(hipMalloc(&tab, n * 8));
    CK(hipMemset(tab, 1, n * 8));
    const int blocks = 256 * 32, tpb = 256;
    const double rd = run<4, 0>(tab, n, out, blocks, tpb, 64);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_scatter_or, dim3(blocks), dim3(tpb), 0, 0, (uint32_t *)tab, 2 * n - 1, 2, (uint32_t *)out);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_scatter_or, dim3(blocks), dim3(tpb), 0, 0, (uint32_t *)tab, 2 * n - 1, 128, (uint32_t *)out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    {
        const uint64_t words = 2 * n;
        const int b2 = 1 << 16, it2 = 4;                                   // 64 M touches per kernel, as one sub-batch
        touch<0>((uint32_t *)tab, words, (uint32_t *)out, b2, tpb, it2);
        const double r0 = touch<0>((uint32_t *)tab, words, (uint32_t *)out, 1 << 14, tpb, 64);
        const double r1 = touch<1>((uint32_t *)tab, words, (uint32_t *)out, b2, tpb, it2);     // other addresses than the read before? no: same seeds -> same addresses as the first touch<0>
        const double r2 = touch<2>((uint32_t *)tab, words, (uint32_t *)out, 1 << 14, tpb, 64);
        const double r3 = touch<1>((uint32_t *)tab, words, (uint32_t *)out, 1 << 14, tpb, 64);
        printf("  4-byte words: reads %.1f G/s; returning atomicOr on 64 M addresses read two kernels earlier %.1f G/s; atomicOr, result unused %.1f G/s; returning %.1f G/s\n", r0, r1, r2, r3);
    }
    for (uint64_t m : {28000000ull, 112000000ull, 448000000ull}) {
        const uint64_t words = 2 * n;
        touch_n<0, 0>((uint32_t *)tab, words, m, (uint32_t *)out);
        printf("  %llu M touches of 4-byte words: reads unsorted %.1f / sorted %.1f G/s; returning atomicOr unsorted %.1f / sorted %.1f G/s\n",
               (unsigned long long)(m / 1000000), touch_n<0, 0>((uint32_t *)tab, words, m, (uint32_t *)out), touch_n<0, 1>((uint32_t *)tab, words, m, (uint32_t *)out),
               touch_n<1, 0>((uint32_t *)tab, words, m, (uint32_t *)out), touch_n<1, 1>((uint32_t *)tab, words, m, (uint32_t *)out));
    }
    printf("%d GB table: random 8-byte reads %.1f G/s, random returning atomicOr %.1f G/s\n", 1 << log2_gb, rd, (double)blocks * tpb * 128 / (ms * 1e-3) / 1e9);
    return 0;
}
int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "calib")) return calib();
    if (argc > 1 && !strcmp(argv[1], "gapped")) return gapped();
    if (argc > 1 && !strcmp(argv[1], "big")) return big(argc > 2 ? atoi(argv[2]) : 3, argc > 3 ? atoi(argv[3]) : 0);
    uint64_t *out; CK(hipMalloc(&out, 64));
    const int blocks = 256 * 32, tpb = 256;
    printf("%-10s %-6s %8s %8s %8s %8s\n", "table", "what", "K=1", "K=2", "K=4", "K=8");
    for (int l2 = 20; l2 <= 28; l2 += 2) {          // entries of 8 B: 8 MB .. 2 GB
        uint64_t n = 1ull << l2, *tab;
        CK(hipMalloc(&tab, n * 8)); CK(hipMemset(tab, 1, n * 8));
        const int iters = 64;
        printf("%6.0f MB  8B     %8.1f %8.1f %8.1f %8.1f\n", n * 8 / 1048576.0, run<1, 0>(tab, n, out, blocks, tpb, iters), run<2, 0>(tab, n, out, blocks, tpb, iters),
               run<4, 0>(tab, n, out, blocks, tpb, iters), run<8, 0>(tab, n, out, blocks, tpb, iters));
        printf("%6.0f MB  line   %8.1f %8.1f %8.1f %8.1f\n", n * 8 / 1048576.0, run<1, 1>(tab, n, out, blocks, tpb, iters), run<2, 1>(tab, n, out, blocks, tpb, iters),
               run<4, 1>(tab, n, out, blocks, tpb, iters), run<8, 1>(tab, n, out, blocks, tpb, iters));
        CK(hipFree(tab));
    }
    return 0;
}
