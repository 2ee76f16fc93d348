// sort_bench.hip — rocPRIM onesweep radix sort of (u64 key, u32 value) pairs on key bits [32, 64), the shape of
// sort_occurrences: time per sort for different onesweep configurations (block size x items per thread, radix bits).
//   hipcc -O3 --offload-arch=gfx950 -DV_A sort_bench.hip -o sort_bench && ./sort_bench 64      (-DV_B / -DV_C: more shapes)
// Measured on MI355X, 64 Mi records: rocPRIM's tuned default 2.45 ms (= 2.85 TB/s of the 104 B/record a 4-pass sort
// moves); hand-picked shapes with the basic rank algorithm are 2-12x slower (8 bits 256x12: 10.9 ms, 7 bits 256x12:
// 6.9 ms, 6 bits 256x16: 4.1 ms); 512-thread blocks at 8 bits and any 11-bit digit do not fit the 160 KB of LDS.
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_fill(uint64_t *k, uint32_t *v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = i * 0x9E3779B97F4A7C15ull + 1; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    k[i] = z; v[i] = (uint32_t)i;
}
__global__ void k_check(const uint64_t *k, size_t n, uint32_t *bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < n && (k[i] >> 32) > (k[i + 1] >> 32)) atomicAdd(bad, 1u);
}
template <class Config> double run(uint64_t *k0, uint64_t *k1, uint32_t *v0, uint32_t *v1, size_t n, uint32_t *bad) {
    size_t tb = 0;
    CK(rocprim::radix_sort_pairs<Config>(nullptr, tb, k0, k1, v0, v1, n, 32, 64));
    void *tmp; CK(hipMalloc(&tmp, tb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(rocprim::radix_sort_pairs<Config>(tmp, tb, k0, k1, v0, v1, n, 32, 64));
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) CK(rocprim::radix_sort_pairs<Config>(tmp, tb, k0, k1, v0, v1, n, 32, 64));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, 0, k1, n, bad);
    uint32_t hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    CK(hipFree(tmp));
    if (hb) printf("  (NOT SORTED: %u inversions)\n", hb);
    return ms / 3;
}
template <unsigned BITS, unsigned BS, unsigned IPT>
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<BS, IPT>, BITS>, 0>;
#define RUN(B, S, I) printf("%u bits, %4u x %2u : %7.2f ms\n", B, S, I, run<cfg<B, S, I>>(k0, k1, v0, v1, n, bad))
int main(int argc, char **argv) {
    const size_t n = (argc > 1 ? (size_t)atol(argv[1]) : 64u) << 20;
    uint64_t *k0, *k1; uint32_t *v0, *v1, *bad;
    CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4)); CK(hipMalloc(&bad, 4));
    hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, 0, k0, v0, n);
    using dflt = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
    printf("n = %zu records (12 B each), key bits [32,64)\n", n);
    printf("default config    : %7.2f ms\n", run<dflt>(k0, k1, v0, v1, n, bad));
#ifdef V_A
    RUN(8, 256, 12); RUN(8, 256, 8); RUN(8, 256, 6); RUN(8, 256, 4);
#endif
#ifdef V_B
    RUN(8, 512, 6); RUN(8, 512, 4); RUN(8, 1024, 3); RUN(8, 128, 12); RUN(8, 128, 16);
#endif
#ifdef V_C
    RUN(7, 256, 12); RUN(7, 512, 8); RUN(6, 256, 16); RUN(6, 512, 12);
#endif
    return 0;
}
