// mix_bench.hip — does a small L2-resident first-level table pay?  Every lookup probes a SMALL table (one
// 64 B line); a fraction `miss` of them then probes a 1 GB table.  Reports G lookups/s against the plain
// single-level rate.  Build: hipcc -O3 --offload-arch=gfx950 mix_bench.hip -o mix_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void k_mix(const uint64_t *__restrict__ small, uint64_t smask, const uint64_t *__restrict__ big, uint64_t bmask,
                      uint32_t miss_1024, int iters, uint64_t *out) {
    uint64_t s = mix((uint64_t)blockIdx.x * blockDim.x + threadIdx.x), acc = 0;
    for (int it = 0; it < iters; ++it) {
        s = mix(s);
        if (smask) {
            const ulonglong2 *b = reinterpret_cast<const ulonglong2 *>(small + ((s >> 20) & smask & ~7ull));
            ulonglong2 a0 = b[0], a1 = b[1], a2 = b[2], a3 = b[3];
            acc += a0.x ^ a0.y ^ a1.x ^ a1.y ^ a2.x ^ a2.y ^ a3.x ^ a3.y;
        }
        if ((uint32_t)(s & 1023u) < miss_1024) {
            const ulonglong2 *b = reinterpret_cast<const ulonglong2 *>(big + ((s >> 12) & bmask & ~7ull));
            ulonglong2 a0 = b[0], a1 = b[1], a2 = b[2], a3 = b[3];
            acc += a0.x ^ a0.y ^ a1.x ^ a1.y ^ a2.x ^ a2.y ^ a3.x ^ a3.y;
        }
    }
    if (acc == 0x1234567ull) out[0] = acc;
}
int main() {
    uint64_t *out, *big; CK(hipMalloc(&out, 64));
    const uint64_t nb = 1ull << 27; CK(hipMalloc(&big, nb * 8)); CK(hipMemset(big, 1, nb * 8));
    const int blocks = 256 * 32, tpb = 256, iters = 64;
    printf("%-12s %-10s %10s\n", "small table", "miss rate", "G lookups/s");
    for (int l2 = 0; l2 <= 22; l2 += (l2 == 0 ? 16 : 2)) {      // 0 = no first level, then 512 KB .. 32 MB
        uint64_t ns = l2 ? (1ull << l2) : 8, *small; CK(hipMalloc(&small, ns * 8)); CK(hipMemset(small, 1, ns * 8));
        for (uint32_t miss : {1024u, 768u, 512u, 256u}) {
            if (!l2 && miss != 1024u) continue;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(tpb), 0, 0, small, l2 ? ns - 1 : 0, big, nb - 1, miss, 2, out);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(tpb), 0, 0, small, l2 ? ns - 1 : 0, big, nb - 1, miss, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%8.1f MB  %8.2f   %10.1f\n", l2 ? ns * 8 / 1048576.0 : 0.0, miss / 1024.0, (double)blocks * tpb * iters / (ms * 1e-3) / 1e9);
        }
        CK(hipFree(small));
    }
    return 0;
}
