// probe_order_bench.hip — what would stage A (k_probe_h2) gain if the runs of a sub-batch arrived ordered by their FIRST filter
// index instead of by hash?  Every run touches 2 Bloom-bit words (loads) and 2 counter words (returning atomicOr); with the
// grouping partitioned on index_of(h0) instead of on the hash's top bits, the first of each pair would sweep its filter window
// by window (a bucket of ~3000 records covers 1/nbuckets of the index range) while the second stays random.
//   mode 0: all four random (today)          mode 1: first counter windowed      mode 2: first counter and first bit windowed
// Windows are handed out the way the bucket kernel's output is consumed: thread t's runs belong to window t * RUNS / runs_per_window.
// Build: hipcc -O3 --offload-arch=gfx950 probe_order_bench.hip -o probe_order_bench;   run: ./probe_order_bench [runs_M [cbf_MB [dbg_MB]]]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t scale(uint64_t x, uint64_t n) { return (uint64_t)(((unsigned __int128)x * n) >> 64); }

template <int RUNS>
__global__ void __launch_bounds__(256) k_probe(uint32_t *__restrict__ cbf, uint64_t cbf_words, const uint32_t *__restrict__ dbg, uint64_t dbg_words,
                                               uint64_t n_runs, uint64_t n_windows, int mode, uint32_t claim, uint64_t *__restrict__ out) {
    const uint64_t d0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * RUNS;
    uint64_t bi[RUNS][2], ci[RUNS][2];
    uint32_t w[RUNS][2];
#pragma unroll
    for (int r = 0; r < RUNS; ++r) {
        const uint64_t d = d0 + r, h0 = mix(d), h1 = mix(h0), h2 = mix(h1), h3 = mix(h2);
        const uint64_t win = d * n_windows / n_runs;                                  // the bucket this run came out of
        ci[r][0] = mode >= 1 ? win * (cbf_words / n_windows) + scale(h0, cbf_words / n_windows) : scale(h0, cbf_words);
        ci[r][1] = scale(h1, cbf_words);
        bi[r][0] = mode >= 2 ? win * (dbg_words / n_windows) + scale(h0, dbg_words / n_windows) : scale(h2, dbg_words);
        bi[r][1] = scale(h3, dbg_words);
    }
#pragma unroll
    for (int r = 0; r < RUNS; ++r) { w[r][0] = d0 + r < n_runs ? dbg[bi[r][0]] : 0u; w[r][1] = d0 + r < n_runs ? dbg[bi[r][1]] : 0u; }
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < RUNS; ++r) acc += w[r][0] ^ w[r][1];
    uint32_t b[RUNS][2];
#pragma unroll
    for (int r = 0; r < RUNS; ++r) {
        const bool live = d0 + r < n_runs && acc != 0x9E3779B9u;                          // (the claims depend on the loads, as in the kernel)
        b[r][0] = live ? atomicOr(&cbf[ci[r][0]], claim) : 0u;
        b[r][1] = live ? atomicOr(&cbf[ci[r][1]], claim) : 0u;
    }
#pragma unroll
    for (int r = 0; r < RUNS; ++r) acc += b[r][0] + b[r][1];
    if (acc == 0x12345u) out[0] = acc;
}

int main(int argc, char **argv) {
    const uint64_t runs = (uint64_t)(argc > 1 ? atof(argv[1]) : 90.0) * 1000000ull;
    const uint64_t cbf_bytes = (uint64_t)(argc > 2 ? atof(argv[2]) : 4300.0) << 20, dbg_bytes = (uint64_t)(argc > 3 ? atof(argv[3]) : 540.0) << 20;
    uint32_t *cbf, *dbg; uint64_t *out;
    CK(hipMalloc(&cbf, cbf_bytes)); CK(hipMalloc(&dbg, dbg_bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(cbf, 0, cbf_bytes)); CK(hipMemset(dbg, 0, dbg_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# %.0f M runs (2 bit-word loads + 2 returning atomicOr each), counters %.1f GB, bits %.2f GB\n", runs / 1e6, cbf_bytes / 1e9, dbg_bytes / 1e9);
    const uint64_t windows[] = {runs / 3000, runs / 3000 / 16, runs / 3000 * 16};
    for (int wi = 0; wi < 3; ++wi)
        for (int mode = 0; mode < 3; ++mode) {
            if (wi && mode == 0) continue;
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                const uint32_t claim = rep & 1 ? 0u : 0x80808080u;                          // (set, then nothing new to set: both are read-modify-writes)
                const int RUNS = 2;
                const uint64_t threads = (runs + RUNS - 1) / RUNS;
                CK(hipEventRecord(e0, nullptr));
                hipLaunchKernelGGL(k_probe<2>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, nullptr, cbf, cbf_bytes / 4, dbg, dbg_bytes / 4, runs, windows[wi], mode, claim, out);
                CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            CK(hipMemset(cbf, 0, cbf_bytes));
            printf("windows %8llu (%.0f KB of counters each)  mode %d (%s): %.2f ms = %.1f G runs/s\n", (unsigned long long)windows[wi], cbf_bytes / 1024.0 / windows[wi], mode,
                   mode == 0 ? "all random" : mode == 1 ? "first counter in window order" : "first counter and first bit in window order", best, runs / best / 1e6);
        }
    return 0;
}
