// What does a wavefront pay for fetching 128-byte buckets the way k_filter_reads_pipe does — every lane that needs one issues 8 x global_load_dwordx4 under a
// sparse exec mask — against a cooperative fetch (8 lanes x one dwordx4 per bucket)?  Per iteration a wavefront has A lanes that want a random bucket.
//   own:  if (want) for q in 0..7: R[q] = bucket[q]         (8 vector-memory instructions per iteration, A lanes active in each)
//   coop: the A wanted buckets are handed to groups of 8 lanes, one dwordx4 per lane, ceil(A/8) instructions per iteration
// Build: hipcc -O3 --offload-arch=gfx950 bucket_fetch.hip -o bucket_fetch ; run: ./bucket_fetch [log2 table bytes = 31]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 2048;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; return x; }

template <int VALU_PAD>
__global__ void __launch_bounds__(64) k_own(const ulonglong2 *__restrict__ tab, uint32_t bmask, uint32_t thresh, unsigned long long *out) {
    const uint32_t lane = threadIdx.x, gid = blockIdx.x * 64u + lane;
    unsigned long long acc = 0; uint32_t pad = gid;
    for (int it = 0; it < ITER; ++it) {
        const uint32_t r = mix(gid * 2654435761u + (uint32_t)it);
        const bool want = (r & 0xFFFFu) < thresh;
        if (want) {
            const ulonglong2 *bp = tab + ((size_t)((r >> 4) & bmask) << 3);
            ulonglong2 R[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) R[q] = bp[q];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc ^= R[q].x + R[q].y;
        }
#pragma unroll
        for (int v = 0; v < VALU_PAD; ++v) pad = pad * 0x27D4EB2Fu + 1u;      // the walker's other work (per-step VALU), to see whether the fetch hides behind it
    }
    if (acc == 0x1234567ull + pad) out[0] = acc;
}
template <int VALU_PAD>
__global__ void __launch_bounds__(64) k_coop(const ulonglong2 *__restrict__ tab, uint32_t bmask, uint32_t thresh, unsigned long long *out) {
    __shared__ uint32_t s_list[64];
    const uint32_t lane = threadIdx.x, gid = blockIdx.x * 64u + lane;
    unsigned long long acc = 0; uint32_t pad = gid;
    for (int it = 0; it < ITER; ++it) {
        const uint32_t r = mix(gid * 2654435761u + (uint32_t)it);
        const bool want = (r & 0xFFFFu) < thresh;
        const unsigned long long m = __ballot(want);
        if (want) s_list[__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (r >> 4) & bmask;
        const uint32_t n = (uint32_t)__popcll(m);
        for (uint32_t p = 0; p * 8u < n; ++p) {                       // wave-uniform
            const uint32_t e = p * 8u + (lane >> 3);
            if (e < n) {
                const ulonglong2 v = tab[((size_t)s_list[e] << 3) + (lane & 7u)];
                acc ^= v.x + v.y;
            }
        }
#pragma unroll
        for (int v = 0; v < VALU_PAD; ++v) pad = pad * 0x27D4EB2Fu + 1u;
    }
    if (acc == 0x1234567ull + pad) out[0] = acc;
}


// groups of FOUR lanes x two dwordx4 per bucket, as k_filter_reads_coop does; UNCOND: lanes without a task load the table's first line (one cached line for
// everybody) so that the loads sit in straight-line code and the compiler can count them (s_waitcnt vmcnt(N) instead of vmcnt(0))
template <bool UNCOND>
__global__ void __launch_bounds__(64) k_coop4(const ulonglong2 *__restrict__ tab, uint32_t bmask, uint32_t thresh, unsigned long long *out) {
    __shared__ uint32_t s_list[64];
    const uint32_t lane = threadIdx.x, gid = blockIdx.x * 64u + lane, grp = lane >> 2, sub = lane & 3u;
    unsigned long long acc = 0;
    for (int it = 0; it < ITER; ++it) {
        const uint32_t r = mix(gid * 2654435761u + (uint32_t)it);
        const bool want = (r & 0xFFFFu) < thresh;
        const unsigned long long m = __ballot(want);
        if (want) s_list[__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (r >> 4) & bmask;
        const uint32_t n = (uint32_t)__popcll(m);
        for (uint32_t p = 0; p * 16u < n || (UNCOND && p == 0); ++p) {                       // wave-uniform
            const uint32_t e = p * 16u + grp;
            if (UNCOND) {
                const uint32_t ent = s_list[e & 63u];
                const ulonglong2 *bp = tab + (e < n ? ((size_t)ent << 3) : (size_t)0) + sub;
                const ulonglong2 v0 = bp[0], v1 = bp[4];
                acc ^= v0.x + v0.y + v1.x + v1.y;
            } else if (e < n) {
                const ulonglong2 *bp = tab + ((size_t)s_list[e] << 3) + sub;
                const ulonglong2 v0 = bp[0], v1 = bp[4];
                acc ^= v0.x + v0.y + v1.x + v1.y;
            }
        }
    }
    if (acc == 0x1234567ull) out[0] = acc;
}

// coop4 with BUFFER loads issued by every lane in every iteration: a lane without a task uses an offset outside the resource, which the hardware answers
// with zeros without touching memory — straight-line code the compiler can count (vmcnt(N)) at no cost in cache requests
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) k_coop4_buf(const ulonglong2 *__restrict__ tab, uint32_t bmask, uint32_t thresh, unsigned long long *out) {
    __shared__ uint32_t s_list[64];
    const uint32_t lane = threadIdx.x, gid = blockIdx.x * 64u + lane, grp = lane >> 2, sub = lane & 3u;
    const uint32_t bytes = (bmask + 1u) << 7;            // (tables below 4 GB)
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, (short)0, (int)bytes, 0x00020000);
    unsigned long long acc = 0;
    for (int it = 0; it < ITER; ++it) {
        const uint32_t r = mix(gid * 2654435761u + (uint32_t)it);
        const bool want = (r & 0xFFFFu) < thresh;
        const unsigned long long m = __ballot(want);
        if (want) s_list[__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (r >> 4) & bmask;
        const uint32_t n = (uint32_t)__popcll(m);
        for (uint32_t p = 0; p * 16u < n || p == 0; ++p) {                       // wave-uniform
            const uint32_t e = p * 16u + grp;
            const uint32_t ent = s_list[e & 63u];
            const uint32_t off = e < n ? (ent << 7) + sub * 16u : 0xFFFFFFFFu;
            const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0), v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off + 64, 0, 0);
            acc ^= (unsigned long long)(v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w);
        }
    }
    if (acc == 0x1234567ull) out[0] = acc;
}

template <class K> static int run(const char *name, K k, const ulonglong2 *tab, uint32_t bmask, unsigned long long *out, int cus, int wps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = cus * 4 * wps;
    const uint32_t ths[] = {1024, 4096, 9830, 16384, 32768, 65535};       // 1/64, 1/16, 0.15 (the walker's), 1/4, 1/2, all
    for (uint32_t th : ths) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, tab, bmask, th, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); best = t < best ? t : best;
        }
        const double wave_iters = (double)grid * ITER, clk = best * 1e-3 * 2.4e9;
        const double fetches = wave_iters * 64.0 * (th / 65536.0);
        printf("%-22s want %5.3f of the lanes: %8.3f ms  %7.1f clocks per wave-iteration per CU  %6.2f G buckets/s  %6.2f TB/s\n", name, th / 65536.0, best,
               clk * cus / wave_iters, fetches / best / 1e6, fetches * 128 / best / 1e9);
    }
    return 0;
}
int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 31;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    ulonglong2 *tab; const size_t bytes = (size_t)1 << lg;
    CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 1, bytes));
    unsigned long long *out; CK(hipMalloc(&out, 64));
    const uint32_t bmask = (uint32_t)((bytes >> 7) - 1);
    printf("# table 2^%d bytes, %d CUs, 4 waves per SIMD, %d iterations per wave; 'clocks per wave-iteration per CU' = time x 2.4 GHz x CUs / (waves x iterations)\n", lg, cus, ITER);
    if (run("own 8 x dwordx4", k_own<0>, tab, bmask, out, cus, 4)) return 1;
    if (run("coop 1 x dwordx4", k_coop<0>, tab, bmask, out, cus, 4)) return 1;
    if (run("coop4 2 x dwordx4", k_coop4<false>, tab, bmask, out, cus, 4)) return 1;
    if (run("coop4 unconditional", k_coop4<true>, tab, bmask, out, cus, 4)) return 1;
    if (lg < 32 && run("coop4 buffer, OOB idle", k_coop4_buf, tab, bmask, out, cus, 4)) return 1;
    return 0;
}
