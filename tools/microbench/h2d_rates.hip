// h2d_rates — host-to-device copy rate of 1 GiB by where the host memory came from (the host-resident leg of bench.py):
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/h2d_rates.hip -o /tmp/h2d_rates && /tmp/h2d_rates
// hipHostMalloc default / non-coherent / NUMA-user, malloc + hipHostRegister, plain malloc; one stream, and two streams side by side.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = (size_t)1 << 30;
    void *d = nullptr, *d2 = nullptr;
    CK(hipMalloc(&d, N)); CK(hipMalloc(&d2, N));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    struct V { const char *name; unsigned flags; int kind; } vs[] = {
        {"hipHostMalloc default", hipHostMallocDefault, 0}, {"hipHostMalloc non-coherent", hipHostMallocNonCoherent, 0},
        {"hipHostMalloc coherent", hipHostMallocCoherent, 0}, {"hipHostMalloc numa-user", hipHostMallocNumaUser, 0},
        {"hipHostMalloc portable|mapped", hipHostMallocPortable | hipHostMallocMapped, 0},
        {"malloc + hipHostRegister", 0, 1}, {"malloc (pageable)", 0, 2}};
    for (auto &v : vs) {
        void *h = nullptr;
        if (v.kind == 0) { if (hipHostMalloc(&h, N, v.flags) != hipSuccess) { printf("%-32s alloc failed\n", v.name); (void)hipGetLastError(); continue; } }
        else { h = aligned_alloc(4096, N); }
        memset(h, 1, N);
        if (v.kind == 1) CK(hipHostRegister(h, N, hipHostRegisterDefault));
        double best = 1e9, best2 = 1e9, bestd = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            double t0 = now();
            CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
            double t1 = now();
            CK(hipMemcpyAsync(d, h, N / 2, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(d2, (char *)h + N / 2, N / 2, hipMemcpyHostToDevice, s2));
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
            double t2 = now();
            CK(hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
            double t3 = now();
            if (t1 - t0 < best) best = t1 - t0;
            if (t2 - t1 < best2) best2 = t2 - t1;
            if (t3 - t2 < bestd) bestd = t3 - t2;
        }
        printf("%-32s H2D %6.1f GB/s   two streams %6.1f GB/s   D2H %6.1f GB/s\n", v.name, N / best / 1e9, N / best2 / 1e9, N / bestd / 1e9);
        if (v.kind == 1) CK(hipHostUnregister(h));
        if (v.kind == 0) CK(hipHostFree(h)); else free(h);
    }
    return 0;
}
