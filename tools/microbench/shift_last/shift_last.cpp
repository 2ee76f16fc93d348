// shift_last.cpp — host side of shift_last.s: launches both kernels (same instructions; 8 against 16 declared VGPRs) and compares every lane's result with
// the same recurrence on the CPU.  usage: ./shift_last [workgroups=65536] [iterations=256] [repeats=4]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static uint64_t expect(uint32_t id, uint32_t iters) {
    uint32_t lo = id * 0x7f4a7c15u, hi = (uint32_t)(((uint64_t)id * 0x9e3779b9u) >> 32);
    uint64_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        const uint64_t x = ((uint64_t)hi << 32) | lo;
        acc ^= x >> ((i + id) & 63u);
        lo += 0x9e3779b9u; hi ^= lo;
    }
    return acc;
}
int main(int argc, char **argv) {
    const uint32_t wgs = argc > 1 ? (uint32_t)atoi(argv[1]) : 65536u, iters = argc > 2 ? (uint32_t)atoi(argv[2]) : 256u;
    const int reps = argc > 3 ? atoi(argv[3]) : 4;
    const size_t n = (size_t)wgs * 256;
    hipModule_t mod; CK(hipModuleLoad(&mod, "shift_last.hsaco"));
    uint64_t *d; CK(hipMalloc(&d, n * 8));
    std::vector<uint64_t> got(n), want(n);
    for (size_t i = 0; i < n; ++i) want[i] = expect((uint32_t)i, iters);
    hipFunction_t f16, f8; CK(hipModuleGetFunction(&f16, mod, "shift_last16")); CK(hipModuleGetFunction(&f8, mod, "shift_last8"));
    size_t total_bad[2] = {0, 0};
    // the two kernels take turns: the wrong lanes show up in launches of shift_last8 that follow OTHER work on the device (the first one after
    // shift_last16 or after a fill; back-to-back launches of shift_last8 alone come out exact)
    for (int r = 0; r < reps; ++r)
        for (int which = 0; which < 2; ++which) {
            const char *name = which ? "shift_last8" : "shift_last16";
            CK(hipMemset(d, 0xEE, n * 8));
            struct { void *out; uint32_t iters; uint32_t pad; } args{d, iters, 0};
            size_t sz = sizeof args;
            void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            CK(hipModuleLaunchKernel(which ? f8 : f16, wgs, 1, 1, 256, 1, 1, 0, nullptr, nullptr, cfg));
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), d, n * 8, hipMemcpyDeviceToHost));
            size_t bad = 0, first = n;
            for (size_t i = 0; i < n; ++i) if (got[i] != want[i]) { if (!bad) first = i; ++bad; }
            total_bad[which] += bad;
            printf("%-13s run %d: %zu lanes, %zu wrong", name, r, n, bad);
            if (bad) printf(" (first: lane %zu = wavefront %zu lane %zu: got %016llx want %016llx)", first, first / 64, first % 64, (unsigned long long)got[first], (unsigned long long)want[first]);
            printf("\n");
        }
    printf("total wrong lanes: shift_last16 %zu, shift_last8 %zu\n", total_bad[0], total_bad[1]);
    return 0;
}
