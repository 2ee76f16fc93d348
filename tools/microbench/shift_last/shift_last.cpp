// shift_last.cpp — host side of shift_last.s: every shape's two kernels (same instructions; 8 against 16 declared VGPRs) launched in turn, every lane
// compared with the same recurrence on the CPU.  usage: ./shift_last [workgroups=131072] [iterations=256] [repeats=2] [shapes=lshr,lshl,ashr,mad,src]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static uint64_t expect(const char *shape, uint32_t id, uint32_t iters) {
    uint32_t lo = id * 0x7f4a7c15u, hi = (uint32_t)(((uint64_t)id * 0x9e3779b9u) >> 32);
    uint64_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        const uint64_t x = ((uint64_t)hi << 32) | lo;
        const uint32_t a = i + id;
        uint64_t r;
        if (!strcmp(shape, "lshr")) r = x >> (a & 63u);
        else if (!strcmp(shape, "lshl")) r = x << (a & 63u);
        else if (!strcmp(shape, "ashr")) r = (uint64_t)((int64_t)x >> (a & 63u));
        else if (!strcmp(shape, "mad")) r = (uint64_t)a * lo + x;
        else r = (((uint64_t)hi << 32) | id) >> (a & 63u);                 // "src"
        acc ^= r;
        lo += 0x9e3779b9u; hi ^= lo;
    }
    return acc;
}
int main(int argc, char **argv) {
    const uint32_t wgs = argc > 1 ? (uint32_t)atoi(argv[1]) : 131072u, iters = argc > 2 ? (uint32_t)atoi(argv[2]) : 256u;
    const int reps = argc > 3 ? atoi(argv[3]) : 2;
    const size_t n = (size_t)wgs * 256;
    hipModule_t mod; CK(hipModuleLoad(&mod, "shift_last.hsaco"));
    uint64_t *d; CK(hipMalloc(&d, n * 8));
    std::vector<uint64_t> got(n), want(n);
    std::vector<std::string> order = {"lshr", "lshl", "ashr", "mad", "src"};
    if (argc > 4) { order.clear(); for (char *t = strtok(argv[4], ","); t; t = strtok(nullptr, ",")) order.push_back(t); }   // (the first launches of a process are the ones that show it)
    for (const std::string &shape_s : order) {
        const char *shape = shape_s.c_str();
        for (size_t i = 0; i < n; ++i) want[i] = expect(shape, (uint32_t)i, iters);
        size_t total[2] = {0, 0};
        std::string detail;
        for (int r = 0; r < reps; ++r)
            for (int which = 0; which < 2; ++which) {      // <shape>16, then <shape>8: the wrong lanes show in launches that follow OTHER work on the device
                hipFunction_t f; CK(hipModuleGetFunction(&f, mod, (std::string(shape) + (which ? "8" : "16")).c_str()));
                CK(hipMemset(d, 0xEE, n * 8));
                struct { void *out; uint32_t iters; uint32_t pad; } args{d, iters, 0};
                size_t sz = sizeof args;
                void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
                CK(hipModuleLaunchKernel(f, wgs, 1, 1, 256, 1, 1, 0, nullptr, nullptr, cfg));
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(got.data(), d, n * 8, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (size_t i = 0; i < n; ++i) bad += got[i] != want[i];
                total[which] += bad;
                detail += " " + std::to_string(bad);
            }
        printf("%-5s %zu lanes x %d launches each: wrong lanes with 16 declared VGPRs %zu, with 8 (operand in the last one) %zu   [per launch, 16/8 in turn:%s]\n",
               shape, n, reps, total[0], total[1], detail.c_str());
    }
    return 0;
}
