// alloc_explain — which property of an allocation predicts the rate of random read-modify-write atomics on it?
//
// rb::alloc_best_placed (csrc/rb_graph.hip) keeps the fastest of eight 8.5 GB allocations because the probe stages' time follows
// "where the pages land" (profiles/r03_alloc_lottery.txt: 24.8 ... 30.6 ms for the same 2 x 2^28 random XORs).  This tool looks for
// the property behind it.  For an allocation of the counting filter's size it reports
//   * the whole-array random-RMW time (the trial's own probe: 2 x 2^26 XOR pairs) and the random-READ time;
//   * the same probe confined to each 1/32 of the array (is a slow allocation slow everywhere, or in places?);
//   * the address (alignment to 2 MB / 1 GB), and — through hipMemGetAddressRange / hipPointerGetAttributes — what the runtime says;
// for allocations made with
//   A. hipMalloc, N in a row while the earlier ones are held (the lottery as the library plays it);
//   B. hipMalloc again after everything was freed (does the class come back?);
//   C. hipExtMallocWithFlags: fine-grained, uncached, contiguous;
//   D. the virtual-memory API: hipMemCreate + hipMemMap in chunks of the minimum granularity, of 32 MB, 256 MB and 1 GB, each chunk a
//      separate physical allocation mapped at consecutive addresses;
//   E. one hipMalloc of 4x the size, probing each quarter (one allocation, four placements).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/microbench/alloc_explain.hip -o /tmp/alloc_explain && /tmp/alloc_explain [GB=8.54]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}
__global__ void k_rmw(uint32_t *words, uint64_t n_words, uint32_t per_thread, uint64_t salt, unsigned long long *sink) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = t * 0x9E3779B97F4A7C15ull + salt;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_thread; ++i) {
        x = mix(x);
        acc |= atomicXor(&words[(uint64_t)(((unsigned __int128)x * n_words) >> 64)], 0x80808080u);
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}
__global__ void k_read(const uint32_t *words, uint64_t n_words, uint32_t per_thread, uint64_t salt, unsigned long long *sink) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = t * 0x9E3779B97F4A7C15ull + salt;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_thread; ++i) {
        x = mix(x);
        acc |= words[(uint64_t)(((unsigned __int128)x * n_words) >> 64)];
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

static unsigned long long *g_sink;
static hipEvent_t g_e0, g_e1;

static float time_rmw(void *p, size_t bytes, unsigned blocks = 16384, unsigned per = 16) {
    float ms = 0;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(g_e0, nullptr));
    for (int pass = 0; pass < 2; ++pass) hipLaunchKernelGGL(k_rmw, dim3(blocks), dim3(256), 0, nullptr, (uint32_t *)p, (uint64_t)(bytes / 4), per, 0x632BE59BD9B4E019ull, g_sink);
    CK(hipEventRecord(g_e1, nullptr));
    CK(hipEventSynchronize(g_e1));
    CK(hipEventElapsedTime(&ms, g_e0, g_e1));
    return ms;
}
static float time_read(void *p, size_t bytes) {
    float ms = 0;
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(g_e0, nullptr));
    hipLaunchKernelGGL(k_read, dim3(16384), dim3(256), 0, nullptr, (const uint32_t *)p, (uint64_t)(bytes / 4), 32u, 0x1234567ull, g_sink);
    CK(hipEventRecord(g_e1, nullptr));
    CK(hipEventSynchronize(g_e1));
    CK(hipEventElapsedTime(&ms, g_e0, g_e1));
    return ms;
}
static void report(const char *what, void *p, size_t bytes) {
    CK(hipMemset(p, 0, bytes));
    time_rmw(p, bytes);                                             // warm
    const float a = time_rmw(p, bytes), b = time_rmw(p, bytes), rd = time_read(p, bytes);
    // the probe confined to each 1/32 of the array (2 x 2^21 pairs each)
    const int S = 32;
    const size_t piece = bytes / S / 4096 * 4096;
    float lo = 1e9f, hi = 0, sum = 0;
    std::vector<float> v;
    for (int s = 0; s < S; ++s) {
        const float t = time_rmw((char *)p + (size_t)s * piece, piece, 2048, 4);
        v.push_back(t); lo = std::min(lo, t); hi = std::max(hi, t); sum += t;
    }
    std::sort(v.begin(), v.end());
    // the same 2 x 2^26 pairs confined to the first 1/16 ... 1/2 of the array: from which working set on do allocations differ?
    char ws[160]; int wn = 0;
    for (int div = 16; div >= 2; div /= 2) wn += snprintf(ws + wn, sizeof ws - wn, " 1/%d %.2f", div, time_rmw(p, bytes / div / 4096 * 4096));
    const uintptr_t a0 = (uintptr_t)p;
    printf("%-46s %8.3f %8.3f ms RMW | %7.3f ms reads | pieces min %.3f med %.3f max %.3f (sum %.2f) | first%s | addr %#lx mod2M %#lx mod1G %#lx\n", what, a, b, rd, lo,
           v[S / 2], hi, sum, ws, (unsigned long)a0, (unsigned long)(a0 & ((1ul << 21) - 1)), (unsigned long)(a0 & ((1ul << 30) - 1)));
    fflush(stdout);
}

struct VmAlloc { void *va = nullptr; size_t bytes = 0; std::vector<hipMemGenericAllocationHandle_t> h; };
static bool vm_alloc(VmAlloc &A, size_t bytes, size_t chunk, int dev) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (chunk < gran) chunk = gran;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t total = (bytes + chunk - 1) / chunk * chunk;
    if (hipMemAddressReserve(&A.va, total, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    A.bytes = total;
    for (size_t o = 0; o < total; o += chunk) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
        A.h.push_back(h);
        if (hipMemMap((char *)A.va + o, chunk, 0, h, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(A.va, total, &acc, 1) != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}
static void vm_free(VmAlloc &A) {
    if (A.va) { (void)hipMemUnmap(A.va, A.bytes); }
    for (auto h : A.h) (void)hipMemRelease(h);
    if (A.va) (void)hipMemAddressFree(A.va, A.bytes);
    A = VmAlloc();
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 8.542;
    const size_t bytes = (size_t)(gb * 1e9) / 4096 * 4096;
    CK(hipSetDevice(0));
    CK(hipMalloc(&g_sink, 64));
    CK(hipEventCreate(&g_e0)); CK(hipEventCreate(&g_e1));
    size_t gmin = 0, grec = 0;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    (void)hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
    (void)hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    printf("# %.3f GB per allocation; random RMW = 2 x 2^26 XOR pairs (the library's placement trial), reads = 2^27 x 4 B; VM granularity min %zu recommended %zu\n", bytes / 1e9, gmin, grec);
    printf("# A. hipMalloc, eight in a row, the earlier ones held\n");
    std::vector<void *> held;
    for (int i = 0; i < 8; ++i) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        held.push_back(p);
        char w[64]; snprintf(w, sizeof w, "A%d hipMalloc (holding %d)", i, i);
        report(w, p, bytes);
    }
    printf("# A'. the same eight again, in the same order (does an allocation keep its class?)\n");
    for (size_t i = 0; i < held.size(); ++i) { char w[64]; snprintf(w, sizeof w, "A%zu again", i); report(w, held[i], bytes); }
    for (void *p : held) CK(hipFree(p));
    held.clear();
    printf("# B. hipMalloc after everything was freed, four times alone (freed in between)\n");
    for (int i = 0; i < 4; ++i) {
        void *p = nullptr; CK(hipMalloc(&p, bytes));
        char w[64]; snprintf(w, sizeof w, "B%d hipMalloc alone", i);
        report(w, p, bytes);
        CK(hipFree(p));
    }
    printf("# C. hipExtMallocWithFlags\n");
    const struct { const char *n; unsigned f; } flags[] = {{"fine-grained", hipDeviceMallocFinegrained}, {"uncached", hipDeviceMallocUncached}, {"contiguous", 0x4}};
    for (auto &f : flags) {
        void *p = nullptr;
        if (hipExtMallocWithFlags(&p, bytes, f.f) != hipSuccess) { (void)hipGetLastError(); printf("C %s: not available\n", f.n); continue; }
        char w[64]; snprintf(w, sizeof w, "C hipExtMallocWithFlags %s", f.n);
        report(w, p, bytes);
        CK(hipFree(p));
    }
    printf("# D. hipMemCreate + hipMemMap, physical chunks of a given size mapped back to back\n");
    // (the minimum granularity is 4 KB here: 2 M physical allocations per array — not tried; 2 MB is the smallest chunk)
    const size_t chunks[] = {(size_t)2 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)1 << 30};
    for (int i = 0; i < 4; ++i) {
        for (int rep = 0; rep < 2; ++rep) {
            VmAlloc A;
            const size_t ch = chunks[i];
            if (!vm_alloc(A, bytes, ch, 0)) { printf("D chunk %zu: VM API failed\n", ch); vm_free(A); break; }
            char w[64]; snprintf(w, sizeof w, "D%d VM chunks of %zu KB (%zu)", rep, ch >> 10, A.h.size());
            report(w, A.va, bytes);
            vm_free(A);
        }
    }
    {
        VmAlloc A;                                                  // one physical allocation for the whole array
        if (vm_alloc(A, bytes, bytes, 0)) { report("D one chunk = the whole array", A.va, bytes); }
        else printf("D one chunk: VM API failed\n");
        vm_free(A);
    }
    printf("# E. one hipMalloc of 4x the size: each quarter probed on its own\n");
    {
        void *p = nullptr;
        if (hipMalloc(&p, 4 * bytes) == hipSuccess) {
            for (int q = 0; q < 4; ++q) { char w[64]; snprintf(w, sizeof w, "E quarter %d of one 4x allocation", q); report(w, (char *)p + (size_t)q * bytes, bytes); }
            CK(hipFree(p));
        } else { (void)hipGetLastError(); printf("E: no room\n"); }
    }
    printf("# F. sizes: is it the SIZE (channel / page interleave) rather than the place? hipMalloc alone, RMW probe scaled to the size\n");
    for (double g : {1.0, 2.0, 4.0, 8.0, 8.542, 16.0}) {
        const size_t b = (size_t)(g * 1e9) / 4096 * 4096;
        void *p = nullptr; CK(hipMalloc(&p, b));
        char w[64]; snprintf(w, sizeof w, "F hipMalloc %.3f GB", g);
        report(w, p, b);
        CK(hipFree(p));
    }
    return 0;
}
