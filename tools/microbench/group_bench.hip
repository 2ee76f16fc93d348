// group_bench.hip — the grouping stage of the insert pipeline (csrc/rb_group.hip) on synthetic records, next to
// rocPRIM's onesweep on the same records; validates the grouped output on the host for small n.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench/group_bench.hip rna-bloom_amd/build/rb_sort.o -o tools/microbench/group_bench
//   ./group_bench <n_records> [mean multiplicity = 5] [check = 0/1]
#include "../../rna-bloom_amd/csrc/rb_group.hip"

#include <stdarg.h>
#include <unordered_map>
namespace rb {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
}
using namespace rb;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
// heavy > 0: 5 % of the records belong to `heavy` hot keys (oversized buckets -> k_group_big)
__host__ __device__ inline uint64_t key_of(uint32_t occ, uint64_t distinct, uint32_t heavy = 0) {
    const uint64_t h = mix64(occ);
    if (heavy && h % 20 == 0) return mix64(0xABCDEF00ull + (h >> 8) % heavy);
    return mix64(h % distinct);
}
__global__ void k_fill(uint64_t *k, uint32_t *v, size_t n, uint64_t distinct, uint32_t heavy) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    k[i] = key_of((uint32_t)i, distinct, heavy); v[i] = (uint32_t)i;
}
int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 20;
    const double mult = argc > 2 ? atof(argv[2]) : 5.0;
    const int check = argc > 3 ? atoi(argv[3]) : 0;
    const uint32_t heavy = argc > 4 ? (uint32_t)atoi(argv[4]) : 0u;
    const int gbits = getenv("GB") ? atoi(getenv("GB")) : 32;
    const uint64_t distinct = (uint64_t)std::max(1.0, n / mult);
    uint64_t *k0, *k1, *uniq; uint32_t *v0, *v1, *vout, *counts, *starts, *nruns; uint8_t *tz; void *temp;
    CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4)); CK(hipMalloc(&vout, n * 4));
    CK(hipMalloc(&uniq, n * 8)); CK(hipMalloc(&counts, n * 4 + 4)); CK(hipMalloc(&starts, n * 4 + 4)); CK(hipMalloc(&nruns, 64)); CK(hipMalloc(&tz, n + 16));
    const size_t tb = group_temp_bytes(n, gbits);
    CK(hipMalloc(&temp, tb));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t seed = 77, ord0 = 5; const uint32_t pos_bits = 7;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, k0, v0, n, distinct, heavy);
        CK(hipEventRecord(e0, st));
        group_records_device(k0, v0, k1, v1, n, gbits, seed, ord0, pos_bits, temp, tb, vout, tz, uniq, counts, starts, nruns, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = std::min(best, ms);
    }
    uint32_t D = 0; CK(hipMemcpy(&D, nruns, 4, hipMemcpyDeviceToHost));
    const GroupPlan P = group_plan(n, gbits);
    { uint32_t nb_ = 0; CK(hipMemcpy(&nb_, (char *)temp + P.off_ticket + 8, 4, hipMemcpyDeviceToHost)); printf("big buckets: %u  ", nb_); }
    printf("n=%zu distinct<=%llu runs=%u  plan: T=%u+%u local=%u+%u tpb=%u xcd=%d  group: %.3f ms  (%.1f B/rec model -> %.2f TB/s)\n", n,
           (unsigned long long)distinct, D, P.t_hi, P.t_lo, P.l_hi, P.l_lo, P.tpb, P.xcd_map, best, 85.0, 85.0 * n / best / 1e9);
    {   // rocPRIM onesweep + what follows it today (for comparison: sort only)
        size_t sb = sort_pairs_temp_bytes(n); void *stmp; CK(hipMalloc(&stmp, sb));
        float bs = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, k0, v0, n, distinct, heavy);
            CK(hipEventRecord(e0, st));
            sort_pairs_u64_u32(stmp, sb, k0, k1, v0, v1, n, 32, 64, st);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) bs = std::min(bs, ms);
        }
        printf("rocPRIM onesweep (32 bits, sort only): %.3f ms\n", bs);
    }
    if (check) {
        std::vector<uint32_t> hv(n), hc(D), hs(D); std::vector<uint64_t> hu(D); std::vector<uint8_t> ht(n);
        CK(hipMemcpy(hv.data(), vout, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), counts, (size_t)D * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), starts, (size_t)D * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hu.data(), uniq, (size_t)D * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ht.data(), tz, n, hipMemcpyDeviceToHost));
        size_t bad = 0, pos = 0, split = 0;
        std::unordered_map<uint64_t, uint32_t> last; last.reserve(distinct * 2);
        std::vector<uint8_t> seen(n, 0);
        std::vector<uint32_t> order(D);
        for (uint32_t r = 0; r < D; ++r) order[r] = r;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hs[a] < hs[b]; });   // runs of oversized buckets are appended out of order
        for (uint32_t ri = 0; ri < D; ++ri) {
            const uint32_t r = order[ri];
            if (hs[r] != pos || hc[r] == 0) { if (bad < 10) printf("run %u: start %u count %u expected start %zu\n", r, hs[r], hc[r], pos); ++bad; }
            uint32_t prev = 0; bool first = true;
            for (uint32_t j = hs[r]; j < hs[r] + hc[r] && j < n; ++j) {
                const uint32_t occ = hv[j];
                if (occ >= n || seen[occ]) { ++bad; continue; }
                seen[occ] = 1;
                if (key_of(occ, distinct, heavy) != hu[r]) { if (bad < 10) printf("run %u rec %u: key mismatch\n", r, j); ++bad; }
                if (!first && occ <= prev) { if (bad < 10) printf("run %u rec %u: order\n", r, j); ++bad; }
                const uint32_t rr = rng31(seed, ord0 + (occ >> pos_bits), occ & ((1u << pos_bits) - 1u)) | 0x8000u;
                if (ht[j] != (uint8_t)(__builtin_ffs((int)rr) - 1)) { if (bad < 10) printf("rec %u: strength\n", j); ++bad; }
                prev = occ; first = false;
            }
            auto it = last.find(hu[r]);
            if (it != last.end()) { ++split; if (hv[hs[r]] <= it->second) { if (bad < 10) printf("run %u: split run out of order\n", r); ++bad; } it->second = prev; }
            else last.emplace(hu[r], prev);
            pos += hc[r];
        }
        if (pos != n) { printf("runs cover %zu of %zu records\n", pos, n); ++bad; }
        printf("check: %s (%zu problems), distinct keys %zu, split runs %zu (%.3f %%)\n", bad ? "FAILED" : "ok", bad, last.size(), split, 100.0 * split / std::max<size_t>(1, last.size()));
        return bad ? 1 : 0;
    }
    return 0;
}
