#include <cstdio>
#include <vector>
#include <algorithm>
#include <numeric>
#include "../rna-bloom_amd/csrc/rb_internal.hpp"
namespace rb { static char e[512]; void set_error(const char* f, ...) { snprintf(e, 512, "%s", f); } }
int main(int argc, char** argv) {
  size_t n = argc > 1 ? atol(argv[1]) : 250000; int bb = argc > 2 ? atoi(argv[2]) : 32;
  std::vector<uint64_t> k(n); std::vector<uint32_t> v(n);
  uint64_t s = 1; for (size_t i = 0; i < n; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; k[i] = (s >> 20) % 5000 * 0x9E3779B97F4A7C15ull; v[i] = i; }
  uint64_t *k0, *k1; uint32_t *v0, *v1; void* tmp;
  hipMalloc(&k0, n*8); hipMalloc(&k1, n*8); hipMalloc(&v0, n*4); hipMalloc(&v1, n*4);
  size_t tb = rb::sort_pairs_temp_bytes(n); hipMalloc(&tmp, tb + 1024);
  hipMemcpy(k0, k.data(), n*8, hipMemcpyHostToDevice); hipMemcpy(v0, v.data(), n*4, hipMemcpyHostToDevice);
  try { rb::sort_pairs_u64_u32(tmp, tb + 1024, k0, k1, v0, v1, n, bb, 64, 0); } catch (...) { printf("sort threw: %s\n", rb::e); return 1; }
  hipDeviceSynchronize();
  std::vector<uint64_t> ko(n); std::vector<uint32_t> vo(n);
  hipMemcpy(ko.data(), k1, n*8, hipMemcpyDeviceToHost); hipMemcpy(vo.data(), v1, n*4, hipMemcpyDeviceToHost);
  size_t bad_pair = 0, bad_order = 0, unstable = 0;
  for (size_t i = 0; i < n; ++i) { if (vo[i] >= n || k[vo[i]] != ko[i]) bad_pair++; }
  for (size_t i = 1; i < n; ++i) { uint64_t a = ko[i-1] >> bb, b = ko[i] >> bb; if (a > b) bad_order++; if (a == b && vo[i-1] > vo[i]) unstable++; }
  printf("n=%zu begin_bit=%d temp=%zu bad_pair=%zu bad_order=%zu unstable=%zu\n", n, bb, tb, bad_pair, bad_order, unstable);
  return 0;
}
