// Issue rates of the integer VALU instructions the walkers are made of, on one MI355X (gfx950): how many clocks a SIMD needs per wave64 instruction.
// Every wavefront runs ITER x 16 independent instructions of one kind (inline asm, 16 accumulators: no dependence stalls with 8 waves per SIMD);
// rate = clocks x SIMDs / wave-instructions.  Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 16384;

#define KERNEL32(NAME, ASM)                                                                                   \
__global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t a, uint32_t b) {                           \
    uint32_t r[16];                                                                                            \
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 16u + i + a;                                             \
    uint32_t x = b | 1u;                                                                                       \
    for (int it = 0; it < ITER; ++it) {                                                                        \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(x));               \
    }                                                                                                          \
    uint32_t s = 0; for (int i = 0; i < 16; ++i) s ^= r[i];                                                    \
    if (s == 0x12345u) out[0] = s;                                                                             \
}
#define KERNEL64(NAME, ASM)                                                                                   \
__global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t a, uint32_t b) {                           \
    uint64_t r[16];                                                                                            \
    for (int i = 0; i < 16; ++i) r[i] = ((uint64_t)(threadIdx.x * 16u + i + a) << 20) | a;                     \
    uint32_t x = (b & 7u) | 1u;                                                                                \
    for (int it = 0; it < ITER; ++it) {                                                                        \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(x));               \
    }                                                                                                          \
    uint64_t s = 0; for (int i = 0; i < 16; ++i) s ^= r[i];                                                    \
    if (s == 0x12345ull) out[0] = (uint32_t)s;                                                                 \
}
KERNEL32(k_xor,      "v_xor_b32 %0, %0, %1")
KERNEL32(k_add,      "v_add_u32 %0, %0, %1")
KERNEL32(k_mul_lo,   "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi,   "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul24,    "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad24,    "v_mad_u32_u24 %0, %0, %1, %1")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL32(k_bfe,      "v_bfe_u32 %0, %0, 3, 17")
KERNEL32(k_lshl_or,  "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL32(k_and_or,   "v_and_or_b32 %0, %0, %1, %1")
KERNEL32(k_add3,     "v_add3_u32 %0, %0, %1, %1")
KERNEL32(k_min,      "v_min_u32 %0, %0, %1")
KERNEL32(k_cndmask,  "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_ffbl,     "v_ffbl_b32 %0, %0")
KERNEL32(k_cmp32,    "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_perm,     "v_perm_b32 %0, %0, %1, %1")
KERNEL32(k_sdwa,     "v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
KERNEL32(k_bitop3,   "v_bitop3_b32 %0, %0, %1, %1 bitop3:0xde")
KERNEL32(k_cnd_s,    "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32(k_cnd_nodep,"v_cndmask_b32 %0, %1, %1, vcc")
KERNEL32(k_cmp_cnd,  "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp_s_cnd,"v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32(k_cmp_xor_cnd,"v_cmp_lt_u32 vcc, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_xor4,     "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1")
KERNEL32(k_addco,    "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_addc,     "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_xor_s,    "v_xor_b32 %0, s20, %0")
KERNEL32(k_mov,      "v_mov_b32 %0, %1")
KERNEL32(k_max,      "v_max_u32 %0, %0, %1")
KERNEL64(k_lshl64,   "v_lshlrev_b64 %0, 2, %0")
KERNEL64(k_lshr64,   "v_lshrrev_b64 %0, 2, %0")
KERNEL64(k_lshr64v,  "v_lshrrev_b64 %0, %1, %0")
KERNEL64(k_cmp64,    "v_cmp_lt_u64 vcc, %0, %0")
KERNEL64(k_cmpi64,   "v_cmp_lt_i64 vcc, %0, %0")
KERNEL64(k_add64,    "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL64(k_mad64,    "v_mad_u64_u32 %0, vcc, %1, %1, %0")
KERNEL64(k_mov64,    "v_mov_b64 %0, %0")

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount; const double mhz = p.clockRate / 1000.0;
    printf("# %s, %d CUs, %.0f MHz (clockRate); grid = CUs x 8 workgroups of 256 (8 waves per SIMD), %d x 16 instructions per wave\n", p.name, cus, mhz, ITER);
    uint32_t *out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char *name; void (*k)(uint32_t *, uint32_t, uint32_t); int n; } ks[] = {
        {"v_xor_b32", k_xor, 1}, {"v_add_u32", k_add, 1}, {"v_mov_b32", k_mov, 1}, {"v_xor_b32 (sgpr operand)", k_xor_s, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1},
        {"v_mul_u32_u24", k_mul24, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_lshl_or_b32", k_lshl_or, 1},
        {"v_and_or_b32", k_and_or, 1}, {"v_add3_u32", k_add3, 1}, {"v_min_u32", k_min, 1}, {"v_max_u32", k_max, 1}, {"v_ffbl_b32", k_ffbl, 1}, {"v_perm_b32", k_perm, 1},
        {"v_xor_b32_sdwa", k_sdwa, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_cmp_lt_u32 -> vcc", k_cmp32, 1},
        {"v_cndmask_b32 (vcc)", k_cndmask, 1}, {"v_cndmask_b32 (vcc, dst not a source)", k_cnd_nodep, 1}, {"v_cndmask_b32_e64 (sgpr pair)", k_cnd_s, 1},
        {"v_cmp -> vcc; v_cndmask vcc", k_cmp_cnd, 2}, {"v_cmp -> s[20:21]; v_cndmask s[20:21]", k_cmp_s_cnd, 2}, {"v_cmp; 3 x v_xor; v_cndmask", k_cmp_xor_cnd, 5}, {"4 x v_xor (one asm)", k_xor4, 4},
        {"v_add_co_u32 -> vcc", k_addco, 1}, {"v_addc_co_u32 vcc -> vcc", k_addc, 1},
        {"v_lshlrev_b64 (const)", k_lshl64, 1}, {"v_lshrrev_b64 (const)", k_lshr64, 1}, {"v_lshrrev_b64 (vgpr)", k_lshr64v, 1}, {"v_cmp_lt_u64", k_cmp64, 1}, {"v_cmp_lt_i64", k_cmpi64, 1},
        {"v_lshl_add_u64", k_add64, 1}, {"v_mad_u64_u32", k_mad64, 1}, {"v_mov_b64", k_mov64, 1},
    };
    const int grid = cus * 8;
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.k, dim3(grid), dim3(256), 0, 0, out, 1u, 3u);
        CK(hipDeviceSynchronize());
        float ms = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.k, dim3(grid), dim3(256), 0, 0, out, 1u, 3u);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms = t < ms ? t : ms;
        }
        const double wave_insts = (double)grid * 4 * ITER * 16 * k.n, simds = cus * 4.0;
        printf("%-24s %7.3f ms  %5.2f clocks per wave64 instruction per SIMD (at 2400 MHz)\n", k.name, ms, ms * 1e-3 * 2.4e9 * simds / wave_insts);
    }
    return 0;
}
