// valu_rate.hip — issue rate of the vector instructions the generated passes lean on (gfx950).  Every kernel runs ITER x 32
// independent instructions of one kind per wave, 16 waves per CU (4 per SIMD) on every CU; cycles per instruction per SIMD =
// time x clock / (instructions per SIMD).  Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITER = 4096;

#define KERNEL32(NAME, ASM)                                                                                      \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t s) {                                     \
        uint32_t a[8];                                                                                           \
        for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i + s;                                              \
        uint32_t b = threadIdx.x | 1u, c = s + 3;                                                                \
        for (int it = 0; it < ITER; ++it) {                                                                      \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                      \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));   \
            }                                                                                                    \
        }                                                                                                        \
        uint32_t x = 0;                                                                                          \
        for (int i = 0; i < 8; ++i) x ^= a[i];                                                                   \
        if (x == 0x12345) out[threadIdx.x] = x;                                                                  \
    }
#define KERNEL64(NAME, ASM)                                                                                      \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t s) {                                     \
        uint64_t a[8];                                                                                           \
        for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i + s;                                              \
        uint64_t b = threadIdx.x | 1u;                                                                           \
        uint32_t c = (s + 3) & 7;                                                                                \
        for (int it = 0; it < ITER; ++it) {                                                                      \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                      \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));   \
            }                                                                                                    \
        }                                                                                                        \
        uint64_t x = 0;                                                                                          \
        for (int i = 0; i < 8; ++i) x ^= a[i];                                                                   \
        if (x == 0x12345) out[threadIdx.x] = (uint32_t)x;                                                        \
    }

KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL32(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, %2")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, %2, 8")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp32, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_sdwa, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL32(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 3, %1")
KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, %2, %0")
KERNEL64(k_cmp64, "v_cmp_lt_u64 vcc, %0, %1")
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %2, %2, %0")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %1")
KERNEL32(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")

int main() {
    uint32_t* d;
    CK(hipMalloc(&d, 4096));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;  // kHz -> MHz
    printf("device %s, %d CUs, clock %.0f MHz (nominal)\n", p.gcnArchName, cus, mhz);
    struct K { const char* name; void (*f)(uint32_t*, uint32_t); };
    std::vector<K> ks = {{"v_add_u32", k_add_u32}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_and_or_b32", k_and_or}, {"v_mul_u32_u24", k_mul_u24},
                         {"v_mad_u32_u24", k_mad_u24}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_dot4_u32_u8", k_dot4},
                         {"v_alignbyte_b32", k_alignbyte}, {"v_bfe_u32", k_bfe}, {"v_perm_b32", k_perm}, {"v_cndmask_b32", k_cndmask},
                         {"v_cmp_lt_u32", k_cmp32}, {"v_and_b32_sdwa", k_sdwa}, {"v_mbcnt_lo", k_mbcnt}, {"v_lshl_add_u64", k_lshl_add_u64},
                         {"v_lshlrev_b64", k_lshlrev_b64}, {"v_cmp_lt_u64", k_cmp64}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mov_b64", k_mov_b64},
                         {"v_add_co_u32", k_add_co}};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // reference: v_add_u32 is full rate (4 cycles per wave64 instruction per SIMD); everything is reported relative to it
    double ref_ms = 0;
    for (auto& k : ks) {
        const int blocks = cus * 4;  // 4 x 256 threads = 16 waves per CU = 4 per SIMD
        hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 1u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ref_ms == 0) ref_ms = ms;
        const double inst_per_simd = (double)ITER * 32 * 4;  // 4 waves per SIMD
        printf("%-18s %8.3f ms  %6.2f x v_add_u32   (%.2f cycles per instruction per SIMD at %.0f MHz)\n", k.name, ms, ms / ref_ms,
               ms * 1e-3 * mhz * 1e6 / inst_per_simd, mhz);
    }
    return 0;
}
