#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// partner value for lane ^ Q through DPP / permlane swaps (gfx950)
template <int Q> __device__ __forceinline__ uint32_t xor_partner(uint32_t x) {
    if constexpr (Q == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (Q == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (Q == 4) {
        const int t = __builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);   // row_half_mirror: ^7
        return (uint32_t)__builtin_amdgcn_mov_dpp(t, 0x1B, 0xF, 0xF, true);      // quad_perm [3,2,1,0]: ^3
    } else if constexpr (Q == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, true); // row_ror:8
    else if constexpr (Q == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (threadIdx.x & 16) ? r[0] : r[1];
    } else {
        auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
}
__global__ void k(uint32_t* out) {
    const uint32_t x = threadIdx.x * 3 + 1;
    out[0 * 64 + threadIdx.x] = xor_partner<1>(x);
    out[1 * 64 + threadIdx.x] = xor_partner<2>(x);
    out[2 * 64 + threadIdx.x] = xor_partner<4>(x);
    out[3 * 64 + threadIdx.x] = xor_partner<8>(x);
    out[4 * 64 + threadIdx.x] = xor_partner<16>(x);
    out[5 * 64 + threadIdx.x] = xor_partner<32>(x);
}
int main() {
    uint32_t* d; hipMalloc(&d, 6 * 64 * 4);
    k<<<1, 64>>>(d);
    uint32_t h[6 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int q[6] = {1, 2, 4, 8, 16, 32};
    int bad = 0;
    for (int s = 0; s < 6; ++s) for (int l = 0; l < 64; ++l) if (h[s * 64 + l] != (uint32_t)((l ^ q[s]) * 3 + 1)) { if (bad < 10) printf("xor %d lane %d got %u want %u\n", q[s], l, h[s*64+l], (l ^ q[s]) * 3 + 1); ++bad; }
    printf("bad=%d\n", bad);
    return bad != 0;
}
