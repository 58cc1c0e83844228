// copy_bw.hip — what a hand-written streaming kernel reaches on this box (the yardstick for the radix passes):
//   1. 16-byte-per-lane copy of one 4 GiB array,
//   2. the three streams of a split-record pass (u32 key + u32 entry + u8 digit per element, 2^30 elements) copied
//      linearly: the same bytes as rs_onesweep_k32_v32_w8 moves, without ranking and scatter.
// build: hipcc -O3 --offload-arch=gfx950 copy_bw.hip -o copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) b[i] = a[i];
}
// one tile of 16 Ki elements per workgroup of 1024 threads, like the sort pass
__global__ __launch_bounds__(1024) void copy3(const uint32_t* __restrict__ k, const uint32_t* __restrict__ v,
                                              const uint8_t* __restrict__ w, uint32_t* __restrict__ ko,
                                              uint32_t* __restrict__ vo, uint8_t* __restrict__ wo, size_t n) {
    const size_t base = (size_t)blockIdx.x * 16384;
    uint32_t kk[16], vv[16];
    uint8_t ww[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) kk[j] = k[base + j * 1024 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < 16; ++j) vv[j] = v[base + j * 1024 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < 16; ++j) ww[j] = w[base + j * 1024 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < 16; ++j) ko[base + j * 1024 + threadIdx.x] = kk[j];
#pragma unroll
    for (int j = 0; j < 16; ++j) vo[base + j * 1024 + threadIdx.x] = vv[j];
#pragma unroll
    for (int j = 0; j < 16; ++j) wo[base + j * 1024 + threadIdx.x] = ww[j];
}

int main() {
    const size_t n = 1ull << 30;
    void *a, *b, *c, *d, *e, *f;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4));
    CK(hipMalloc(&e, n)); CK(hipMalloc(&f, n));
    CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 2, n * 4)); CK(hipMemset(c, 3, n * 4)); CK(hipMemset(d, 4, n * 4));
    CK(hipMemset(e, 5, n)); CK(hipMemset(f, 6, n));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 64}) {
        copy16<<<grid, 256>>>((const uint4*)a, (uint4*)b, n / 4);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) copy16<<<grid, 256>>>((const uint4*)a, (uint4*)b, n / 4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("copy16 4 GiB grid=%6d: %.3f ms -> %.2f TB/s (read + write)\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
    }
    copy3<<<(unsigned)(n / 16384), 1024>>>((uint32_t*)a, (uint32_t*)c, (uint8_t*)e, (uint32_t*)b, (uint32_t*)d, (uint8_t*)f, n);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r)
        copy3<<<(unsigned)(n / 16384), 1024>>>((uint32_t*)a, (uint32_t*)c, (uint8_t*)e, (uint32_t*)b, (uint32_t*)d, (uint8_t*)f, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("copy3 (u32 + u32 + u8) x 2^30, 16 Ki tiles: %.3f ms -> %.2f TB/s (18 B per element)\n", ms, 18.0 * n / ms / 1e9);
    return 0;
}
