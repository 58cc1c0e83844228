// Do same-address LDS atomics of ONE wave instruction return in ascending lane order on gfx950?
// (decides whether a radix rank can come from ds_add_rtn instead of 8 ballots per element)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void probe(const uint32_t* __restrict__ digits, uint32_t* __restrict__ ret, int rounds) {
    __shared__ uint32_t cnt[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) (&cnt[0][0])[i] = 0;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const uint32_t d = digits[((size_t)blockIdx.x * rounds + r) * blockDim.x + threadIdx.x];
        const uint32_t old = atomicAdd(&cnt[wave][d], 1u);
        ret[((size_t)blockIdx.x * rounds + r) * blockDim.x + threadIdx.x] = old;
    }
}
int main() {
    const int blocks = 2048, threads = 256, rounds = 64;
    const size_t n = (size_t)blocks * threads * rounds;
    std::vector<uint32_t> h(n), out(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) {
        const int mode = (i / (threads * rounds)) % 4;  // per block: all-same, 2 values, 16 values, 256 values
        h[i] = mode == 0 ? 7 : mode == 1 ? (rand() & 1) * 64 : mode == 2 ? (rand() & 15) * 4 : (rand() & 255);
    }
    uint32_t *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d, o, rounds);
    hipMemcpy(out.data(), o, n * 4, hipMemcpyDeviceToHost);
    // expected: per (block, wave) running counters, lanes in ascending order inside one instruction
    size_t bad = 0;
    for (int b = 0; b < blocks; ++b) {
        uint32_t cnt[4][256] = {};
        for (int r = 0; r < rounds; ++r)
            for (int t = 0; t < threads; ++t) {
                const size_t i = ((size_t)b * rounds + r) * threads + t;
                const uint32_t want = cnt[t >> 6][h[i]]++;
                if (out[i] != want) ++bad;
            }
    }
    printf("same-address LDS atomics in lane order: %s (%zu of %zu returns differ)\n", bad ? "NO" : "yes", bad, n);
    return 0;
}
