// mall_scatter.hip — two yardsticks for the "cut a radix pass" question:
//   1. what a streaming copy reaches when its working set stays inside the 256 MiB Infinity Cache (would sorting a few
//      buckets at a time, all passes back to back, run above the HBM rate?);
//   2. what the write pattern of a radix pass costs as a function of the digit width alone: every 16 Ki-element tile is read
//      linearly and written as D runs of 16384 / D elements each to D far-apart places — the scatter of a pass over uniform
//      keys with perfect knowledge of the destinations (no ranking, no look-back).  A pass with 2^b-way digits cannot be
//      faster than this; 3 passes at 11 bits have to beat 4 passes at 8 bits.
// build: hipcc -O3 --offload-arch=gfx950 mall_scatter.hip -o mall_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// `rounds` sweeps over the same n16 vectors inside one launch (persistent grid): after the first sweep a working set below
// the cache size is served from the Infinity Cache
__global__ __launch_bounds__(256) void copy16_rounds(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n16, int rounds) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (int r = 0; r < rounds; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) b[i] = a[i];
}
__global__ __launch_bounds__(256) void read16_rounds(const uint4* __restrict__ a, uint32_t* __restrict__ sink, size_t n16, int rounds) {
    const size_t stride = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
            const uint4 v = a[i];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    if (acc == 0x12345678u) sink[0] = acc;
}

// tile t (16 Ki elements, two u32 streams) -> D runs; run d of tile t lands at seg_base + d * (seg_len / D) + (t % tiles_per_seg) * RUN
// seg_len = elements per "bucket" (the in-bucket passes scatter inside a bucket); seg_len = n: whole-array scatter
// SHIFT (round 6, VERDICT r5 item 4 i): every run starts at a pseudo-random offset of 0..15 elements from its 64-byte-aligned place
// — what the runs of a real pass look like (their starts are prefix sums of digit counts) — so a wave's 256-byte store covers
// five partial lines instead of four whole ones.  The difference to SHIFT = false is what a destination-aligned write-out
// could buy at most.
template <int D, bool GROUPED, bool SHIFT = false>
__global__ __launch_bounds__(1024) void scatter_runs(const uint32_t* __restrict__ k, const uint32_t* __restrict__ v,
                                                     uint32_t* __restrict__ ko, uint32_t* __restrict__ vo, size_t seg_len) {
    constexpr int RUN = 16384 / D;
    // XCD-aware tile order of the production passes (RsCfg::GROUP = 8): workgroup b runs on XCD b % 8, which owns the tile
    // groups x, x + 8, ... of 8 consecutive tiles each — neighbouring runs meet in one L2
    const size_t slot = blockIdx.x / 8, x = blockIdx.x % 8;
    const size_t tile = GROUPED ? ((slot / 8) * 8 + x) * 8 + slot % 8 : (size_t)blockIdx.x;
    const size_t base = tile * 16384;
    const size_t tiles_per_seg = seg_len / 16384;
    const size_t seg = tile / tiles_per_seg, tin = tile % tiles_per_seg;
    uint32_t kk[16], vv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) kk[j] = k[base + j * 1024 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < 16; ++j) vv[j] = v[base + j * 1024 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t i = j * 1024 + threadIdx.x;
        const uint32_t d = i / RUN, r = i % RUN;
        const uint32_t sh = SHIFT ? (uint32_t)((tile * 2654435761ull + d * 40503u) >> 7) & 15u : 0u;
        const size_t dst = seg * seg_len + (size_t)d * (seg_len / D) + tin * RUN + r + sh;
        ko[dst] = kk[j];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t i = j * 1024 + threadIdx.x;
        const uint32_t d = i / RUN, r = i % RUN;
        const uint32_t sh = SHIFT ? (uint32_t)((tile * 2654435761ull + d * 40503u) >> 7) & 15u : 0u;
        const size_t dst = seg * seg_len + (size_t)d * (seg_len / D) + tin * RUN + r + sh;
        vo[dst] = vv[j];
    }
}

// ---- round 6: the FAITHFUL form of the alignment question.  Runs of a real pass ABUT: run (tile t, digit d) starts where run
// (t - 1, d) ended, so every line is eventually written in full (by two tiles at a run boundary) — the shifted form above leaves
// gaps and overlaps instead, which no pass produces.  Here run (t, d) has 49..64 elements (64 - hash & 15), its start is the
// prefix sum over the earlier tiles (table `starts`, [tiles + 1][256]), a wave writes one run per trip (lanes behind the run's end
// idle).  ALIGNED: the same runs, the same number of elements, but every run starts on its own 64-byte boundary (64 t) — what a
// destination-aligned write-out would produce if it cost nothing.
__global__ void ragged_lens_kernel(uint32_t* __restrict__ starts, uint32_t tiles) {   // one thread per digit: prefix over the tiles
    const uint32_t d = threadIdx.x;
    uint32_t run = 0;
    for (uint32_t t = 0; t <= tiles; ++t) {
        starts[(size_t)t * 256 + d] = run;
        run += 64u - ((uint32_t)(((uint64_t)t * 2654435761ull + d * 40503u) >> 7) & 15u);
    }
}
template <bool ALIGNED>
__global__ __launch_bounds__(1024) void scatter_ragged(const uint32_t* __restrict__ k, const uint32_t* __restrict__ v,
                                                       uint32_t* __restrict__ ko, uint32_t* __restrict__ vo,
                                                       const uint32_t* __restrict__ starts, size_t seg_len) {
    const size_t slot = blockIdx.x / 8, x = blockIdx.x % 8;
    const size_t tile = ((slot / 8) * 8 + x) * 8 + slot % 8;
    const size_t base = tile * 16384;
    const size_t tiles_per_seg = seg_len / 16384;
    const size_t seg = tile / tiles_per_seg, tin = tile % tiles_per_seg;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t kk[16], vv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) kk[j] = k[base + j * 1024 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < 16; ++j) vv[j] = v[base + j * 1024 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t d = j * 16 + wave;
        const uint32_t s0 = starts[tin * 256 + d], len = starts[(tin + 1) * 256 + d] - s0;
        const size_t dst = seg * seg_len + (size_t)d * (seg_len / 256) + (ALIGNED ? tin * 64 : (size_t)s0) + lane;
        if (lane < len) ko[dst] = kk[j];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t d = j * 16 + wave;
        const uint32_t s0 = starts[tin * 256 + d], len = starts[(tin + 1) * 256 + d] - s0;
        const size_t dst = seg * seg_len + (size_t)d * (seg_len / 256) + (ALIGNED ? tin * 64 : (size_t)s0) + lane;
        if (lane < len) vo[dst] = vv[j];
    }
}
template <bool ALIGNED>
float time_ragged(uint32_t* a, uint32_t* c, uint32_t* b, uint32_t* d, const uint32_t* starts, size_t n, size_t seg_len, hipEvent_t e0, hipEvent_t e1) {
    scatter_ragged<ALIGNED><<<(unsigned)(n / 16384), 1024>>>(a, c, b, d, starts, seg_len);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) scatter_ragged<ALIGNED><<<(unsigned)(n / 16384), 1024>>>(a, c, b, d, starts, seg_len);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

template <int D, bool GROUPED = true, bool SHIFT = false>
float time_scatter(uint32_t* a, uint32_t* c, uint32_t* b, uint32_t* d, size_t n, size_t seg_len, hipEvent_t e0, hipEvent_t e1) {
    scatter_runs<D, GROUPED, SHIFT><<<(unsigned)(n / 16384), 1024>>>(a, c, b, d, seg_len);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) scatter_runs<D, GROUPED, SHIFT><<<(unsigned)(n / 16384), 1024>>>(a, c, b, d, seg_len);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

int main() {
    const size_t n = 1ull << 30;
    void *a, *b, *c, *d;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4 + 256)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4 + 256));
    CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 2, n * 4)); CK(hipMemset(c, 3, n * 4)); CK(hipMemset(d, 4, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ---- 1. working-set sweep
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 512, 4096}) {
        const size_t bytes = mb << 20, n16 = bytes / 16;
        const int rounds = (int)((8ull << 30) / bytes) < 2 ? 2 : (int)((8ull << 30) / bytes);
        copy16_rounds<<<2048, 256>>>((const uint4*)a, (uint4*)b, n16, 2);
        CK(hipEventRecord(e0));
        copy16_rounds<<<2048, 256>>>((const uint4*)a, (uint4*)b, n16, rounds);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipEventRecord(e0));
        read16_rounds<<<2048, 256>>>((const uint4*)a, (uint32_t*)d, n16, rounds);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
        printf("working set %5zu MiB in + %5zu MiB out, %4d sweeps: copy %.2f TB/s (read + write)   read-only %.2f TB/s\n", mb, mb, rounds,
               2.0 * bytes * rounds / ms / 1e9, 1.0 * bytes * rounds / ms2 / 1e9);
    }
    // ---- 2. scatter cost by digit width (8 B per element in, 8 B out)
    for (size_t seg_len : {n, (size_t)1 << 22}) {   // whole array / inside buckets of 4 Mi records (32 MiB)
        printf("scatter of 2^30 (u32, u32) records, 16 Ki tiles, %s:\n", seg_len == n ? "over the whole array" : "inside buckets of 2^22 records");
        const float t256 = time_scatter<256>((uint32_t*)a, (uint32_t*)c, (uint32_t*)b, (uint32_t*)d, n, seg_len, e0, e1);
        const float t512 = time_scatter<512>((uint32_t*)a, (uint32_t*)c, (uint32_t*)b, (uint32_t*)d, n, seg_len, e0, e1);
        const float t1k = time_scatter<1024>((uint32_t*)a, (uint32_t*)c, (uint32_t*)b, (uint32_t*)d, n, seg_len, e0, e1);
        const float t2k = time_scatter<2048>((uint32_t*)a, (uint32_t*)c, (uint32_t*)b, (uint32_t*)d, n, seg_len, e0, e1);
        const float t16 = time_scatter<16>((uint32_t*)a, (uint32_t*)c, (uint32_t*)b, (uint32_t*)d, n, seg_len, e0, e1);
        const float t256p = time_scatter<256, false>((uint32_t*)a, (uint32_t*)c, (uint32_t*)b, (uint32_t*)d, n, seg_len, e0, e1);
        const float t2kp = time_scatter<2048, false>((uint32_t*)a, (uint32_t*)c, (uint32_t*)b, (uint32_t*)d, n, seg_len, e0, e1);
        printf("  (tiles in blockIdx order instead of the XCD-aware order: D = 256 %.3f ms, D = 2048 %.3f ms)\n", t256p, t2kp);
        const float t256s = time_scatter<256, true, true>((uint32_t*)a, (uint32_t*)c, (uint32_t*)b, (uint32_t*)d, n, seg_len, e0, e1);
        const float t256a = time_scatter<256>((uint32_t*)a, (uint32_t*)c, (uint32_t*)b, (uint32_t*)d, n, seg_len, e0, e1);
        printf("  D = 256, run starts 64-byte aligned %.3f ms / shifted by 0..15 elements (as in a real pass) %.3f ms: alignment is worth %.1f %%\n",
               t256a, t256s, 100.0 * (t256s - t256a) / t256s);
        {
            const uint32_t tps = (uint32_t)(seg_len / 16384);
            uint32_t* starts;
            CK(hipMalloc(&starts, ((size_t)tps + 1) * 256 * 4));
            ragged_lens_kernel<<<1, 256>>>(starts, tps);
            CK(hipDeviceSynchronize());
            const float ta = time_ragged<true>((uint32_t*)a, (uint32_t*)c, (uint32_t*)b, (uint32_t*)d, starts, n, seg_len, e0, e1);
            const float tu = time_ragged<false>((uint32_t*)a, (uint32_t*)c, (uint32_t*)b, (uint32_t*)d, starts, n, seg_len, e0, e1);
            printf("  D = 256, ABUTTING runs of 49..64 elements (88 %% of the elements written): starts as the prefix sums give them %.3f ms / "
                   "each run on its own 64-byte boundary %.3f ms: a destination-aligned write-out is worth at most %.1f %%\n",
                   tu, ta, 100.0 * (tu - ta) / tu);
            CK(hipFree(starts));
        }
        printf("  D =   16 (4 KiB runs): %.3f ms  %.2f TB/s\n", t16, 16.0 * n / t16 / 1e9);
        printf("  D =  256 (256 B runs): %.3f ms  %.2f TB/s\n", t256, 16.0 * n / t256 / 1e9);
        printf("  D =  512 (128 B runs): %.3f ms  %.2f TB/s\n", t512, 16.0 * n / t512 / 1e9);
        printf("  D = 1024 ( 64 B runs): %.3f ms  %.2f TB/s\n", t1k, 16.0 * n / t1k / 1e9);
        printf("  D = 2048 ( 32 B runs): %.3f ms  %.2f TB/s\n", t2k, 16.0 * n / t2k / 1e9);
        printf("  40-bit key: 5 passes x 8 bits = %.2f ms, 4 x 10 bits = %.2f ms; 32-bit in-bucket key: 4 x 8 = %.2f ms, 3 x 11 = %.2f ms\n",
               5 * t256, 4 * t1k, 4 * t256, 3 * t2k);
    }
    return 0;
}
