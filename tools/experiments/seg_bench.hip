// seg_bench.hip — does a radix pass get faster when its scatter stays inside a bucket?  2^lg records (u32 key, u32 entry),
// 4 LSD passes over the key: (a) one sort over the whole array, (b) the same records pre-partitioned into S equal
// segments, every pass ONE segmented launch (tile -> segment map, per-segment digit starts, look-back restarts per
// segment) — the shape an MSD-first build of C1 would have (top digit first, then LSD inside its buckets).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../coffeedb_amd/csrc seg_bench.hip -o seg_bench
// run:   ./seg_bench [log2 n = 30] [segments = 183] [rounds = 3]
#include "radix_sort.h"

#include <cstdio>
#include <vector>

using namespace cdb;

__device__ __forceinline__ uint32_t mix(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
__global__ void fill_kernel(uint32_t* k, uint32_t* v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    k[i] = mix((uint32_t)i * 2654435761u + 12345u);
    v[i] = (uint32_t)i;
}
__global__ void pack_kernel(const uint32_t* k, const uint32_t* v, uint64_t* r, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) r[i] = ((uint64_t)k[i] << 32) | v[i];
}
// hist[g][p][d] over the elements of segment g
__global__ void seg_hist_kernel(const uint32_t* k, const SegInfo* segs, unsigned long long* hist) {
    __shared__ uint32_t sh[4 * 256];
    const uint32_t g = blockIdx.y;
    for (int i = threadIdx.x; i < 4 * 256; i += 256) sh[i] = 0;
    __syncthreads();
    const uint64_t b = segs[g].begin, e = segs[g].end;
    for (uint64_t i = b + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < e; i += (uint64_t)gridDim.x * 256) {
        const uint32_t kk = k[i];
        for (int p = 0; p < 4; ++p) atomicAdd(&sh[p * 256 + ((kk >> (8 * p)) & 255u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 256; i += 256)
        if (sh[i]) atomicAdd(&hist[((size_t)g * 8 + i / 256) * 256 + i % 256], (unsigned long long)sh[i]);
}
__global__ void check_kernel(const uint32_t* k, const uint32_t* v, const SegInfo* segs, uint32_t nseg, uint64_t n, unsigned long long* out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i + 1 < n) {
        bool boundary = false;
        for (uint32_t g = 0; g < nseg; ++g) boundary |= segs[g].begin == i + 1;
        if (!boundary && (k[i] > k[i + 1] || (k[i] == k[i + 1] && v[i] > v[i + 1]))) atomicAdd(&out[0], 1ull);
    }
    if (k[i] != mix(v[i] * 2654435761u + 12345u)) atomicAdd(&out[2], 1ull);
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? std::atoi(argv[1]) : 30;
    const uint32_t nseg_arg = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 183u;
    const int rounds = argc > 3 ? std::atoi(argv[3]) : 3;
    const uint64_t n = 1ull << lg;
    hipStream_t s;
    CDB_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    StreamScope sscope(s);
    uint32_t *k[2], *v[2];
    for (int i = 0; i < 2; ++i) {
        CDB_HIP(hipMalloc(&k[i], n * 4 + 256));
        CDB_HIP(hipMalloc(&v[i], n * 4 + 256));
    }
    unsigned long long* d_out;
    CDB_HIP(hipMalloc(&d_out, 4 * 8));
    if (!rs_atomic_rank_ok(s)) std::printf("one-atomic ranking self-test FAILED on this device\n");
#ifdef SEG_NT        // smaller workgroups: 16 keys per thread, keys and values staged at once — two or more workgroups per CU
    using CfgG = RsCfg<16, false, true, SEG_NT, false, 1, 0, SEG_LB, false, true, true, 1, RS_GROUP>;
#elif defined(SEG_IPT)       // other tile sizes (keys and values share the staging buffer); SEG_MINW = 8: two 1024-thread workgroups per CU
#ifndef SEG_MINW
#define SEG_MINW 1
#endif
#ifndef SEG_EARLYV
#define SEG_EARLYV true
#endif
    using CfgG = RsCfg<SEG_IPT, true, SEG_EARLYV, 1024, false, SEG_MINW, 0, 4, false, true, true, 1, RS_GROUP>;
#elif defined(SEG_NOREUSE)   // keys and values both staged at once (128 KB): one write-out phase instead of two
    using CfgG = RsCfg<16, false, true, 1024, false, 1, 0, 4, false, true, true, 1, RS_GROUP>;
#else
    using CfgG = RsCfg<16, true, true, 1024, false, 1, 0, 4, false, true, true, 1, RS_GROUP>;
#endif
    Profiler prof;
    prof.enabled = true;
    // (a) one sort over everything
    {
        RadixWorkspace ws;
        ws.allow_group = true;
        double best = 1e30;
        for (int r = 0; r <= rounds; ++r) {
            hipLaunchKernelGGL(fill_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, k[0], v[0], n);
            prof.reset();
            SortStats st;
            radix_sort_cfg<uint32_t, uint32_t, CfgG>(s, ws, prof, k[0], k[1], v[0], v[1], n, 0, 32, &st, 8);
            CDB_HIP(hipStreamSynchronize(s));
            radix_check_error(s, ws);
            prof.resolve();
            double ms = 0;
            uint64_t launches = 0;
            for (auto& kv : prof.recs)
                if (kv.first.rfind("rs_onesweep", 0) == 0) { ms += kv.second.ms; launches += kv.second.launches; }
            if (r > 0) best = std::min(best, ms / (double)launches);
        }
        std::printf("whole array, (u32, u32) records:        %.3f ms per pass  %.0f GB/s algorithmic (16 B x n)\n", best, 16.0 * n / (best * 1e-3) / 1e9);
        ws.release();
    }
#ifdef SEG_AOS
    {   // (c) array of structures: one u64 per record (key << 32 | entry), key-only passes on the high word, segmented
        using CfgK = RsCfg<16, false, false, 1024, false, 1, 0, 4, false, true, true, 1, RS_GROUP>;
        const uint32_t nseg = nseg_arg;
        std::vector<SegInfo> h_segs(nseg);
        uint32_t tiles = 0;
        for (uint32_t g = 0; g < nseg; ++g) {
            const uint64_t b = n * g / nseg, e = n * (g + 1) / nseg;
            h_segs[g] = SegInfo{b, e, tiles, 0u, 0ull, 0ull};
            tiles += (uint32_t)ceil_div(e - b, (uint64_t)RS_SEG_TILE);
        }
        SegInfo* d_segs; uint32_t* d_tile_seg; unsigned long long *d_hist, *d_starts;
        CDB_HIP(hipMalloc(&d_segs, nseg * sizeof(SegInfo)));
        CDB_HIP(hipMalloc(&d_tile_seg, tiles * 4));
        CDB_HIP(hipMalloc(&d_hist, (size_t)nseg * 8 * 256 * 8));
        CDB_HIP(hipMalloc(&d_starts, (size_t)nseg * 8 * 256 * 8));
        CDB_HIP(hipMemcpy(d_segs, h_segs.data(), nseg * sizeof(SegInfo), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(rs_seg_tilemap_kernel, dim3((unsigned)ceil_div(tiles, 256u)), dim3(256), 0, s, d_segs, nseg, tiles, d_tile_seg);
        RadixWorkspace ws;
        ws.prepare((uint64_t)tiles * RS_SEG_TILE, RS_SEG_TILE, s);
        SegArgs sa; sa.tile_seg = d_tile_seg; sa.segs = d_segs; sa.tiles = tiles; sa.start_stride = 8 * 256;
        const uint32_t grid = (uint32_t)(ceil_div(tiles, 8u * RS_GROUP) * 8u * RS_GROUP);
        uint64_t* r[2] = {reinterpret_cast<uint64_t*>(k[0]), reinterpret_cast<uint64_t*>(k[1])};  // (k[0], v[0] are separate blocks:
        CDB_HIP(hipFree(v[0])); CDB_HIP(hipFree(v[1]));                                               //  reallocate as 8 n bytes)
        CDB_HIP(hipFree(k[0])); CDB_HIP(hipFree(k[1]));
        CDB_HIP(hipMalloc(&r[0], n * 8 + 256)); CDB_HIP(hipMalloc(&r[1], n * 8 + 256));
        uint32_t *tk, *tv;
        CDB_HIP(hipMalloc(&tk, n * 4 + 256)); CDB_HIP(hipMalloc(&tv, n * 4 + 256));
        double best = 1e30;
        for (int rr = 0; rr <= rounds; ++rr) {
            hipLaunchKernelGGL(fill_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, tk, tv, n);
            hipLaunchKernelGGL(pack_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, tk, tv, r[0], n);
            CDB_HIP(hipMemsetAsync(d_hist, 0, (size_t)nseg * 8 * 256 * 8, s));
            hipLaunchKernelGGL(seg_hist_kernel, dim3(std::max(1u, 2048u / nseg), nseg), dim3(256), 0, s, tk, d_segs, d_hist);
            hipLaunchKernelGGL(rs_seg_digit_start_kernel, dim3(nseg), dim3(256), 0, s, d_hist, d_segs, 4, d_starts);
            prof.reset();
            int cur = 0;
            for (int p = 0; p < 4; ++p) {
                const uint32_t e = ws.next_epoch(s);
                int t = prof.begin(s);
                hipLaunchKernelGGL((rs_onesweep_kernel<uint64_t, NoVal, CfgK, NoGen, NoVal, SegArgs>), dim3(grid), dim3(1024), 0, s,
                                   (const uint64_t*)r[cur], r[cur ^ 1], (const NoVal*)nullptr, (NoVal*)nullptr, n, 32 + 8 * p, 0xFFu,
                                   (const unsigned long long*)(d_starts + (size_t)p * 256), ws.status.as<uint64_t>(), ws.xticket_ptr(e), e,
                                   ws.err_ptr(), NoGen(), (const NoVal*)nullptr, (NoVal*)nullptr, -1, sa);
                prof.end(t, "rs_aos", 2 * n * 8, s);
                cur ^= 1;
            }
            CDB_HIP(hipStreamSynchronize(s));
            radix_check_error(s, ws);
            prof.resolve();
            double ms = 0; uint64_t launches = 0;
            for (auto& kv : prof.recs)
                if (kv.first.rfind("rs_aos", 0) == 0) { ms += kv.second.ms; launches += kv.second.launches; }
            if (rr > 0) best = std::min(best, ms / (double)launches);
        }
        std::printf("%4u segments, u64 (key << 32 | entry) records: %.3f ms per pass  %.0f GB/s algorithmic (16 B x n)\n", nseg, best, 16.0 * n / (best * 1e-3) / 1e9);
        return 0;
    }
#endif
    // (b) segments
    for (uint32_t nseg : {nseg_arg, 16u, 1u}) {
        std::vector<SegInfo> h_segs(nseg);
        uint32_t tiles = 0;
        for (uint32_t g = 0; g < nseg; ++g) {
            const uint64_t b = n * g / nseg, e = n * (g + 1) / nseg;
            h_segs[g] = SegInfo{b, e, tiles, 0u, 0ull, 0ull};
            tiles += (uint32_t)ceil_div(e - b, (uint64_t)(CfgG::NT * CfgG::IPT));
        }
        SegInfo* d_segs;
        uint32_t* d_tile_seg;
        unsigned long long *d_hist, *d_starts;
        CDB_HIP(hipMalloc(&d_segs, nseg * sizeof(SegInfo)));
        CDB_HIP(hipMalloc(&d_tile_seg, tiles * 4));
        CDB_HIP(hipMalloc(&d_hist, (size_t)nseg * 8 * 256 * 8));
        CDB_HIP(hipMalloc(&d_starts, (size_t)nseg * 8 * 256 * 8));
        CDB_HIP(hipMemcpy(d_segs, h_segs.data(), nseg * sizeof(SegInfo), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(rs_seg_tilemap_kernel, dim3((unsigned)ceil_div(tiles, 256u)), dim3(256), 0, s, d_segs, nseg, tiles, d_tile_seg);
        RadixWorkspace ws;
        ws.prepare((uint64_t)tiles * (CfgG::NT * CfgG::IPT), CfgG::NT * CfgG::IPT, s);
        SegArgs sa;
        sa.tile_seg = d_tile_seg;
        sa.segs = d_segs;
        sa.tiles = tiles;
        sa.start_stride = 8 * 256;
        const uint32_t grid = (uint32_t)(ceil_div(tiles, 8u * RS_GROUP) * 8u * RS_GROUP);
        double best = 1e30;
        bool ok = true;
        for (int r = 0; r <= rounds; ++r) {
            hipLaunchKernelGGL(fill_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, k[0], v[0], n);
            CDB_HIP(hipMemsetAsync(d_hist, 0, (size_t)nseg * 8 * 256 * 8, s));
            hipLaunchKernelGGL(seg_hist_kernel, dim3(std::max(1u, 2048u / nseg), nseg), dim3(256), 0, s, k[0], d_segs, d_hist);
            hipLaunchKernelGGL(rs_seg_digit_start_kernel, dim3(nseg), dim3(256), 0, s, d_hist, d_segs, 4, d_starts);
            prof.reset();
            int cur = 0;
            for (int p = 0; p < 4; ++p) {
                const uint32_t e = ws.next_epoch(s);
                int t = prof.begin(s);
                hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CfgG, NoGen, NoVal, SegArgs>), dim3(grid), dim3(CfgG::NT), 0, s,
                                   (const uint32_t*)k[cur], k[cur ^ 1], (const uint32_t*)v[cur], v[cur ^ 1], n, 8 * p, 0xFFu,
                                   (const unsigned long long*)(d_starts + (size_t)p * 256), ws.status.as<uint64_t>(), ws.xticket_ptr(e), e,
                                   ws.err_ptr(), NoGen(), (const NoVal*)nullptr, (NoVal*)nullptr, -1, sa);
                prof.end(t, "rs_seg", 2 * n * 8, s);
                cur ^= 1;
            }
            CDB_HIP(hipStreamSynchronize(s));
            radix_check_error(s, ws);
            prof.resolve();
            double ms = 0;
            uint64_t launches = 0;
            for (auto& kv : prof.recs)
                if (kv.first.rfind("rs_seg", 0) == 0) { ms += kv.second.ms; launches += kv.second.launches; }
            if (r > 0) best = std::min(best, ms / (double)launches);
            if (r == 0) {
                CDB_HIP(hipMemsetAsync(d_out, 0, 4 * 8, s));
                hipLaunchKernelGGL(check_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, k[cur], v[cur], d_segs, nseg, n, d_out);
                unsigned long long out[4];
                CDB_HIP(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, s));
                CDB_HIP(hipStreamSynchronize(s));
                ok = out[0] == 0 && out[2] == 0;
                if (!ok) std::printf("  WRONG: inversions %llu mismatched %llu\n", out[0], out[2]);
            }
        }
        std::printf("%4u segments, (u32, u32) records: %s  %.3f ms per pass  %.0f GB/s algorithmic (16 B x n)\n", nseg, ok ? "ok   " : "WRONG", best,
                    16.0 * n / (best * 1e-3) / 1e9);
        std::fflush(stdout);
        ws.release();
        CDB_HIP(hipFree(d_segs)); CDB_HIP(hipFree(d_tile_seg)); CDB_HIP(hipFree(d_hist)); CDB_HIP(hipFree(d_starts));
    }
    return 0;
}
