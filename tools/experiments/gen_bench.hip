// gen_bench.hip — the C1 initial sort (radix_sort_msd: generated pass on the top digit + 4 segmented passes) outside the
// library, on 2^lg bytes of printable ASCII in 1024-byte documents: the generated pass can be timed and ablated in seconds
// (-DRS_GEN_ABL=bits, see radix_sort.h) without building the library.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../coffeedb_amd/csrc gen_bench.hip -o gen_bench [-DRS_GEN_ABL=n]
// run:   ./gen_bench [log2 n = 30] [rounds = 3] [pair form = 1]
#include "radix_sort.h"

#include <cstdio>
#include <vector>

using namespace cdb;

__device__ __forceinline__ uint32_t mix(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
__global__ void fill_text(uint8_t* t, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) t[i] = (uint8_t)(0x20 + mix((uint32_t)i * 2654435761u + 99u) % 95u);
}
// top-digit histogram by brute force: key of every position (6 symbols, base 96, documents of `dlen` bytes)
__global__ void top_hist(const uint8_t* t, uint64_t n, uint32_t dlen, uint32_t span, unsigned long long m, int pair, unsigned long long* hist) {
    __shared__ uint32_t sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (uint64_t)gridDim.x * 256) {
        const uint32_t rem = dlen - (uint32_t)(p % dlen);
        uint64_t key = 0;
        for (uint32_t q = 0; q < 6; ++q) key = key * 96 + (q < rem ? (uint64_t)(t[p + q] - 0x20 + 1) : 0ull);
        atomicAdd(&sh[(uint32_t)(key / m)], 1u);
    }
    __syncthreads();
    if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}
// sorted by (top * m + k32 = kept >> ... ) — the last pass leaves (u32)(key >> 8) and the low byte: check order + permutation
__global__ void check_kernel(const uint32_t* k, const uint8_t* w, const uint32_t* v, const uint8_t* t, uint64_t n, uint32_t dlen, int bits,
                             unsigned long long* out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = ((uint64_t)k[i] << 8) | w[i];
    if (i + 1 < n) {
        const uint64_t b = ((uint64_t)k[i + 1] << 8) | w[i + 1];
        const uint32_t e1 = v[i + 1];  // equal keys: in text order (every pass is stable)
        const uint64_t p0 = (uint64_t)(v[i] & ((1u << bits) - 1u)) * dlen + (v[i] >> bits), p1 = (uint64_t)(e1 & ((1u << bits) - 1u)) * dlen + (e1 >> bits);
        if (a > b || (a == b && p0 > p1)) atomicAdd(&out[0], 1ull);
    }
    // entry = (off << bits) | doc -> position -> key
    const uint32_t e = v[i];
    const uint64_t doc = e & ((1u << bits) - 1u), off = e >> bits, p = doc * dlen + off;
    const uint32_t rem = dlen - (uint32_t)off;
    uint64_t key = 0;
    for (uint32_t q = 0; q < 6; ++q) key = key * 96 + (q < rem ? (uint64_t)(t[p + q] - 0x20 + 1) : 0ull);
    if (key != a) atomicAdd(&out[1], 1ull);
    if ((i & 1023) == 0) {
        unsigned long long s = 0;
        for (uint64_t j = i; j < n && j < i + 1024; ++j) {
            const uint32_t ej = v[j];
            s += (uint64_t)(ej & ((1u << bits) - 1u)) * dlen + (ej >> bits);
        }
        atomicAdd(&out[2], s);
    }
}

// ---- look-back-free form: per-tile top-digit counts by brute force, scanned over the tiles into tile bases
__global__ void tile_top_count(const uint8_t* t, uint64_t n, uint32_t dlen, unsigned long long m, uint32_t gt, uint32_t* counts) {
    __shared__ uint32_t sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t b0 = (uint64_t)blockIdx.x * gt;
    for (uint64_t p = b0 + threadIdx.x; p < b0 + gt && p < n; p += 256) {
        const uint32_t rem = dlen - (uint32_t)(p % dlen);
        uint64_t key = 0;
        for (uint32_t q = 0; q < 6; ++q) key = key * 96 + (q < rem ? (uint64_t)(t[p + q] - 0x20 + 1) : 0ull);
        atomicAdd(&sh[(uint32_t)(key / m)], 1u);
    }
    __syncthreads();
    counts[(size_t)blockIdx.x * 256 + threadIdx.x] = sh[threadIdx.x];
}
__global__ void tile_scan(const uint32_t* counts, uint32_t tiles, const unsigned long long* digit_start, unsigned long long* base) {
    const uint32_t d = threadIdx.x;
    unsigned long long run = digit_start[d];
    for (uint32_t t = 0; t < tiles; ++t) {
        base[(size_t)t * 256 + d] = run;
        run += counts[(size_t)t * 256 + d];
    }
}
__global__ void check_pass(const uint32_t* k, const uint32_t* v, const uint8_t* t, uint64_t n, uint32_t dlen, int bits, unsigned long long m,
                           const unsigned long long* digit_start, unsigned long long* out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = v[i];
    const uint64_t doc = e & ((1u << bits) - 1u), off = e >> bits, p = doc * dlen + off;
    const uint32_t rem = dlen - (uint32_t)off;
    uint64_t key = 0;
    for (uint32_t q = 0; q < 6; ++q) key = key * 96 + (q < rem ? (uint64_t)(t[p + q] - 0x20 + 1) : 0ull);
    const uint32_t top = (uint32_t)(key / m);
    if (i < digit_start[top] || (top < 255 && i >= digit_start[top + 1] && digit_start[top + 1] > digit_start[top])) atomicAdd(&out[0], 1ull);
    if ((uint32_t)(key - (uint64_t)top * m) != k[i]) atomicAdd(&out[1], 1ull);
    if (i > 0) {
        const uint32_t e0 = v[i - 1];
        const uint64_t p0 = (uint64_t)(e0 & ((1u << bits) - 1u)) * dlen + (e0 >> bits);
        if (i != digit_start[top] && p0 >= p) atomicAdd(&out[2], 1ull);  // inside a bucket: text order (stable)
    }
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? std::atoi(argv[1]) : 30;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 3;
    const int pair = argc > 3 ? std::atoi(argv[3]) : 1;
    const uint64_t n = 1ull << lg;
    const uint32_t dlen = 1024;
    const uint64_t D = n / dlen;
    int bits = 0;
    while ((1ull << bits) < D) ++bits;
    hipStream_t s;
    CDB_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    StreamScope sscope(s);
    uint8_t* text;
    CDB_HIP(hipMalloc(&text, n + 256));
    CDB_HIP(hipMemset(text, 0, n + 256));
    hipLaunchKernelGGL(fill_text, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, text, n);
    std::vector<uint64_t> h_ds(D + 1);
    for (uint64_t d = 0; d <= D; ++d) h_ds[d] = d * dlen;
    uint64_t* d_ds;
    CDB_HIP(hipMalloc(&d_ds, (D + 1) * 8));
    CDB_HIP(hipMemcpy(d_ds, h_ds.data(), (D + 1) * 8, hipMemcpyHostToDevice));
    uint16_t h_map[256] = {0};
    for (int b = 0x20; b < 0x7F; ++b) h_map[b] = (uint16_t)(b - 0x20 + 1);
    uint16_t* d_map;
    CDB_HIP(hipMalloc(&d_map, 512));
    CDB_HIP(hipMemcpy(d_map, h_map, 512, hipMemcpyHostToDevice));
    const uint32_t B = 96;
    const uint64_t P4 = (uint64_t)B * B * B * B;
    const uint32_t span = pair ? (uint32_t)((1ull << 32) / P4) : 0;
    const unsigned long long m = pair ? (unsigned long long)span * P4 : 1ull << 32;
    unsigned long long *d_hist, *d_out;
    CDB_HIP(hipMalloc(&d_hist, 256 * 8));
    CDB_HIP(hipMalloc(&d_out, 4 * 8));
    CDB_HIP(hipMemsetAsync(d_hist, 0, 256 * 8, s));
    hipLaunchKernelGGL(top_hist, dim3(4096), dim3(256), 0, s, text, n, dlen, span, m, pair, d_hist);
    std::vector<uint64_t> h_top(256);
    CDB_HIP(hipMemcpyAsync(h_top.data(), d_hist, 256 * 8, hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
    uint32_t *k[2], *v[2];
    uint8_t *w, *flags;
    for (int i = 0; i < 2; ++i) {
        CDB_HIP(hipMalloc(&k[i], n * 4 + 256));
        CDB_HIP(hipMalloc(&v[i], n * 4 + 256));
    }
    CDB_HIP(hipMalloc(&w, n + 256));
    CDB_HIP(hipMalloc(&flags, n + 256));
    SegEdge* edges;
    CDB_HIP(hipMalloc(&edges, (ceil_div(n, (uint64_t)RS_SEG_TILE) + 256) * 256 * sizeof(SegEdge)));
    if (!rs_atomic_rank_ok(s)) std::printf("one-atomic ranking self-test FAILED on this device\n");
    TextGen gen{text, d_ds, d_map, D, bits, B, 6, 0, true};
    if (pair) {
        gen.msd_pair = true;
        if (!rs_pair_setup(gen, B, span)) { std::printf("pair form not applicable\n"); return 1; }
    } else {
        gen.msd_shift = 32;
    }
    SegFinalKeepArgs keep;
    keep.flags = flags;
    keep.edges = edges;
    keep.kbase = B;
    keep.kmagic = (uint64_t)(~0ull / B) + 1ull;
    RadixWorkspace ws;
    ws.allow_group = true;
    MsdWorkspace mw;
    Profiler prof;
    prof.enabled = true;
    std::map<std::string, double> best;
    {   // the generated pass alone, with the timing ablations of the pass itself (no look-back, linear write-out: safe for
        // any key distribution, which the ablations of the generator (RS_GEN_ABL) need)
#ifndef GEN_NT
#define GEN_NT 1024
#endif
#ifndef GEN_GROUP
#define GEN_GROUP RS_GROUP
#endif
#ifndef GEN_TICKET
#define GEN_TICKET true
#endif
#ifndef GEN_LB
#define GEN_LB 4
#endif
        #ifndef GEN_ABL
#define GEN_ABL 3
#endif
        using CfgA = RsCfg<16, true, true, GEN_NT, false, 1, GEN_ABL, GEN_LB, false, true, GEN_TICKET, 1, GEN_GROUP>;
        using CfgN = RsCfg<16, true, true, GEN_NT, false, 1, 0, GEN_LB, false, true, GEN_TICKET, 1, GEN_GROUP>;
        constexpr uint64_t GT = 16 * GEN_NT;  // tile of the stand-alone generated pass
        const uint32_t tiles = (uint32_t)ceil_div(n, GT);
        ws.prepare(n + 256 * (uint64_t)RS_SEG_TILE, (int)GT, s);
        ws.tile_doc.ensure(((size_t)tiles + 1) * 8);
        hipLaunchKernelGGL(rs_tiledoc_kernel, dim3((unsigned)ceil_div((uint64_t)tiles + 1, 256)), dim3(256), 0, s, (const uint64_t*)d_ds, D, n,
                           GT, (uint64_t)tiles, ws.tile_doc.as<uint64_t>());
        TextGen g2 = gen;
        g2.tile_doc = ws.tile_doc.as<uint64_t>();
        unsigned long long* d_start = ws.hist.as<unsigned long long>();
        CDB_HIP(hipMemcpyAsync(d_start, h_top.data(), 256 * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(rs_digit_start_kernel, dim3(1), dim3(256), 0, s, d_start, d_start + RS_MAX_PASSES * 256);
        const uint32_t grid = (uint32_t)(ceil_div(tiles, 8u * GEN_GROUP) * 8u * GEN_GROUP);
        for (int abl = 0; abl < 2; ++abl) {
            if (RS_GEN_ABL != 0 && abl == 0) continue;  // (an ablated generator only with the safe write-out)
            double bms = 1e30;
            for (int r = 0; r <= rounds; ++r) {
                const uint32_t e = ws.next_epoch(s);
                hipEvent_t a, b;
                CDB_HIP(hipEventCreate(&a)); CDB_HIP(hipEventCreate(&b));
                CDB_HIP(hipEventRecord(a, s));
                if (pair) {
                    TextGenPair gp;
                    static_cast<TextGen&>(gp) = g2;
                    if (abl)
                        hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CfgA, TextGenPair, uint8_t>), dim3(grid), dim3(GEN_NT), 0, s, (const uint32_t*)nullptr,
                                           k[1], (const uint32_t*)nullptr, v[1], n, 0, 0xFFu, (const unsigned long long*)(d_start + RS_MAX_PASSES * 256),
                                           ws.status.as<uint64_t>(), ws.xticket_ptr(e), e, ws.err_ptr(), gp, (const uint8_t*)nullptr, (uint8_t*)nullptr, 0);
                    else
                        hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CfgN, TextGenPair, uint8_t>), dim3(grid), dim3(GEN_NT), 0, s, (const uint32_t*)nullptr,
                                           k[1], (const uint32_t*)nullptr, v[1], n, 0, 0xFFu, (const unsigned long long*)(d_start + RS_MAX_PASSES * 256),
                                           ws.status.as<uint64_t>(), ws.xticket_ptr(e), e, ws.err_ptr(), gp, (const uint8_t*)nullptr, (uint8_t*)nullptr, 0);
                } else if (abl)
                    hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CfgA, TextGen, uint8_t>), dim3(grid), dim3(GEN_NT), 0, s, (const uint32_t*)nullptr,
                                       k[1], (const uint32_t*)nullptr, v[1], n, 0, 0xFFu, (const unsigned long long*)(d_start + RS_MAX_PASSES * 256),
                                       ws.status.as<uint64_t>(), ws.xticket_ptr(e), e, ws.err_ptr(), g2, (const uint8_t*)nullptr, (uint8_t*)nullptr, 0);
                else
                    hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CfgN, TextGen, uint8_t>), dim3(grid), dim3(GEN_NT), 0, s, (const uint32_t*)nullptr,
                                       k[1], (const uint32_t*)nullptr, v[1], n, 0, 0xFFu, (const unsigned long long*)(d_start + RS_MAX_PASSES * 256),
                                       ws.status.as<uint64_t>(), ws.xticket_ptr(e), e, ws.err_ptr(), g2, (const uint8_t*)nullptr, (uint8_t*)nullptr, 0);
                CDB_HIP(hipEventRecord(b, s));
                CDB_HIP(hipStreamSynchronize(s));
                float ms = 0;
                CDB_HIP(hipEventElapsedTime(&ms, a, b));
                if (r > 0) bms = std::min(bms, (double)ms);
            }
            std::printf("generated pass alone, RS_GEN_ABL=%d, %s: %.3f ms\n", RS_GEN_ABL, abl ? "no look-back + linear write-out" : "as in production", bms);
        }
    }
    if (pair && RS_GEN_ABL == 0) {   // the same pass without status words or look-back: tile bases from counted per-tile digits
        constexpr uint64_t GT = 16 * GEN_NT;
        const uint32_t tiles = (uint32_t)ceil_div(n, GT);
        uint32_t* d_cnt;
        unsigned long long* d_base;
        CDB_HIP(hipMalloc(&d_cnt, (size_t)tiles * 256 * 4));
        CDB_HIP(hipMalloc(&d_base, (size_t)tiles * 256 * 8));
        unsigned long long* d_start = ws.hist.as<unsigned long long>();
        hipLaunchKernelGGL(tile_top_count, dim3(tiles), dim3(256), 0, s, text, n, dlen, m, (uint32_t)GT, d_cnt);
        hipLaunchKernelGGL(tile_scan, dim3(1), dim3(256), 0, s, (const uint32_t*)d_cnt, tiles, (const unsigned long long*)(d_start + RS_MAX_PASSES * 256), d_base);
        CDB_HIP(hipStreamSynchronize(s));
        using CfgN = RsCfg<16, true, true, GEN_NT, false, 1, 0, GEN_LB, false, true, GEN_TICKET, 1, GEN_GROUP>;
        TextGenPair gp;
        static_cast<TextGen&>(gp) = gen;
        gp.tile_doc = ws.tile_doc.as<uint64_t>();
        gp.tile_base = d_base;
        const uint32_t grid = (uint32_t)(ceil_div(tiles, 8u * GEN_GROUP) * 8u * GEN_GROUP);
        double bms = 1e30;
        for (int r = 0; r <= rounds; ++r) {
            const uint32_t e = ws.next_epoch(s);
            hipEvent_t a, b;
            CDB_HIP(hipEventCreate(&a)); CDB_HIP(hipEventCreate(&b));
            CDB_HIP(hipEventRecord(a, s));
            hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CfgN, TextGenPair, uint8_t>), dim3(grid), dim3(GEN_NT), 0, s, (const uint32_t*)nullptr,
                               k[1], (const uint32_t*)nullptr, v[1], n, 0, 0xFFu, (const unsigned long long*)(d_start + RS_MAX_PASSES * 256),
                               ws.status.as<uint64_t>(), ws.xticket_ptr(e), e, ws.err_ptr(), gp, (const uint8_t*)nullptr, (uint8_t*)nullptr, 0);
            CDB_HIP(hipEventRecord(b, s));
            CDB_HIP(hipStreamSynchronize(s));
            float ms = 0;
            CDB_HIP(hipEventElapsedTime(&ms, a, b));
            if (r > 0) bms = std::min(bms, (double)ms);
        }
        CDB_HIP(hipMemsetAsync(d_out, 0, 4 * 8, s));
        hipLaunchKernelGGL(check_pass, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, (const uint32_t*)k[1], (const uint32_t*)v[1], text, n, dlen, bits, m,
                           (const unsigned long long*)(d_start + RS_MAX_PASSES * 256), d_out);
        unsigned long long out[4];
        CDB_HIP(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        std::printf("generated pass alone, counted tile bases (no look-back), %d threads x 16: %.3f ms   check: wrong bucket %llu, wrong key %llu, out of text order %llu\n",
                    GEN_NT, bms, out[0], out[1], out[2]);
    }
    if (RS_GEN_ABL != 0 || GEN_NT != 1024) return 0;  // (GEN_NT=512 needs BLK in radix_sort.h relaxed to NT == 512: an experiment)
    for (int r = 0; r <= rounds; ++r) {
        prof.reset();
        SortStats st;
        radix_sort_msd(s, ws, mw, prof, k[0], k[1], v[0], v[1], w, n, h_top.data(), gen, m, keep, &st);
        CDB_HIP(hipStreamSynchronize(s));
        radix_check_error(s, ws);
        prof.resolve();
        for (auto& kv : prof.recs) {
            const double ms = kv.second.ms / (double)kv.second.launches;
            if (r > 0) best[kv.first] = best.count(kv.first) ? std::min(best[kv.first], ms) : ms;
        }
        if (r == 0) {
            CDB_HIP(hipMemsetAsync(d_out, 0, 4 * 8, s));
            hipLaunchKernelGGL(check_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, k[1], w, v[1], text, n, dlen, bits, d_out);
            unsigned long long out[4];
            CDB_HIP(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, s));
            CDB_HIP(hipStreamSynchronize(s));
            const unsigned long long want = (unsigned long long)(n - 1) * n / 2;
            std::printf("%s: inversions %llu, wrong keys %llu, position sum %s\n", out[0] == 0 && out[1] == 0 && out[2] == want ? "ok" : "WRONG",
                        out[0], out[1], out[2] == want ? "ok" : "WRONG");
        }
    }
    for (auto& kv : best) std::printf("%-36s %.3f ms per launch\n", kv.first.c_str(), kv.second);
    return 0;
}
