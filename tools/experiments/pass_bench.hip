// pass_bench.hip — stand-alone A/B of rs_onesweep_kernel configurations on the C1 record shape (u32 key, u32 entry,
// u8 low digit; 2^30 records), outside the library: compiles in seconds, so kernel ideas can be tried quickly.
// Every configuration sorts the same pseudo-random records (1 leading pass on the digit byte + 4 passes over the
// key) and is checked for sortedness and for the permutation checksum.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../coffeedb_amd/csrc pass_bench.hip -o pass_bench
// run:   ./pass_bench [log2 n = 30] [rounds = 3]
#include "radix_sort.h"

#include <cstdio>
#include <vector>

using namespace cdb;

__device__ __forceinline__ uint32_t mix(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
__global__ void fill_kernel(uint32_t* k, uint32_t* v, uint8_t* w, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = mix((uint32_t)i * 2654435761u + 12345u);
    k[i] = h;
    v[i] = (uint32_t)i;
    w[i] = (uint8_t)mix(h + 77u);
}
__global__ void hist_kernel(const uint32_t* k, const uint8_t* w, uint64_t n, unsigned long long* hist) {
    __shared__ uint32_t sh[5 * 256];
    for (int i = threadIdx.x; i < 5 * 256; i += 256) sh[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t kk = k[i];
        atomicAdd(&sh[w[i]], 1u);
        for (int p = 0; p < 4; ++p) atomicAdd(&sh[(p + 1) * 256 + ((kk >> (8 * p)) & 255u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 5 * 256; i += 256)
        if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}
// out[0] = adjacent inversions of (key, digit), out[1] = sum of entries, out[2] = entries whose record does not match
__global__ void check_kernel(const uint32_t* k, const uint32_t* v, const uint8_t* w, uint64_t n, unsigned long long* out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = ((uint64_t)k[i] << 8) | w[i];
    if (i + 1 < n) {
        const uint64_t b = ((uint64_t)k[i + 1] << 8) | w[i + 1];
        if (a > b || (a == b && v[i] > v[i + 1])) atomicAdd(&out[0], 1ull);
    }
    const uint32_t h = mix(v[i] * 2654435761u + 12345u);
    if (k[i] != h || w[i] != (uint8_t)mix(h + 77u)) atomicAdd(&out[2], 1ull);
    if ((i & 1023) == 0) {
        unsigned long long s = 0;
        for (uint64_t j = i; j < n && j < i + 1024; ++j) s += v[j];
        atomicAdd(&out[1], s);
    }
}

struct Bufs {
    uint32_t *k[2], *v[2];
    uint8_t* w[2];
    unsigned long long* d_hist;
    unsigned long long* d_out;
    std::vector<uint64_t> h_hist;
    uint64_t n;
};

template <typename Cfg>
void run(const char* name, Bufs& b, hipStream_t s, int rounds) {
    RadixWorkspace ws;
    Profiler prof;
    prof.enabled = true;
    double best = 1e30;
    bool ok = true;
    for (int r = 0; r < rounds + 1; ++r) {
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)ceil_div(b.n, 256)), dim3(256), 0, s, b.k[0], b.v[0], b.w[0], b.n);
        prof.reset();
        SortStats st;
        int sel = radix_sort_cfg<uint32_t, uint32_t, Cfg, NoGen, uint8_t>(s, ws, prof, b.k[0], b.k[1], b.v[0], b.v[1], b.n, 0, 32, &st, 8,
                                                                       b.h_hist.data(), (const NoGen*)nullptr, b.w[0], b.w[1], 1);
        CDB_HIP(hipStreamSynchronize(s));
        radix_check_error(s, ws);
        prof.resolve();
        double ms = 0;
        uint64_t launches = 0;
        for (auto& kv : prof.recs)
            if (kv.first.rfind("rs_onesweep", 0) == 0) { ms += kv.second.ms; launches += kv.second.launches; }
        if (r > 0) best = std::min(best, ms / (double)launches);
        if (r == 0) {
            CDB_HIP(hipMemsetAsync(b.d_out, 0, 4 * sizeof(unsigned long long), s));
            hipLaunchKernelGGL(check_kernel, dim3((unsigned)ceil_div(b.n, 256)), dim3(256), 0, s, b.k[sel], b.v[sel], b.w[sel], b.n, b.d_out);
            unsigned long long out[4];
            CDB_HIP(hipMemcpyAsync(out, b.d_out, sizeof(out), hipMemcpyDeviceToHost, s));
            CDB_HIP(hipStreamSynchronize(s));
            const unsigned long long want = (unsigned long long)(b.n - 1) * b.n / 2;
            ok = out[0] == 0 && out[1] == want && out[2] == 0;
            if (!ok) std::printf("  [%s] WRONG: inversions %llu sum %llu (want %llu) mismatched %llu\n", name, out[0], out[1], want, out[2]);
        }
    }
    std::printf("%-44s %s  %.3f ms per pass  %.0f GB/s algorithmic (18 B x n)\n", name, ok ? "ok   " : "WRONG", best, 18.0 * b.n / (best * 1e-3) / 1e9);
    std::fflush(stdout);
    ws.release();
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? std::atoi(argv[1]) : 30;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 3;
    Bufs b;
    b.n = 1ull << lg;
    hipStream_t s;
    CDB_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    StreamScope sscope(s);
    for (int i = 0; i < 2; ++i) {
        CDB_HIP(hipMalloc(&b.k[i], b.n * 4 + 256));
        CDB_HIP(hipMalloc(&b.v[i], b.n * 4 + 256));
        CDB_HIP(hipMalloc(&b.w[i], b.n + 256));
    }
    CDB_HIP(hipMalloc(&b.d_hist, 5 * 256 * 8));
    CDB_HIP(hipMalloc(&b.d_out, 4 * 8));
    CDB_HIP(hipMemsetAsync(b.d_hist, 0, 5 * 256 * 8, s));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)ceil_div(b.n, 256)), dim3(256), 0, s, b.k[0], b.v[0], b.w[0], b.n);
    hipLaunchKernelGGL(hist_kernel, dim3(2048), dim3(256), 0, s, b.k[0], b.w[0], b.n, b.d_hist);
    b.h_hist.resize(5 * 256);
    CDB_HIP(hipMemcpyAsync(b.h_hist.data(), b.d_hist, 5 * 256 * 8, hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
    if (!rs_atomic_rank_ok(s)) std::printf("one-atomic ranking self-test FAILED on this device\n");
    // IPT, REUSE, EARLYV, NT, NONTEMP, MINW, ABL, LB, DMA, ATOMRANK, TICKET, LOAD
#define RUN(...) run<RsCfg<__VA_ARGS__>>(#__VA_ARGS__, b, s, rounds)
#include "pass_bench_cfgs.inc"
    return 0;
}
