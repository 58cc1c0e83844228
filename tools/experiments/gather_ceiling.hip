// gather_ceiling.hip — the random-sector rate of one MI355X: the yardstick for the batched binary search (query.hip), whose probes
// are dependent random reads of one suffix-array entry and then of a few text bytes (reference: index.cpp:260-287).
//   independent : every lane issues UNROLL loads at hashed addresses before it uses any of them (memory-level parallelism inside a lane)
//   dependent   : the address of a lane's next load is computed from the value the previous one returned (a bisection's shape;
//                 the only parallelism is the number of lanes resident on the device)
// Every access reads 8 bytes at the start of a random 64-byte sector of a working set of W bytes; swept over W (beyond the 256 MB
// MALL / well beyond) and over the waves resident per CU.  Prints one JSON line per point: G sectors per second.
// build: hipcc -O3 --offload-arch=gfx950 gather_ceiling.hip -o gather_ceiling ; run: ./gather_ceiling [GB ...]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void fill(uint64_t* __restrict__ buf, uint64_t nwords) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) buf[i] = mix(i + 0x9E3779B97F4A7C15ull);
}
// sector index in [0, nsect) from 32 hash bits (nsect < 2^32)
__device__ __forceinline__ uint64_t pick(uint64_t h, uint32_t nsect) { return ((h >> 32) * (uint64_t)nsect) >> 32; }

template <int UNROLL>
__global__ __launch_bounds__(256) void gather_independent(const uint64_t* __restrict__ buf, uint32_t nsect, int rounds, uint64_t* __restrict__ out) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0, h = mix(gid * 0x9E3779B97F4A7C15ull + 1);
    for (int r = 0; r < rounds; ++r) {
        uint64_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            h = h * 6364136223846793005ull + 1442695040888963407ull;
            v[u] = buf[pick(h, nsect) * 8];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    if (acc == 0x1234567ull) out[0] = acc;
}
__global__ __launch_bounds__(256) void gather_dependent(const uint64_t* __restrict__ buf, uint32_t nsect, int steps, uint64_t* __restrict__ out) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t h = mix(gid * 0x9E3779B97F4A7C15ull + 7);
    for (int s = 0; s < steps; ++s) h = buf[pick(h, nsect) * 8] + (uint64_t)s * 0x9E3779B97F4A7C15ull;   // (the loaded word IS random: fill)
    if (h == 0x1234567ull) out[0] = h;
}

int main(int argc, char** argv) {
    std::vector<double> sizes_gb;
    for (int i = 1; i < argc; ++i) sizes_gb.push_back(atof(argv[i]));
    if (sizes_gb.empty()) sizes_gb = {0.125, 4, 40, 100};
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint64_t* out;
    CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (double gb : sizes_gb) {
        const uint64_t bytes = (uint64_t)(gb * 1e9) / 64 * 64;
        const uint64_t nsect64 = bytes / 64;
        if (nsect64 >= (1ull << 32)) { printf("{\"error\": \"working set too large\"}\n"); continue; }
        uint64_t* buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) { printf("{\"working_set_GB\": %.3f, \"error\": \"hipMalloc\"}\n", gb); continue; }
        hipLaunchKernelGGL(fill, dim3(cus * 8), dim3(256), 0, 0, buf, bytes / 8);
        CK(hipDeviceSynchronize());
        const uint32_t nsect = (uint32_t)nsect64;
        const char* only = getenv("GATHER_WPC");            // (one point only: the run under the PMC counters)
        for (int wpc : {1, 2, 4, 8, 16, 32}) {          // waves resident per CU (4 per workgroup of 256)
            if (only && atoi(only) != wpc) continue;
            const int grid = cus * wpc / 4 > 0 ? cus * wpc / 4 : 1;
            const uint64_t lanes = (uint64_t)grid * 256;
            for (int mode = 0; mode < 3; ++mode) {       // 0 dependent, 1 independent x4, 2 independent x16
                const int per_lane = 512;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0, 0));
                    if (mode == 0) hipLaunchKernelGGL(gather_dependent, dim3(grid), dim3(256), 0, 0, buf, nsect, per_lane, out);
                    else if (mode == 1) hipLaunchKernelGGL(gather_independent<4>, dim3(grid), dim3(256), 0, 0, buf, nsect, per_lane / 4, out);
                    else hipLaunchKernelGGL(gather_independent<16>, dim3(grid), dim3(256), 0, 0, buf, nsect, per_lane / 16, out);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const double sectors = (double)lanes * per_lane;
                printf("{\"working_set_GB\": %.3f, \"waves_per_cu\": %d, \"mode\": \"%s\", \"lanes\": %llu, \"loads_per_lane\": %d, \"ms\": %.3f, "
                       "\"G_sectors_per_s\": %.2f, \"GBps_at_64B\": %.1f, \"ns_per_dependent_load\": %.1f}\n",
                       gb, wpc, mode == 0 ? "dependent" : (mode == 1 ? "independent_x4" : "independent_x16"), (unsigned long long)lanes, per_lane, best,
                       sectors / (best * 1e-3) / 1e9, sectors * 64 / (best * 1e-3) / 1e9, mode == 0 ? best * 1e6 / per_lane : 0.0);
                fflush(stdout);
            }
        }
        CK(hipFree(buf));
    }
    return 0;
}
