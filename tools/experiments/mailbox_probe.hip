// Round trip host -> resident kernel -> host through a mailbox: where should the REQUEST word live?
//   A: host-mapped pinned memory (the GPU polls across PCIe)       B: fine-grained device memory written by the host through the BAR
// The response always goes to host-mapped memory.  hipcc --offload-arch=gfx950 mailbox_probe.hip -o mailbox_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#include <x86intrin.h>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void pong(volatile uint64_t* req, volatile uint64_t* resp, uint64_t rounds, int inflight) {
    uint64_t seq = 0;
    while (seq < rounds) {
        uint64_t v = __hip_atomic_load((uint64_t*)req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == seq + 1) {
            seq = v;
            __hip_atomic_store((uint64_t*)resp, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// several polls in flight: lanes of the wave take turns, staggered by s_sleep
__global__ void pong_staggered(volatile uint64_t* req, volatile uint64_t* resp, uint64_t rounds) {
    __shared__ uint64_t s_seen;
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_seen = 0;
    __syncthreads();
    for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(20);
    while (true) {
        uint64_t cur = __hip_atomic_load(&s_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur >= rounds) break;
        uint64_t v = __hip_atomic_load((uint64_t*)req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((threadIdx.x & 63) == 0 && v > cur) {
            if (__hip_atomic_compare_exchange_strong(&s_seen, &cur, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                __hip_atomic_store((uint64_t*)resp, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
int run(const char* name, uint64_t* req_host, uint64_t* req_dev, int mode) {
    uint64_t *resp_h = nullptr, *resp_d = nullptr;
    CK(hipHostMalloc((void**)&resp_h, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&resp_d, resp_h, 0));
    *resp_h = 0;
    *req_host = 0;
    const uint64_t rounds = 20000;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (mode == 0) hipLaunchKernelGGL(pong, dim3(1), dim3(64), 0, s, req_dev, resp_d, rounds, 1);
    else hipLaunchKernelGGL(pong_staggered, dim3(1), dim3(256), 0, s, req_dev, resp_d, rounds);
    std::vector<double> lat;
    for (uint64_t k = 1; k <= rounds; ++k) {
        const double t = now_us();
        __atomic_store_n(req_host, k, __ATOMIC_RELEASE);
        _mm_sfence();
        while (__atomic_load_n(resp_h, __ATOMIC_ACQUIRE) != k) _mm_pause();
        lat.push_back(now_us() - t);
    }
    CK(hipStreamSynchronize(s));
    std::sort(lat.begin(), lat.end());
    printf("%-44s median %.2f us  p10 %.2f  p90 %.2f\n", name, lat[lat.size() / 2], lat[lat.size() / 10], lat[lat.size() * 9 / 10]);
    CK(hipStreamDestroy(s));
    CK(hipHostFree(resp_h));
    return 0;
}
int main() {
    CK(hipSetDevice(0));
    uint64_t *a_h = nullptr, *a_d = nullptr;
    CK(hipHostMalloc((void**)&a_h, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&a_d, a_h, 0));
    if (run("A request in host-mapped memory", a_h, a_d, 0)) return 1;
    if (run("A' same, 4 waves polling staggered", a_h, a_d, 1)) return 1;
    uint64_t* b = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&b, 4096, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { printf("fine-grained device memory: %s\n", hipGetErrorString(e)); return 0; }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, b) == hipSuccess) printf("fine-grained: type %d hostPointer %p devicePointer %p\n", (int)at.type, at.hostPointer, at.devicePointer);
    if (getenv("TRY_BAR")) {
        if (run("B request in fine-grained device memory", b, b, 0)) return 1;
        if (run("B' same, 4 waves polling staggered", b, b, 1)) return 1;
    }
    return 0;
}
