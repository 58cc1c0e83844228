// How long does the driver take to hand out VRAM?  One big block vs many, first touch vs re-allocation after hipFree,
// and the virtual-memory API (hipMemCreate + hipMemMap) for comparison.  hipcc --offload-arch=gfx950 alloc_cost.hip -o alloc_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    const size_t GiB = 1ull << 30;
    for (int round = 0; round < 2; ++round) {
        double t = now();
        void* p = nullptr;
        CK(hipMalloc(&p, 64 * GiB));
        printf("round %d: hipMalloc(64 GiB) %.1f ms\n", round, now() - t);
        t = now();
        CK(hipMemsetAsync(p, 0, 64 * GiB, 0));
        CK(hipDeviceSynchronize());
        printf("         memset 64 GiB %.1f ms\n", now() - t);
        t = now();
        CK(hipFree(p));
        printf("         hipFree %.1f ms\n", now() - t);
    }
    {
        double t = now();
        std::vector<void*> v(16);
        for (auto& p : v) CK(hipMalloc(&p, 4 * GiB));
        printf("16 x hipMalloc(4 GiB) %.1f ms\n", now() - t);
        t = now();
        for (auto& p : v) CK(hipFree(p));
        printf("16 x hipFree %.1f ms\n", now() - t);
    }
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        printf("vmm granularity %zu\n", gran);
        double t = now();
        void* va = nullptr;
        CK(hipMemAddressReserve(&va, 64 * GiB, 0, nullptr, 0));
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, 64 * GiB, &prop, 0));
        double t1 = now();
        CK(hipMemMap(va, 64 * GiB, 0, h, 0));
        hipMemAccessDesc ad = {};
        ad.location = prop.location;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, 64 * GiB, &ad, 1));
        printf("vmm: reserve+create %.1f ms, map+access %.1f ms\n", t1 - t, now() - t1);
        t = now();
        CK(hipMemsetAsync(va, 0, 64 * GiB, 0));
        CK(hipDeviceSynchronize());
        printf("     memset %.1f ms\n", now() - t);
        CK(hipMemUnmap(va, 64 * GiB));
        CK(hipMemRelease(h));
        CK(hipMemAddressFree(va, 64 * GiB));
    }
    return 0;
}
