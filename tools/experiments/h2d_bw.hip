// h2d_bw.hip — what the PCIe link gives a host-to-device copy of 1 GiB of pinned memory in 16 MiB chunks, by the number of
// streams the chunks are dealt to (one stream = one SDMA engine at a time): the floor of cdb_build_view's upload.
// build: hipcc -O3 --offload-arch=gfx950 h2d_bw.hip -o h2d_bw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
    const size_t n = 1ull << 30, chunk = 16u << 20;
    void *h, *d;
    CK(hipHostMalloc(&h, n, hipHostMallocDefault));
    memset(h, 7, n);
    CK(hipMalloc(&d, n));
    for (int ns : {1, 2, 3, 4, 8}) {
        std::vector<hipStream_t> st(ns);
        for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            size_t c = 0;
            for (size_t o = 0; o < n; o += chunk, ++c)
                CK(hipMemcpyAsync((char*)d + o, (char*)h + o, chunk, hipMemcpyHostToDevice, st[c % ns]));
            for (auto& s : st) CK(hipStreamSynchronize(s));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep) best = ms < best ? ms : best;
        }
        printf("%d stream(s): %.2f ms = %.1f GB/s\n", ns, best, n / best / 1e6);
        for (auto& s : st) CK(hipStreamDestroy(s));
    }
    {   // one call for the whole GiB
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("one hipMemcpy: %.2f ms = %.1f GB/s\n", ms, n / ms / 1e6);
    }
    return 0;
}
