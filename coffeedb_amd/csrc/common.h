// common.h — host-side plumbing shared by the HIP translation units of libcoffeedb_gpu.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace cdb {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define CDB_HIP(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            throw ::cdb::Error(std::string("HIP error in " #expr ": ") + hipGetErrorString(e_)); \
    } while (0)

inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
inline int bit_width64(uint64_t x) { return x == 0 ? 0 : 64 - __builtin_clzll(x); }

// Owning device allocation (hipMalloc / hipFree).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        CDB_HIP(hipMalloc(&p, n));
        bytes = n;
    }
    void ensure(size_t n) { if (n > bytes) alloc(n); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Per-kernel timing with HIP events recorded on the launching stream (bench.py's roofline leg reads
// these through cdb_profile_get).  Events are resolved lazily after a stream synchronise.
struct Profiler {
    struct Rec { double ms = 0; uint64_t launches = 0; uint64_t bytes = 0; };
    struct Pending { std::string name; hipEvent_t a, b; uint64_t bytes; };
    bool enabled = false;
    std::map<std::string, Rec> recs;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;

    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        CDB_HIP(hipEventCreate(&e));
        return e;
    }
    // usage: auto t = prof.begin(stream); launch...; prof.end(t, "name", bytes, stream);
    int begin(hipStream_t s) {
        if (!enabled) return -1;
        Pending p{"", get_event(), get_event(), 0};
        CDB_HIP(hipEventRecord(p.a, s));
        pending.push_back(p);
        return (int)pending.size() - 1;
    }
    void end(int tok, const char* name, uint64_t bytes, hipStream_t s) {
        if (tok < 0) return;
        pending[tok].name = name;
        pending[tok].bytes = bytes;
        CDB_HIP(hipEventRecord(pending[tok].b, s));
    }
    // call after the stream has been synchronised
    void resolve() {
        for (auto& p : pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                Rec& r = recs[p.name];
                r.ms += ms;
                r.launches += 1;
                r.bytes += p.bytes;
            }
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        pending.clear();
    }
    void reset() { recs.clear(); }
    ~Profiler() {
        for (auto& p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

}  // namespace cdb
