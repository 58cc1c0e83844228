// common.h — host-side plumbing shared by the HIP translation units of libcoffeedb_gpu.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace cdb {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
// the bucket-wise build chose a key form that exists in the sweep kernels only (variable-length keys, partial symbol) and then had
// to take another records form: build_suffix_array redoes the build with plain dense keys (counted: stat "dense_key_retries")
struct RetryWithDenseKeys : Error {
    using Error::Error;
};

#define CDB_HIP(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            throw ::cdb::Error(std::string("HIP error in " #expr ": ") + hipGetErrorString(e_)); \
    } while (0)

inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
inline int bit_width64(uint64_t x) { return x == 0 ? 0 : 64 - __builtin_clzll(x); }

// The stream the calling thread is currently issuing device work on (set by every C-ABI entry point through
// StreamScope).  DevPool uses it to make block reuse stream-aware: a block released while kernels of that stream
// may still touch it is tagged (stream, event) and handed to ANOTHER stream only once the event has completed;
// the same stream may take it back at once (stream order).
inline thread_local hipStream_t tls_stream = nullptr;
// Library calls in flight (every C-ABI entry that does device work counts itself in: capi.hip guarded(), shards.hip guarded_on()).
// The order proof behind a build (verify.hip) launches its slices in the gaps: a batched search is bound by the memory system's
// random-sector rate, and a sweep that reads a random text sector per suffix beside it cost the query 5 x its time (8 GiB Zipf:
// 18 -> 100 ms per million patterns).
inline std::atomic<int>& foreground_calls() {
    static std::atomic<int> c{0};
    return c;
}
struct ForegroundCall {
    ForegroundCall() { foreground_calls().fetch_add(1, std::memory_order_acq_rel); }
    ~ForegroundCall() { foreground_calls().fetch_sub(1, std::memory_order_acq_rel); }
    ForegroundCall(const ForegroundCall&) = delete;
    ForegroundCall& operator=(const ForegroundCall&) = delete;
};

struct StreamScope {
    hipStream_t prev;
    explicit StreamScope(hipStream_t s) : prev(tls_stream) { tls_stream = s; }
    ~StreamScope() { tls_stream = prev; }
    StreamScope(const StreamScope&) = delete;
    StreamScope& operator=(const StreamScope&) = delete;
};

// Process-wide cache of device blocks.  hipMalloc of multi-GiB buffers costs ~30 ms per GiB on MI355X
// (the driver maps and clears VRAM), i.e. ~1 s for the ~30 GiB working set of a 1 GiB build — ten times
// the build itself.  CoffeeDB rebuilds a fresh index object on every `build` (database.cpp:170-281) while
// the old one keeps serving, so freed blocks are kept and handed to the next build instead of going
// back to the driver; cdb_release_cached_memory() returns them.  288 GB of HBM makes this cheap.
// Every handle works on its own non-blocking stream and a new index is built while the old one serves
// queries (database.cpp:276-280), so a freed block may still be in use by queued kernels of the releasing
// stream: it carries an event recorded on that stream and is given to a different stream only after the
// event completed (same stream: immediately).
class DevPool {
public:
    static DevPool& get() {
        static DevPool p;
        return p;
    }
    void* alloc(size_t bytes, int device, size_t& actual) {
        const size_t gran = bytes >= (8u << 20) ? (2u << 20) : 256;
        const size_t need = (bytes + gran - 1) / gran * gran;
        Block busy{};
        bool have_busy = false;
        {
            std::lock_guard<std::mutex> g(mu_);
            int best = -1, best_busy = -1;
            for (int i = 0; i < (int)free_.size(); ++i) {
                Block& b = free_[i];
                if (b.device != device || b.bytes < need) continue;
                const size_t slack = need >= (64u << 20) ? need / 4 : need + (1u << 20);
                if (b.bytes > need + slack) continue;
                bool ready = !b.ev || (b.stream == tls_stream && tls_stream != nullptr);
                if (!ready && hipEventQuery(b.ev) == hipSuccess) {
                    put_event(b.device, b.ev);
                    b.ev = nullptr;
                    ready = true;
                } else if (!ready) {
                    (void)hipGetLastError();  // hipErrorNotReady is not an error
                }
                if (ready) {
                    // (blocks premap() put aside for the next generation go last: a rebuild of the SAME handle never needs them)
                    if (best < 0 || (b.spare != free_[best].spare ? !b.spare : b.bytes < free_[best].bytes)) best = i;
                } else if (best_busy < 0 || b.bytes < free_[best_busy].bytes) {
                    best_busy = i;
                }
            }
            if (best >= 0) {
                Block b = free_[best];
                free_.erase(free_.begin() + best);
                cached_ -= b.bytes;
                if (b.spare) spare_ -= std::min(spare_, b.bytes);
                if (b.ev) put_event(b.device, b.ev);
                actual = b.bytes;
                account(actual);
                return b.p;
            }
            if (best_busy >= 0) {  // waiting for the other stream beats a fresh hipMalloc ...
                busy = free_[best_busy];
                free_.erase(free_.begin() + best_busy);
                cached_ -= busy.bytes;
                have_busy = true;
            }
        }
        if (have_busy) {
            // ... but NOT under the pool's lock: every allocation and release of every handle — the queries of
            // the index still serving included — would stall behind a multi-second build on the other stream
            if (hipEventSynchronize(busy.ev) != hipSuccess) (void)hipGetLastError();
            {
                std::lock_guard<std::mutex> g(mu_);
                put_event(busy.device, busy.ev);
            }
            actual = busy.bytes;
            {
                std::lock_guard<std::mutex> g(mu_);
                account(actual);
            }
            return busy.p;
        }
        void* p = nullptr;
        if (need >= (16u << 20)) big_mallocs_.fetch_add(1, std::memory_order_relaxed);
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) {  // give cached blocks back to the driver and retry once
            (void)hipGetLastError();
            trim();
            e = hipMalloc(&p, need);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();  // (callers may catch this and carry on: leave no sticky error behind)
            throw Error(std::string("HIP error in hipMalloc: ") + hipGetErrorString(e));
        }
        actual = need;
        {
            std::lock_guard<std::mutex> g(mu_);
            account(actual);
        }
        return p;
    }
    // true when `bytes` can be had WITHOUT giving cached blocks back to the driver: a cached block of a fitting size exists, or
    // the driver still has that much untouched.  (A failed hipMalloc makes alloc() trim the whole cache and ask again — seconds
    // at tens of GiB, the driver scrubs what it hands out; optional scratch is better skipped than paid for like that.)
    bool can_serve(size_t bytes, int device) {
        const size_t gran = bytes >= (8u << 20) ? (2u << 20) : 256;
        const size_t need = (bytes + gran - 1) / gran * gran;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (const Block& b : free_) {
                if (b.device != device || b.bytes < need) continue;
                const size_t slack = need >= (64u << 20) ? need / 4 : need + (1u << 20);
                if (b.bytes <= need + slack) return true;
            }
        }
        size_t fre = 0, tot = 0;
        if (hipMemGetInfo(&fre, &tot) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return (double)fre >= (double)need * 1.05 + (double)(256u << 20);
    }
    // Maps what the NEXT generation will miss.  `sizes` = the blocks a published index keeps; a rebuild beside it (database.cpp:
    // 276-280: the new index is built while the old one serves) asks for them again while they are held, so the cache — which
    // holds the build's scratch — comes up short by that much, and a fresh hipMalloc of tens of GB costs more than the build
    // (0.6-1 s for 30 GB on a fresh box).  Twins of those blocks are allocated HERE, off the caller's path, and go straight into
    // the cache marked `spare` (no accounting: nothing is handed out; alloc() takes spare blocks last).  Only when this index is
    // the one generation alive (in steady state the old generation's blocks return to the cache and ARE the spare generation),
    // only while no earlier spare set is still unused, and never at the price of the device's last free memory.
    // Returns the bytes newly mapped.
    size_t premap(std::vector<size_t> sizes, int device, bool unconditional = false) {
        size_t total = 0;
        for (size_t& b : sizes) {
            const size_t gran = b >= (8u << 20) ? (2u << 20) : 256;
            b = (b + gran - 1) / gran * gran;
            total += b;
        }
        if (!unconditional) {
            std::lock_guard<std::mutex> g(mu_);
            if (in_use_ > total + total / 4) return 0;     // another generation (or other columns' indexes) is alive
            if (spare_ * 10 >= total * 9) return 0;        // the last spare set has not been used
        }
        std::sort(sizes.begin(), sizes.end(), [](size_t a, size_t b) { return a > b; });
        size_t mapped = 0;
        for (size_t need : sizes) {
            size_t fre = 0, tot = 0;
            if (hipMemGetInfo(&fre, &tot) != hipSuccess || (double)fre < (double)need * 1.1 + (double)(2ull << 30)) {
                (void)hipGetLastError();
                break;
            }
            void* p = nullptr;
            if (hipMalloc(&p, need) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            std::lock_guard<std::mutex> g(mu_);
            if (cached_ + need <= limit_) {
                free_.push_back(Block{p, need, device, nullptr, nullptr, true});
                cached_ += need;
                spare_ += need;
                mapped += need;
            } else {
                (void)hipFree(p);
                break;
            }
        }
        return mapped;
    }
    // bytes handed out right now / the most ever handed out since reset_peak() (what an index and its build really hold
    // in HBM: cdb_memory_stats; cached blocks are not counted, the caller's own buffers — a resident text — neither)
    void stats(size_t& in_use, size_t& peak, size_t& cached) {
        std::lock_guard<std::mutex> g(mu_);
        in_use = in_use_;
        peak = peak_;
        cached = cached_;
    }
    void reset_peak() {
        std::lock_guard<std::mutex> g(mu_);
        peak_ = in_use_;
    }
    void free(void* p, size_t bytes, int device) {
        {
            std::lock_guard<std::mutex> g(mu_);
            in_use_ -= std::min(in_use_, bytes);
        }
        // work queued on the releasing thread's stream may still use the block
        hipStream_t st = tls_stream;
        hipEvent_t ev = nullptr;
        if (st) {
            std::lock_guard<std::mutex> g(mu_);
            ev = get_event(device);
        }
        if (st && (!ev || hipEventRecord(ev, st) != hipSuccess)) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(st);
            if (ev) {
                std::lock_guard<std::mutex> g(mu_);
                put_event(device, ev);
            }
            ev = nullptr;
            st = nullptr;
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            if (cached_ + bytes <= limit_) {
                free_.push_back(Block{p, bytes, device, st, ev, false});
                cached_ += bytes;
                return;
            }
            if (ev) put_event(device, ev);
        }
        // the cache is full: hand the block back to the driver (hipFree waits for the device)
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        (void)hipFree(p);
        if (cur != device) (void)hipSetDevice(cur);
    }
    // A stream is about to be destroyed (its owner has synchronised it): blocks it released are plainly free now, and
    // their events must not be queried any more — an event keeps a pointer to the stream it was recorded on, and
    // hipEventQuery on one whose stream is gone reads freed memory (seen as a spurious "stream is capturing" error after
    // a few hundred handle create / destroy cycles).
    void retire_stream(hipStream_t s) {
        if (!s) return;
        std::vector<hipEvent_t> dead;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (Block& b : free_)
                if (b.stream == s) {
                    if (b.ev) dead.push_back(b.ev);
                    b.ev = nullptr;
                    b.stream = nullptr;
                }
        }
        for (hipEvent_t e : dead) (void)hipEventDestroy(e);
    }
    void set_limit(size_t bytes) {
        {
            std::lock_guard<std::mutex> g(mu_);
            limit_ = bytes;
            if (cached_ <= limit_) return;
        }
        trim();
    }
    void trim() {
        std::vector<Block> blocks;
        {
            std::lock_guard<std::mutex> g(mu_);
            blocks.swap(free_);
            cached_ = 0;
            spare_ = 0;
        }
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (auto& b : blocks) {
            (void)hipSetDevice(b.device);
            (void)hipFree(b.p);  // (waits for the device: pending work on the block has finished afterwards)
            if (b.ev) {
                std::lock_guard<std::mutex> g(mu_);
                put_event(b.device, b.ev);
            }
        }
        (void)hipSetDevice(cur);
    }
    size_t cached_bytes() {
        std::lock_guard<std::mutex> g(mu_);
        return cached_;
    }
    // requests of 16 MiB and more that the cache could not serve and went to hipMalloc (what cdb_reserve's spare generation and
    // premap() exist to avoid on a caller's path; stat "pool_big_mallocs")
    uint64_t big_mallocs() const { return big_mallocs_.load(std::memory_order_relaxed); }

private:
    struct Block { void* p; size_t bytes; int device; hipStream_t stream; hipEvent_t ev; bool spare; };
    // (both called with mu_ held.  Events are not recycled: one that was recorded on a stream which has been destroyed
    //  since still points at it)
    hipEvent_t get_event(int) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return e;
    }
    void put_event(int, hipEvent_t e) { (void)hipEventDestroy(e); }
    void account(size_t bytes) {  // (mu_ held)
        in_use_ += bytes;
        if (in_use_ > peak_) peak_ = in_use_;
    }
    std::mutex mu_;
    std::vector<Block> free_;
    size_t cached_ = 0;
    size_t in_use_ = 0, peak_ = 0;
    size_t spare_ = 0;  // bytes of cached blocks premap() put aside and nobody has taken yet
    std::atomic<uint64_t> big_mallocs_{0};
    size_t limit_ = ~(size_t)0;  // bytes kept for reuse (cdb_set_cache_limit); unlimited by default: re-allocating
                                 // the working set of a multi-GiB build costs more than the build itself
};

// Owning device allocation, served by DevPool.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int device = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), device(o.device) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; device = o.device; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        CDB_HIP(hipGetDevice(&device));
        p = DevPool::get().alloc(n, device, bytes);
    }
    void ensure(size_t n) { if (n > bytes) alloc(n); }
    void release() {
        if (p) DevPool::get().free(p, bytes, device);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Result arrays handed to the caller.  A fresh malloc() block is pageable and untouched: a multi-GB
// device-to-host copy into it runs at a few GB/s (page faults + the driver's staging copies).  Large results
// therefore come from a process-wide cache of PINNED blocks (hipHostMalloc; page-locking is paid once per
// block, later batches reuse it); small ones stay plain malloc.  host_free() tells the two apart.
class HostPool {
public:
    static HostPool& get() {
        static HostPool p;
        return p;
    }
    static constexpr size_t kMinPinned = 1u << 20;
    void* alloc(size_t bytes) {
        const size_t need = (bytes + 4095) / 4096 * 4096;
        {
            std::lock_guard<std::mutex> g(mu_);
            int best = -1;
            for (int i = 0; i < (int)free_.size(); ++i) {
                if (free_[i].bytes < need || free_[i].bytes > 2 * need + (8u << 20)) continue;
                if (best < 0 || free_[i].bytes < free_[best].bytes) best = i;
            }
            if (best >= 0) {
                Block b = free_[best];
                free_.erase(free_.begin() + best);
                cached_ -= b.bytes;
                live_[b.p] = b.bytes;
                return b.p;
            }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, need, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            trim();
            if (hipHostMalloc(&p, need, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
        }
        std::lock_guard<std::mutex> g(mu_);
        live_[p] = need;
        return p;
    }
    bool release(void* p) {  // false: not one of ours
        size_t bytes;
        {
            std::lock_guard<std::mutex> g(mu_);
            auto it = live_.find(p);
            if (it == live_.end()) return false;
            bytes = it->second;
            live_.erase(it);
            if (cached_ + bytes <= limit_) {
                free_.push_back(Block{p, bytes});
                cached_ += bytes;
                return true;
            }
        }
        (void)hipHostFree(p);
        return true;
    }
    void trim() {
        std::vector<Block> blocks;
        {
            std::lock_guard<std::mutex> g(mu_);
            blocks.swap(free_);
            cached_ = 0;
        }
        for (auto& b : blocks) (void)hipHostFree(b.p);
    }
    size_t cached_bytes() {
        std::lock_guard<std::mutex> g(mu_);
        return cached_;
    }

private:
    struct Block { void* p; size_t bytes; };
    std::mutex mu_;
    std::vector<Block> free_;
    std::map<void*, size_t> live_;
    size_t cached_ = 0;
    size_t limit_ = 8ull << 30;  // pinned bytes kept for the next batch
};

inline void* host_alloc(size_t bytes, bool zero = false) {
    if (bytes == 0) bytes = 8;
    void* p = bytes >= HostPool::kMinPinned ? HostPool::get().alloc(bytes) : nullptr;
    if (!p) p = std::malloc(bytes);
    if (!p) throw std::bad_alloc();
    if (zero) std::memset(p, 0, bytes);
    return p;
}
inline void host_free(void* p) {
    if (p && !HostPool::get().release(p)) std::free(p);
}

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
