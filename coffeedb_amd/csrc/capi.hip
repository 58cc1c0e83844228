// capi.hip — the C ABI of libcoffeedb_gpu.so (include/coffeedb_gpu.h).  Host logic only: staging of
// cdb_add, the reference's width rule, uploads/downloads and error translation.  All device work is in
// sa_build.hip / query.hip / radix_sort.h.
#include <dirent.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "../../include/coffeedb_gpu.h"
#include "index_impl.h"

using namespace cdb;

namespace {

double wall_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// queries run concurrently on one handle (database.cpp:388), so the last-error string has its own lock
void set_error(Index& ix, const char* msg) {
    std::lock_guard<std::mutex> g(ix.err_mu);
    ix.err = msg;
}

template <typename F>
int guarded(cdb_index* h, F&& f) {
    ForegroundCall fg;  // (the order proof of any handle yields to calls in flight: common.h)
    try {
        f();
        return CDB_OK;
    } catch (const Error& e) {
        set_error(h->ix, e.what());
        const bool dev = std::strncmp(e.what(), "HIP error", 9) == 0;
        const bool internal = std::strstr(e.what(), "internal") != nullptr;
        return dev ? CDB_E_DEVICE : (internal ? CDB_E_INTERNAL : CDB_E_INVALID);
    } catch (const std::bad_alloc&) {
        set_error(h->ix, "out of host memory");
        return CDB_E_DEVICE;
    } catch (const std::exception& e) {
        set_error(h->ix, e.what());
        return CDB_E_INTERNAL;
    }
}

// bits / mask / size / width exactly as string_index::build does (reference src/index.cpp:182-208).  Computed
// into a value first: an index is only touched once everything about the new build is known to be valid.
struct Layout {
    uint64_t size = 0, mask = 1, bits = 1, ndocs = 0;
    int width = 4, off_bits = 1;
};
Layout layout_from(uint64_t ndocs, uint64_t size, uint64_t longest) {
    uint64_t mask1 = 1, mask2 = 1;
    while (mask1 < ndocs) mask1 = (mask1 << 1) + 1;
    while (mask2 < longest) mask2 = (mask2 << 1) + 1;
    const int bits1 = __builtin_popcountll(mask1), bits2 = __builtin_popcountll(mask2);
    if (bits1 + bits2 > 64) throw Error("The amount of data exceeds the maximum range that CoffeeDB can handle");
    if (bits1 > 32) throw Error("The number of objects exceeds the maximum range that CoffeeDB can handle");
    Layout L;
    L.size = size;
    L.mask = mask1;
    L.bits = (uint64_t)bits1;
    L.width = bits1 + bits2 <= 32 ? 4 : 8;
    L.off_bits = bits2;
    L.ndocs = ndocs;
    return L;
}
Layout layout_of(const std::vector<uint64_t>& doc_start, uint64_t ndocs) {
    uint64_t size = 0, longest = 0;
    for (uint64_t d = 0; d < ndocs; ++d) {
        if (doc_start[d + 1] < doc_start[d]) throw Error("doc_start must be non-decreasing");
        const uint64_t len = doc_start[d + 1] - doc_start[d];
        size += len;
        longest = std::max(longest, len);
    }
    return layout_from(ndocs, size, longest);
}
void commit_layout(Index& ix, const Layout& L) {
    ix.size = L.size;
    ix.mask = L.mask;
    ix.bits = L.bits;
    ix.width = L.width;
    ix.off_bits = L.off_bits;
    ix.ndocs = L.ndocs;
}

// document tables into fresh device blocks (committed by the caller once everything else succeeded)
void upload_tables(Index& ix, const std::vector<uint64_t>& doc_start, const std::vector<int64_t>& ids, uint64_t ndocs,
                   DevBuf& d_start, DevBuf& d_ids) {
    hipStream_t s = ix.stream;
    d_start.alloc((ndocs + 1) * sizeof(uint64_t));
    CDB_HIP(hipMemcpyAsync(d_start.p, doc_start.data(), (ndocs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    d_ids.alloc(std::max<uint64_t>(ndocs, 1) * sizeof(int64_t));
    if (ndocs) CDB_HIP(hipMemcpyAsync(d_ids.p, ids.data(), ndocs * sizeof(int64_t), hipMemcpyHostToDevice, s));
}

// Host-to-device copy of a large PAGEABLE buffer (the staged column).  The runtime moves pageable memory through its own
// staging at ~11 GB/s (95 ms per GiB on MI355X); here four host threads copy 16 MiB chunks into pinned blocks of the
// host cache and queue the DMA behind each, so the page-touching memcpy of one chunk overlaps the DMA of the others.
// s2 (optional): a second stream for every other host thread's chunks — one stream keeps one SDMA engine busy at 55 GB/s, two
// reach the link's 57 GB/s (tools/experiments/h2d_bw.hip); s2 first waits for the work already queued on s (the destination block
// may come from the pool with work of s still pending on it), and every copy has completed when the function returns.
hipStream_t upload_stream(Index& ix) {  // the index's second stream (also the build's: sa_build.hip), created on first use
    if (!ix.aux_stream) CDB_HIP(hipStreamCreateWithFlags(&ix.aux_stream, hipStreamNonBlocking));
    return ix.aux_stream;
}
void upload_fork(hipStream_t s, hipStream_t s2) {
    if (!s2) return;
    hipEvent_t ev = nullptr;
    CDB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const hipError_t e1 = hipEventRecord(ev, s), e2 = e1 == hipSuccess ? hipStreamWaitEvent(s2, ev, 0) : e1;
    (void)hipEventDestroy(ev);
    CDB_HIP(e2);
}
void upload_pageable(void* dst, const char* src, size_t bytes, hipStream_t s, int device, hipStream_t s2 = nullptr) {
    constexpr size_t CHUNK = 16u << 20;
    // (a host thread fills pinned chunks at ~10 GB/s: four of them stay below the link's 57 GB/s, eight do not)
    static const int t_env = getenv("CDB_UPLOAD_THREADS") ? std::atoi(getenv("CDB_UPLOAD_THREADS")) : 0;
    const int T = std::max(2, std::min({t_env > 0 ? t_env : 8, (int)std::thread::hardware_concurrency() / 2, (int)(bytes / (2 * CHUNK))}));
    if (bytes < 4 * CHUNK) {
        if (bytes) CDB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        return;
    }
    upload_fork(s, s2);
    const size_t nchunks = (bytes + CHUNK - 1) / CHUNK;
    std::string failure;
    std::mutex fmu;
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            void* pin[2] = {nullptr, nullptr};
            hipEvent_t ev[2] = {nullptr, nullptr};
            try {
                CDB_HIP(hipSetDevice(device));
                for (int k = 0; k < 2; ++k) {
                    pin[k] = HostPool::get().alloc(CHUNK);
                    if (!pin[k]) throw Error("HIP error: no pinned staging memory");
                    CDB_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
                }
                int k = 0;
                bool used[2] = {false, false};
                for (size_t c = t; c < nchunks; c += T, k ^= 1) {
                    const size_t off = c * CHUNK, len = std::min(CHUNK, bytes - off);
                    if (used[k]) CDB_HIP(hipEventSynchronize(ev[k]));  // the DMA out of this block has finished
                    std::memcpy(pin[k], src + off, len);
                    hipStream_t cs = (s2 && (t & 1)) ? s2 : s;
                    CDB_HIP(hipMemcpyAsync(static_cast<char*>(dst) + off, pin[k], len, hipMemcpyHostToDevice, cs));
                    CDB_HIP(hipEventRecord(ev[k], cs));
                    used[k] = true;
                }
                for (int q = 0; q < 2; ++q)
                    if (used[q]) CDB_HIP(hipEventSynchronize(ev[q]));
            } catch (const std::exception& e) {
                (void)hipStreamSynchronize(s);  // a DMA out of the pinned blocks may still be in flight: it ends before they go back
                if (s2) (void)hipStreamSynchronize(s2);
                std::lock_guard<std::mutex> g(fmu);
                if (failure.empty()) failure = e.what();
            }
            for (int q = 0; q < 2; ++q) {
                if (ev[q]) (void)hipEventDestroy(ev[q]);
                if (pin[q]) (void)HostPool::get().release(pin[q]);
            }
        });
    for (auto& x : th) x.join();
    if (!failure.empty()) throw Error(failure);
}

// The same for a column that is NOT contiguous on the host: document d is the lens[d] bytes at ptrs[d] — string_index's own
// state (index.h:58: non-owning string_views into database.cpp's strings).  The gather into the pinned chunks IS the
// staging copy (there is no other one); doc_start = the running sum of lens.
void upload_views(void* dst, const char* const* ptrs, const uint64_t* doc_start, uint64_t ndocs, size_t bytes, hipStream_t s,
                  int device, hipStream_t s2 = nullptr) {
    constexpr size_t CHUNK = 16u << 20;
    // (the gather of scattered 1 KiB strings runs at ~5 GB/s per host thread: four threads would make it — not the PCIe link
    //  at ~52 GB/s — the bound of the shim's build(); up to twelve keep the link busy)
    const int hw = (int)std::thread::hardware_concurrency();
    static const int t_env = getenv("CDB_UPLOAD_THREADS") ? std::atoi(getenv("CDB_UPLOAD_THREADS")) : 0;  // (measurements)
    const int T = bytes >= 4 * CHUNK ? std::max(4, std::min({t_env > 0 ? t_env : 12, hw / 2, (int)(bytes / (2 * CHUNK))})) : 1;
    const size_t nchunks = (bytes + CHUNK - 1) / CHUNK;
    if (T > 1) upload_fork(s, s2);
    else s2 = nullptr;
    std::string failure;
    std::mutex fmu;
    std::vector<std::thread> th;
    auto work = [&](int t) {
        void* pin[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        try {
            CDB_HIP(hipSetDevice(device));
            for (int k = 0; k < 2; ++k) {
                pin[k] = HostPool::get().alloc(CHUNK);
                if (!pin[k]) throw Error("HIP error: no pinned staging memory");
                CDB_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
            }
            int k = 0;
            bool used[2] = {false, false};
            for (size_t c = t; c < nchunks; c += T, k ^= 1) {
                const size_t off = c * CHUNK, len = std::min(CHUNK, bytes - off);
                if (used[k]) CDB_HIP(hipEventSynchronize(ev[k]));  // the DMA out of this block has finished
                // documents overlapping [off, off + len): the first one is the last d with doc_start[d] <= off
                uint64_t d = std::upper_bound(doc_start, doc_start + ndocs + 1, (uint64_t)off) - doc_start - 1;
                size_t at = 0;
                while (at < len) {
                    const uint64_t ds = doc_start[d], de = doc_start[d + 1];
                    const uint64_t from = std::max<uint64_t>(ds, off + at), to = std::min<uint64_t>(de, off + len);
                    if (to > from) {
                        std::memcpy(static_cast<char*>(pin[k]) + at, ptrs[d] + (from - ds), to - from);
                        at += to - from;
                    }
                    ++d;
                }
                hipStream_t cs = (s2 && (t & 1)) ? s2 : s;
                CDB_HIP(hipMemcpyAsync(static_cast<char*>(dst) + off, pin[k], len, hipMemcpyHostToDevice, cs));
                CDB_HIP(hipEventRecord(ev[k], cs));
                used[k] = true;
            }
            for (int q = 0; q < 2; ++q)
                if (used[q]) CDB_HIP(hipEventSynchronize(ev[q]));
        } catch (const std::exception& e) {
            (void)hipStreamSynchronize(s);  // a DMA out of the pinned blocks may still be in flight: it ends before they go back
            if (s2) (void)hipStreamSynchronize(s2);
            std::lock_guard<std::mutex> g(fmu);
            if (failure.empty()) failure = e.what();
        }
        for (int q = 0; q < 2; ++q) {
            if (ev[q]) (void)hipEventDestroy(ev[q]);
            if (pin[q]) (void)HostPool::get().release(pin[q]);
        }
    };
    if (T == 1) {
        work(0);
    } else {
        for (int t = 0; t < T; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    if (!failure.empty()) throw Error(failure);
}

// back to "never built" (queries answer {}): a failed build or load must not leave new parameters over an old array
void reset_unbuilt(Index& ix) {
    proof_stop(ix);           // (the order proof reads the arrays released below)
    ix.proof.state.store(0);
    query_resident_stop(ix);  // (the resident query workgroup reads the arrays released below)
    (void)hipStreamSynchronize(ix.stream);
    ix.release_sa();
    ix.drop_keys();
    ix.d_pivots.release();
    ix.pivot_levels = 0;
    ix.width = 0;
    ix.size = 0;
    ix.q_spec_cap = 0;
}

// every entry point that touches the device: make the handle's device current and tell the block cache which
// stream this thread issues work on (common.h: blocks released meanwhile are tagged with it)
struct DeviceScope {
    StreamScope ss;
    explicit DeviceScope(Index& ix) : ss(ix.stream) { CDB_HIP(hipSetDevice(ix.device)); }
};

// resident build: longest document, order check and re-basing of the caller's device tables
__global__ __launch_bounds__(256) void layout_kernel(const uint64_t* __restrict__ src_start,
                                                     const int64_t* __restrict__ src_ids, uint64_t ndocs,
                                                     uint64_t* __restrict__ dst_start, int64_t* __restrict__ dst_ids,
                                                     unsigned long long* __restrict__ out /*[2]: max len, disorder*/) {
    __shared__ unsigned long long s_max[4];
    const uint64_t base = src_start[0];
    uint64_t mx = 0, bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x; d <= ndocs; d += stride) {
        const uint64_t a = src_start[d];
        dst_start[d] = a - base;
        if (d < ndocs) {
            const uint64_t b = src_start[d + 1];
            if (b < a) bad = 1;
            else mx = b - a > mx ? b - a : mx;
            dst_ids[d] = src_ids[d];
        }
    }
    for (int off = 32; off; off >>= 1) {
        const uint64_t o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
        bad |= __shfl_xor(bad, off);
    }
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
    if (bad && (threadIdx.x & 63) == 0) atomicExch(out + 1, 1ull);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) mx = s_max[w] > mx ? s_max[w] : mx;
        atomicMax(out, (unsigned long long)mx);
    }
}

}  // namespace

namespace cdb {
void ensure_host_tables(Index& ix) {
    if (ix.host_tables_valid) return;
    hipStream_t s = ix.stream;
    ix.ids.resize(ix.ndocs);
    ix.doc_start.resize(ix.ndocs + 1);
    if (ix.ndocs) CDB_HIP(hipMemcpyAsync(ix.ids.data(), ix.d_ids.p, ix.ndocs * 8, hipMemcpyDeviceToHost, s));
    CDB_HIP(hipMemcpyAsync(ix.doc_start.data(), ix.d_doc_start.p, (ix.ndocs + 1) * 8, hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
    ix.host_tables_valid = true;
}

// cdb_add* append to the HOST staging copy of the column.  After cdb_load / cdb_build_device / cdb_build_resident
// (and after a cdb_build, which frees its staging copy) the documents live on the device only: fetch tables and
// text back first, so that "load, add, rebuild" (a restart) works instead of corrupting the tables.
void ensure_host_staging(Index& ix) {
    if (ix.host_text_valid) return;
    std::lock_guard<std::mutex> g(ix.mu);
    StreamScope ss(ix.stream);
    CDB_HIP(hipSetDevice(ix.device));
    if (ix.width == 0 || !ix.d_text) {  // nothing on the device either: an empty column
        ix.ids.clear();
        ix.doc_start.assign(1, 0);
        ix.host_text.clear();
        ix.host_tables_valid = true;
        ix.host_text_valid = true;
        return;
    }
    ensure_host_tables(ix);
    ix.host_text.resize(ix.size);
    if (ix.size) CDB_HIP(hipMemcpyAsync(&ix.host_text[0], ix.d_text, ix.size, hipMemcpyDeviceToHost, ix.stream));
    CDB_HIP(hipStreamSynchronize(ix.stream));
    ix.host_text_valid = true;
}
}  // namespace cdb

namespace {
struct ReserveJob {
    std::mutex mu;
    std::thread th;
    ~ReserveJob() {
        if (th.joinable()) th.join();
    }
};
ReserveJob& reserve_job() {
    // (the pools exist BEFORE the job, so they are destroyed AFTER it: a reservation still building at process exit is joined
    //  while the block caches it allocates from are alive)
    (void)DevPool::get();
    (void)HostPool::get();
    static ReserveJob j;
    return j;
}
thread_local bool t_in_reserve = false;
void reserve_join() {
    if (t_in_reserve) return;
    ReserveJob& j = reserve_job();
    std::lock_guard<std::mutex> g(j.mu);
    if (j.th.joinable()) j.th.join();
}
// text[i] = table[hash(i) >> 52]: 4096 slots filled in proportion to the sample's byte histogram
__global__ __launch_bounds__(256) void reserve_fill_kernel(uint8_t* __restrict__ text, uint64_t n, const uint8_t* __restrict__ table,
                                                           uint64_t* __restrict__ doc_start, int64_t* __restrict__ ids, uint64_t ndocs) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        uint64_t x = i + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        text[i] = table[(x ^ (x >> 31)) >> 52];
    }
    for (uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x; d <= ndocs; d += stride) {
        doc_start[d] = d == ndocs ? n : (unsigned __int128)n * d / ndocs;
        if (d < ndocs) ids[d] = (int64_t)d;
    }
}
}  // namespace

extern "C" {

int cdb_create(cdb_index** out, int device) {
    if (!out) return CDB_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return CDB_E_DEVICE;
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return CDB_E_DEVICE;
    }
    if (device >= count) return CDB_E_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CDB_E_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return CDB_E_DEVICE;  // kernels exist for gfx950 only
    cdb_index* h = new (std::nothrow) cdb_index();
    if (!h) return CDB_E_DEVICE;
    h->ix.device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->ix.stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return CDB_E_DEVICE;
    }
    *out = h;
    return CDB_OK;
}

void cdb_destroy(cdb_index* h) {
    if (!h) return;
    (void)hipSetDevice(h->ix.device);
    hipStream_t s = h->ix.stream;
    proof_forget(h->ix);
    if (h->ix.proof.stream) (void)hipStreamDestroy(h->ix.proof.stream);
    if (h->ix.proof.d_out) (void)hipFree(h->ix.proof.d_out);
    query_resident_stop(h->ix);
    if (s) (void)hipStreamSynchronize(s);
    if (h->ix.res_stream) (void)hipStreamDestroy(h->ix.res_stream);
    if (h->ix.aux_stream) {
        (void)hipStreamSynchronize(h->ix.aux_stream);
        (void)hipStreamDestroy(h->ix.aux_stream);
    }
    for (hipEvent_t e : h->ix.aux_ev)
        if (e) (void)hipEventDestroy(e);
    if (h->ix.h_res) (void)hipHostFree(h->ix.h_res);
    if (h->ix.h_single) (void)hipHostFree(h->ix.h_single);
    h->ix.stream = nullptr;
    delete h;  // (device blocks go back to the cache untagged: the stream is idle)
    if (s) {
        DevPool::get().retire_stream(s);  // blocks released earlier under this stream: their events die with it
        (void)hipStreamDestroy(s);
    }
}

const char* cdb_last_error(const cdb_index* h) {
    if (!h) return "null handle";
    // a per-thread copy: another thread's failing call cannot pull the string away under the reader
    static thread_local std::string copy;
    {
        std::lock_guard<std::mutex> g(h->ix.err_mu);
        copy = h->ix.err;
    }
    return copy.c_str();
}

int cdb_add(cdb_index* h, int64_t id, const char* value, size_t len) {
    if (!h || (!value && len)) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        ensure_host_staging(ix);
        ix.host_text.append(value, len);
        try {
            ix.ids.push_back(id);
            ix.doc_start.push_back(ix.host_text.size());
        } catch (...) {  // keep the three staging arrays consistent
            ix.host_text.resize(ix.doc_start.back());
            ix.ids.resize(ix.doc_start.size() - 1);
            throw;
        }
    });
}

int cdb_add_bulk(cdb_index* h, const int64_t* ids, const char* blob, const uint64_t* doc_start, uint64_t ndocs) {
    if (!h || (ndocs && (!ids || !doc_start))) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        if (!ndocs) return;
        for (uint64_t d = 0; d < ndocs; ++d)
            if (doc_start[d + 1] < doc_start[d]) throw Error("doc_start must be non-decreasing");
        ensure_host_staging(ix);
        ix.ids.reserve(ix.ids.size() + ndocs);  // (allocations first: nothing below can fail half way)
        ix.doc_start.reserve(ix.doc_start.size() + ndocs);
        const uint64_t base = ix.host_text.size();
        ix.host_text.append(blob + doc_start[0], doc_start[ndocs] - doc_start[0]);
        for (uint64_t d = 0; d < ndocs; ++d) {
            ix.ids.push_back(ids[d]);
            ix.doc_start.push_back(base + doc_start[d + 1] - doc_start[0]);
        }
    });
}

// ---- f3: raw-record ingest (reference on-disk record, database.cpp:182-271 reader / :334-378 writer) ----
//   int64 id | int32 nfields | nfields x { int32 keylen | key | int8 type | value }
//   value: type 0 bool = 1 byte, 1 integer = 8 bytes, 2 double = 8 bytes, 3 string = int32 len | bytes
int cdb_raw_record_find_string(const void* record, size_t len, const char* key, int64_t* id, const char** value,
                               size_t* value_len) {
    if (!record || !key || !id || !value || !value_len) return -1;
    const unsigned char* p = static_cast<const unsigned char*>(record);
    const unsigned char* end = p + len;
    auto take = [&](void* dst, size_t n) -> bool {
        if ((size_t)(end - p) < n) return false;
        std::memcpy(dst, p, n);
        p += n;
        return true;
    };
    int32_t nfields = 0;
    if (!take(id, 8) || !take(&nfields, 4) || nfields <= 0) return -1;
    const size_t klen = std::strlen(key);
    int found = 0;
    for (int32_t f = 0; f < nfields; ++f) {
        int32_t kl = 0;
        if (!take(&kl, 4) || kl <= 0 || (size_t)(end - p) < (size_t)kl + 1) return -1;
        const bool match = (size_t)kl == klen && std::memcmp(p, key, klen) == 0;
        p += kl;
        const int8_t type = (int8_t)*p++;
        if (type == 0) {
            if (end - p < 1) return -1;
            p += 1;
        } else if (type == 1 || type == 2) {
            if (end - p < 8) return -1;
            p += 8;
        } else if (type == 3) {
            int32_t vl = 0;
            if (!take(&vl, 4) || vl < 0 || (size_t)(end - p) < (size_t)vl) return -1;
            if (match) {
                *value = reinterpret_cast<const char*>(p);
                *value_len = (size_t)vl;
                found = 1;
            }
            p += vl;
        } else {
            return -1;
        }
    }
    return found;
}

int cdb_add_raw_record(cdb_index* h, const char* key, const void* record, size_t len) {
    if (!h) return CDB_E_INVALID;
    int64_t id = 0;
    const char* v = nullptr;
    size_t vl = 0;
    const int r = cdb_raw_record_find_string(record, len, key, &id, &v, &vl);
    if (r < 0) {
        set_error(h->ix, "malformed raw record");
        return CDB_E_INVALID;
    }
    if (r == 0) return CDB_OK;  // the object has no string value under this key
    return cdb_add(h, id, v, vl);
}

// Bulk form of the raw-file ingest (database.cpp:170-275 walks storage_location/raw/ with a directory_iterator, parses
// every record into a map and hands string_index::add a view of each value): here every record file of `dir` is read
// once and the string stored under `key` goes straight into the concatenated staging column (text, doc_start, ids) —
// no per-object map, no per-value std::string.  Files are taken in ascending NAME order (the reference's
// directory order is unspecified; document order only shows in the entry encoding, never in a result).
int cdb_add_raw_dir(cdb_index* h, const char* dir, const char* key, uint64_t* records, uint64_t* added) {
    if (!h || !dir || !key) return CDB_E_INVALID;
    if (records) *records = 0;
    if (added) *added = 0;
    return guarded(h, [&] {
        Index& ix = h->ix;
        ensure_host_staging(ix);
        uint64_t nrec = 0, nadd = 0;
        cdb::read_raw_dir(dir, key, ix.ids, ix.doc_start, ix.host_text, nrec, nadd);
        if (records) *records = nrec;
        if (added) *added = nadd;
    });
}
}  // extern "C"

// every record file of `dir` (ascending name order) appended to a staged column; all or nothing
void cdb::read_raw_dir(const char* dir, const char* key, std::vector<int64_t>& ids, std::vector<uint64_t>& doc_start, std::string& text,
                       uint64_t& nrec, uint64_t& nadd) {
    {
        DIR* d = opendir(dir);
        if (!d) throw Error(std::string("Cannot open directory: ") + dir);
        std::vector<std::string> names;
        while (const dirent* e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            names.emplace_back(e->d_name);
        }
        closedir(d);
        std::sort(names.begin(), names.end());
        const size_t mark_ids = ids.size(), mark_text = text.size();
        std::vector<char> buf;
        nrec = nadd = 0;
        try {
            for (const std::string& name : names) {
                const std::string path = std::string(dir) + "/" + name;
                FILE* fp = std::fopen(path.c_str(), "rb");
                if (!fp) throw Error("Cannot open file: " + path);
                struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{fp};
                if (std::fseek(fp, 0, SEEK_END) != 0) throw Error("Cannot read file: " + path);
                const long len = std::ftell(fp);
                if (len < 0 || std::fseek(fp, 0, SEEK_SET) != 0) throw Error("Cannot read file: " + path);
                buf.resize((size_t)len);
                if (len && std::fread(buf.data(), 1, (size_t)len, fp) != (size_t)len) throw Error("Cannot read file: " + path);
                int64_t id = 0;
                const char* v = nullptr;
                size_t vl = 0;
                const int r = cdb_raw_record_find_string(buf.data(), buf.size(), key, &id, &v, &vl);
                if (r < 0) throw Error("malformed raw record: " + path);
                ++nrec;
                if (r == 0) continue;  // the object has no string value under this key
                text.append(v, vl);
                ids.push_back(id);
                doc_start.push_back(text.size());
                ++nadd;
            }
        } catch (...) {  // all or nothing: a half-read directory must not leave a partial column behind
            ids.resize(mark_ids);
            doc_start.resize(mark_ids + 1);
            text.resize(mark_text);
            throw;
        }
    }
}

extern "C" {

// ---- f4: persistence of a built index (the reference rebuilds every index at start, server.cpp:44) ----
namespace {
constexpr uint64_t SAVE_MAGIC = 0x3230584449424443ull;  // "CDBIDX02"
struct SaveHeader {
    uint64_t magic, size, ndocs, bits, mask, width, compat, sorted;
};
}  // namespace

int cdb_save(cdb_index* h, const char* path) {
    if (!h || !path) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        if (ix.width == 0) throw Error("index has not been built");
        ensure_host_tables(ix);
        FILE* fp = std::fopen(path, "wb");
        if (!fp) throw Error(std::string("Cannot open file: ") + path);
        struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{fp};
        const SaveHeader hd{SAVE_MAGIC, ix.size, ix.ndocs, ix.bits, ix.mask, (uint64_t)ix.width, ix.reference_compat ? 1ull : 0ull,
                            ix.sa_sorted ? 1ull : 0ull};
        bool ok = std::fwrite(&hd, sizeof(hd), 1, fp) == 1;
        ok = ok && (ix.ndocs == 0 || std::fwrite(ix.ids.data(), 8, ix.ndocs, fp) == ix.ndocs);
        ok = ok && std::fwrite(ix.doc_start.data(), 8, ix.ndocs + 1, fp) == ix.ndocs + 1;
        std::vector<char> buf(std::min<uint64_t>(std::max<uint64_t>(ix.size * (uint64_t)ix.width, 1), 256ull << 20));
        auto dump = [&](const void* dptr, uint64_t bytes) {
            for (uint64_t o = 0; o < bytes && ok; o += buf.size()) {
                const uint64_t c = std::min<uint64_t>(buf.size(), bytes - o);
                CDB_HIP(hipMemcpyAsync(buf.data(), static_cast<const char*>(dptr) + o, c, hipMemcpyDeviceToHost, ix.stream));
                CDB_HIP(hipStreamSynchronize(ix.stream));
                ok = std::fwrite(buf.data(), 1, c, fp) == c;
            }
        };
        dump(ix.d_text, ix.size);
        if (ix.sa_packed) {  // the file holds the reference's u64 entries whatever the storage: expanded chunk by chunk
            DevBuf chunk;
            const uint64_t per = std::max<uint64_t>(buf.size() / 8, 1);
            chunk.alloc(per * 8);
            for (uint64_t first = 0; first < ix.size && ok; first += per) {
                const uint64_t cnt = std::min<uint64_t>(per, ix.size - first);
                sa_expand(ix, first, cnt, chunk.as<uint64_t>());
                dump(chunk.p, cnt * 8);
            }
        } else {
            dump(ix.d_sa.p, ix.size * (uint64_t)ix.width);
        }
        if (!ok) throw Error(std::string("Cannot write file: ") + path);
    });
}

int cdb_load(cdb_index* h, const char* path) {
    reserve_join();
    if (!h || !path) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        FILE* fp = std::fopen(path, "rb");
        if (!fp) throw Error(std::string("Cannot open file: ") + path);
        struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{fp};
        // ---- everything is read and checked into locals; the handle changes only when the file proved consistent
        SaveHeader hd{};
        if (std::fread(&hd, sizeof(hd), 1, fp) != 1 || hd.magic != SAVE_MAGIC || (hd.width != 4 && hd.width != 8))
            throw Error(std::string("Not a saved index: ") + path);
        if (std::fseek(fp, 0, SEEK_END) != 0) throw Error(std::string("Cannot read file: ") + path);
        const long long fsize = std::ftell(fp);
        if (hd.ndocs >= (1ull << 33) || hd.size >= (1ull << 48) ||
            fsize < 0 || (unsigned long long)fsize != sizeof(hd) + 8 * hd.ndocs + 8 * (hd.ndocs + 1) + hd.size + hd.size * hd.width)
            throw Error(std::string("Truncated index file: ") + path);
        if (std::fseek(fp, (long)sizeof(hd), SEEK_SET) != 0) throw Error(std::string("Cannot read file: ") + path);
        std::vector<int64_t> ids(hd.ndocs);
        std::vector<uint64_t> doc_start(hd.ndocs + 1);
        bool ok = hd.ndocs == 0 || std::fread(ids.data(), 8, hd.ndocs, fp) == hd.ndocs;
        ok = ok && std::fread(doc_start.data(), 8, hd.ndocs + 1, fp) == hd.ndocs + 1;
        if (!ok) throw Error(std::string("Truncated index file: ") + path);
        if (doc_start[0] != 0 || doc_start[hd.ndocs] != hd.size) throw Error(std::string("Corrupt index file (document table): ") + path);
        const Layout L = layout_of(doc_start, hd.ndocs);  // (also: doc_start non-decreasing)
        if (L.size != hd.size || L.bits != hd.bits || L.mask != hd.mask || (uint64_t)L.width != hd.width)
            throw Error(std::string("Corrupt index file (entry layout): ") + path);
        DevBuf text, sa, sa_hi, d_start, d_ids;
        text.alloc(hd.size + TEXT_PAD);
        CDB_HIP(hipMemsetAsync((uint8_t*)text.p + hd.size, 0, TEXT_PAD, ix.stream));
        // 8-byte entries below 2^40 are stored packed (the storage a build of this column would leave): they are packed chunk by
        // chunk while the file is read, so the plain array never exists on the device and nothing can fail after the commit
        const bool pack = ix.pack_sa && hd.width == 8 && (int)L.bits + L.off_bits <= 40 && hd.size > 0;
        if (pack) {
            sa.alloc(hd.size * sizeof(uint32_t));
            sa_hi.alloc(hd.size);
        } else {
            sa.alloc(std::max<uint64_t>(hd.size * hd.width, 16));
        }
        std::vector<char> buf(std::min<uint64_t>(std::max<uint64_t>(hd.size * hd.width, 1), 256ull << 20));
        auto fill = [&](void* dptr, uint64_t bytes) {
            for (uint64_t o = 0; o < bytes; o += buf.size()) {
                const uint64_t c = std::min<uint64_t>(buf.size(), bytes - o);
                if (std::fread(buf.data(), 1, c, fp) != c) throw Error(std::string("Truncated index file: ") + path);
                CDB_HIP(hipMemcpyAsync(static_cast<char*>(dptr) + o, buf.data(), c, hipMemcpyHostToDevice, ix.stream));
                CDB_HIP(hipStreamSynchronize(ix.stream));
            }
        };
        fill(text.p, hd.size);
        upload_tables(ix, doc_start, ids, hd.ndocs, d_start, d_ids);
        // every entry must name a real (document, offset): queries decode entries without further checks
        const char* bad_sa = "Corrupt index file (suffix array): ";
        if (pack) {
            DevBuf chunk;
            const uint64_t per = buf.size() / 8;
            chunk.alloc(std::max<uint64_t>(per * 8, 16));
            for (uint64_t first = 0; first < hd.size; first += per) {
                const uint64_t cnt = std::min<uint64_t>(per, hd.size - first);
                if (std::fread(buf.data(), 8, cnt, fp) != cnt) throw Error(std::string("Truncated index file: ") + path);
                CDB_HIP(hipMemcpyAsync(chunk.p, buf.data(), cnt * 8, hipMemcpyHostToDevice, ix.stream));
                if (count_invalid_entries(ix.stream, chunk.p, 8, cnt, d_start.as<uint64_t>(), hd.ndocs, (int)L.bits, L.mask) != 0)
                    throw Error(std::string(bad_sa) + path);
                sa_pack_chunk(ix.stream, chunk.as<uint64_t>(), cnt, sa.as<uint32_t>(), sa_hi.as<uint8_t>(), first);
                CDB_HIP(hipStreamSynchronize(ix.stream));  // (buf and chunk are reused)
            }
        } else {
            fill(sa.p, hd.size * hd.width);
            if (count_invalid_entries(ix.stream, sa.p, (int)hd.width, hd.size, d_start.as<uint64_t>(), hd.ndocs, (int)L.bits, L.mask) != 0)
                throw Error(std::string(bad_sa) + path);
        }
        CDB_HIP(hipStreamSynchronize(ix.stream));
        // ---- commit
        reset_unbuilt(ix);
        commit_layout(ix, L);
        ix.ids.swap(ids);
        ix.doc_start.swap(doc_start);
        ix.host_tables_valid = true;
        ix.host_text.clear();
        ix.host_text.shrink_to_fit();
        ix.host_text_valid = false;  // the text lives on the device (cdb_add* fetch it back)
        ix.reference_compat = hd.compat != 0;
        ix.sa_sorted = hd.sorted != 0;  // a reference-compat ordering keeps the reference's exact probe sequence
        ix.d_text_owned = std::move(text);
        ix.d_text = ix.d_text_owned.as<uint8_t>();
        ix.text_padded = true;
        ix.d_sa = std::move(sa);
        if (pack) {
            ix.d_sa_hi = std::move(sa_hi);
            ix.sa_packed = true;
        }
        ix.d_doc_start = std::move(d_start);
        ix.d_ids = std::move(d_ids);
        // a file's entries were checked one by one (each names a real suffix), their ORDER was not: the proof behind a build runs
        // behind a load as well (damage -> the array is rebuilt from the loaded text)
        if (ix.self_check >= 3 || ix.premap_generation) {
            ix.proof.of_loaded_file = true;
            proof_start(ix);
        }
    });
}

// ---- cdb_reserve: the working set of the first build, mapped BEFORE the build is asked for --------------------------------
// The first build of a fresh process pays for VRAM the driver maps (and, for pages another process released, scrubs) on first
// use: 0.3-2.6 s for a 4 GiB column by the state of the box against 0.13 s warm.  server.cpp:43-44 loads the data
// from disk and only then builds, so that time can hide behind the ingest: cdb_reserve starts a helper thread that builds a
// throw-away index over SYNTHETIC text of the announced size (independent bytes drawn from the sample's byte histogram, so the
// alphabet — and with it record widths and block sizes — resembles the real column) and destroys it again.  Its blocks stay in
// the process-wide block cache (DevPool), which hands them to the real build.  Best effort: any failure just leaves the cache
// as it was.  Every cdb_build* / cdb_load first waits for a reservation still in flight.
int cdb_reserve(int device, uint64_t text_bytes, uint64_t ndocs, const char* sample, size_t sample_len) {
    if (text_bytes == 0 || text_bytes >= (1ull << 40)) return CDB_E_INVALID;
    // 4 % and 64 MiB over the announced size: a cached block serves a request only when it is at least as large (DevPool keeps up
    // to a quarter of slack), so the synthetic column must not come out smaller than the real one
    text_bytes = (text_bytes + text_bytes / 25 + (64ull << 20) + 15) & ~15ull;
    if (ndocs == 0) ndocs = std::max<uint64_t>(1, text_bytes / 1024);
    else ndocs += ndocs / 25;
    if (ndocs > text_bytes) ndocs = text_bytes;
    std::vector<uint8_t> table(4096);
    {
        uint64_t hist[256] = {0};
        if (sample && sample_len) {
            for (size_t i = 0; i < sample_len; ++i) hist[(uint8_t)sample[i]]++;
        } else {
            for (int b = 0x20; b <= 0x7E; ++b) hist[b] = 1;  // (no sample: printable ASCII)
        }
        uint64_t total = 0;
        for (uint64_t v : hist) total += v;
        size_t at = 0;
        uint64_t run = 0;
        for (int b = 0; b < 256; ++b) {
            run += hist[b];
            const size_t end = (size_t)((unsigned __int128)run * 4096 / total);
            const size_t stop = hist[b] ? std::max(end, std::min<size_t>(at + 1, 4096)) : end;  // (every byte of the sample keeps a slot)
            while (at < stop && at < 4096) table[at++] = (uint8_t)b;
        }
        for (; at < 4096; ++at) table[at] = at ? table[at - 1] : (uint8_t)0x20;
    }
    // device < 0 = the CALLER's current device (the helper thread's own current device is always 0)
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return CDB_E_DEVICE;
    ReserveJob& j = reserve_job();
    std::lock_guard<std::mutex> g(j.mu);  // join and assignment under ONE lock: two concurrent calls cannot assign to a joinable thread
    if (j.th.joinable()) j.th.join();
    try {
        j.th = std::thread([device, text_bytes, ndocs, table] {
            t_in_reserve = true;
            const double t_start = wall_ms();
            cdb_index* h = nullptr;
            std::vector<size_t> twin_sizes;
            int twin_dev = 0;
            try {
                if (cdb_create(&h, device) != CDB_OK) return;
                Index& ix = h->ix;
                twin_dev = ix.device;
                ix.self_check = 1;  // (a throw-away array: no order proof behind it, no second generation)
                ix.premap_generation = false;
                CDB_HIP(hipSetDevice(ix.device));
                DevBuf text, d_table, d_start, d_ids;
                text.alloc(text_bytes + TEXT_PAD);
                d_table.alloc(4096);
                d_start.alloc((ndocs + 1) * sizeof(uint64_t));
                d_ids.alloc(ndocs * sizeof(int64_t));
                CDB_HIP(hipMemcpyAsync(d_table.p, table.data(), 4096, hipMemcpyHostToDevice, ix.stream));
                CDB_HIP(hipMemsetAsync((uint8_t*)text.p + text_bytes, 0, TEXT_PAD, ix.stream));
                hipLaunchKernelGGL(reserve_fill_kernel, dim3(4096), dim3(256), 0, ix.stream, text.as<uint8_t>(), text_bytes,
                                   (const uint8_t*)d_table.as<uint8_t>(), d_start.as<uint64_t>(), d_ids.as<int64_t>(), ndocs);
                CDB_HIP(hipStreamSynchronize(ix.stream));
                {   // the pinned staging chunks of the first upload (upload_views / upload_pageable: two 16 MiB blocks per copy thread)
                    std::vector<void*> pins;
                    for (int k = 0; k < 24; ++k)
                        if (void* q = HostPool::get().alloc(16u << 20)) pins.push_back(q);
                    for (void* q : pins) (void)HostPool::get().release(q);
                }
                const int rc = cdb_build_resident(h, text.p, d_start.as<uint64_t>(), d_ids.as<int64_t>(), ndocs);
                // ... and the SECOND generation: the first `build` operation after start-up constructs a new index while this one
                // serves (database.cpp:276-280) and asks for the arrays an index keeps once more — twins of those blocks are mapped
                // now, while nothing is being served (marked spare: the first build does not take them; DevPool::premap)
                if (rc == CDB_OK) {
                    twin_sizes = retained_block_sizes(ix);
                    twin_sizes.push_back(text.bytes);  // (the real index owns its text: cdb_build / cdb_build_view(s))
                }
                if (getenv("CDB_BUILD_TRACE"))
                    std::fprintf(stderr, "[reserve] %llu bytes, %llu documents: throw-away build rc %d, %.1f ms in all\n", (unsigned long long)text_bytes,
                                 (unsigned long long)ndocs, rc, wall_ms() - t_start);
            } catch (...) {
            }
            if (h) cdb_destroy(h);  // (its arrays and the build's scratch go back to the block cache: that is the reservation)
            try {
                const double t2 = wall_ms();
                const size_t got = twin_sizes.empty() ? 0 : DevPool::get().premap(twin_sizes, twin_dev, true);
                if (getenv("CDB_BUILD_TRACE"))
                    std::fprintf(stderr, "[reserve] second generation: %.1f GB mapped in %.1f ms\n", (double)got / 1e9, wall_ms() - t2);
            } catch (...) {
            }
        });
    } catch (...) {
        return CDB_E_DEVICE;
    }
    return CDB_OK;
}

void cdb_reserve_wait(void) { reserve_join(); }

int cdb_build(cdb_index* h) {
    if (!h) return CDB_E_INVALID;
    reserve_join();
    return guarded(h, [&] {
        Index& ix = h->ix;
        ensure_host_staging(ix);  // (a rebuild after cdb_load / a device build: the staging copy is fetched back)
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        const Layout L = layout_of(ix.doc_start, ix.ids.size());  // throws the reference's capacity errors: nothing changed yet
        const uint64_t n = L.size;
        bool committed = false;  // a failure before the commit point (allocation, upload) leaves the previous index serving
        try {
            DevBuf text, d_start, d_ids;
            text.alloc(n + TEXT_PAD);
            CDB_HIP(hipMemsetAsync((uint8_t*)text.p + n, 0, TEXT_PAD, ix.stream));
            const double tu = wall_ms();
            upload_pageable(text.p, ix.host_text.data(), n, ix.stream, ix.device, upload_stream(ix));
            upload_tables(ix, ix.doc_start, ix.ids, L.ndocs, d_start, d_ids);
            committed = true;
            reset_unbuilt(ix);  // (waits for the stream: the old arrays are idle; ix.mu keeps queries out)
            ix.host_upload_ms = wall_ms() - tu;
            commit_layout(ix, L);
            ix.d_text_owned = std::move(text);
            ix.d_text = ix.d_text_owned.as<uint8_t>();
            ix.text_padded = true;
            ix.d_doc_start = std::move(d_start);
            ix.d_ids = std::move(d_ids);
            build_suffix_array(ix);
        } catch (...) {
            if (committed) reset_unbuilt(ix);
            else (void)hipStreamSynchronize(ix.stream);
            throw;
        }
        // the staging copy has done its job (database.cpp builds a fresh index object per build and never adds to a
        // built one); cdb_add* fetch the column back from the device if they are called again
        if (ix.host_text.size() >= (1u << 20)) {
            // (returning a GiB of pages to the kernel takes ~65 ms: not the caller's time — a helper thread lets go of it)
            const double tf = wall_ms();
            std::string gone;
            gone.swap(ix.host_text);
            ix.host_text_valid = false;
            try {
                std::thread([g2 = std::move(gone)]() mutable { std::string().swap(g2); }).detach();
            } catch (...) {  // no thread to be had: release it here (the moved-from lambda state already did, or `gone` does)
            }
            ix.host_free_ms = wall_ms() - tf;
        }
    });
}

int cdb_build_device(cdb_index* h, const void* d_text, const uint64_t* doc_start, const int64_t* ids, uint64_t ndocs) {
    reserve_join();
    if (!h || (ndocs && (!doc_start || !ids))) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        // ---- validate first
        if (((uintptr_t)d_text & 15u) != 0) throw Error("device text must be 16-byte aligned");
        const uint64_t first = ndocs ? doc_start[0] : 0;
        if (first & 15u) throw Error("first document must start 16-byte aligned");
        std::vector<int64_t> hid(ids, ids + ndocs);
        std::vector<uint64_t> hstart(ndocs + 1);
        hstart[0] = 0;
        for (uint64_t d = 0; d < ndocs; ++d) {
            if (doc_start[d + 1] < doc_start[d]) throw Error("doc_start must be non-decreasing");
            hstart[d + 1] = doc_start[d + 1] - first;
        }
        const Layout L = layout_of(hstart, ndocs);
        bool committed = false;
        try {
            DevBuf d_start, d_ids;
            upload_tables(ix, hstart, hid, ndocs, d_start, d_ids);
            committed = true;
            reset_unbuilt(ix);
            commit_layout(ix, L);
            ix.ids.swap(hid);
            ix.doc_start.swap(hstart);
            ix.host_tables_valid = true;
            std::string().swap(ix.host_text);
            ix.host_text_valid = false;
            ix.d_text_owned.release();
            ix.d_text = static_cast<const uint8_t*>(d_text) + first;
            ix.text_padded = false;
            ix.d_doc_start = std::move(d_start);
            ix.d_ids = std::move(d_ids);
            build_suffix_array(ix);
        } catch (...) {
            if (committed) reset_unbuilt(ix);
            else (void)hipStreamSynchronize(ix.stream);
            throw;
        }
    });
}

/* cdb_build_view: build straight from the caller's host column — what string_index really holds (index.h:58:
 * non-owning string_views into database.cpp's map, database.cpp:262-264) — without the staging copy of cdb_add_bulk
 * (a memcpy into fresh pages: 180 ms per GiB, 3x the build it feeds).  The column goes to the device through the
 * chunked pinned upload; afterwards the handle owns device copies only (cdb_add* fetch them back when needed). */
int cdb_build_view(cdb_index* h, const int64_t* ids, const char* blob, const uint64_t* doc_start, uint64_t ndocs) {
    reserve_join();
    if (!h || (ndocs && (!ids || !doc_start || (!blob && doc_start[ndocs] > doc_start[0])))) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        const uint64_t first = ndocs ? doc_start[0] : 0;
        // The text starts travelling FIRST, on a helper thread, into a block nothing refers to yet; the document tables are
        // copied and validated meanwhile (1-2 ms per million documents on one core — it used to sit in front of the upload).
        // A column that fails validation is simply dropped after the copy.
        const uint64_t n_claimed = ndocs ? doc_start[ndocs] - first : 0;
        if (ndocs && (doc_start[ndocs] < first || n_claimed >= (1ull << 48))) throw Error("doc_start must be non-decreasing");
        // O(1) checks BEFORE anything is allocated or uploaded: the longest document has at least n / ndocs bytes, so a
        // column the width rule refuses for that length gets the reference's capacity error (index.cpp:195-200), not a hipMalloc
        // error after the block cache was trimmed; and a text that does not fit the device is refused without trimming it
        if (ndocs) (void)layout_from(ndocs, n_claimed, (n_claimed + ndocs - 1) / ndocs);
        {
            size_t fre = 0, tot = 0;
            if (hipMemGetInfo(&fre, &tot) == hipSuccess) {
                if ((double)n_claimed + (double)TEXT_PAD > (double)fre + (double)cdb_cached_memory_bytes())
                    throw Error("HIP error: the column does not fit the device memory that is free");
            } else {
                (void)hipGetLastError();
            }
        }
        DevBuf text;
        text.alloc(n_claimed + TEXT_PAD);
        CDB_HIP(hipMemsetAsync((uint8_t*)text.p + n_claimed, 0, TEXT_PAD, ix.stream));
        const double tu = wall_ms();
        std::string uerr;
        hipStream_t s2 = upload_stream(ix);
        std::thread up([&] {
            try {
                CDB_HIP(hipSetDevice(ix.device));
                if (n_claimed) upload_pageable(text.p, blob + first, n_claimed, ix.stream, ix.device, s2);
            } catch (const std::exception& e) {
                uerr = e.what();
            }
        });
        struct Joiner {
            std::thread& t;
            ~Joiner() { if (t.joinable()) t.join(); }
        } joiner{up};
        std::vector<int64_t> hid(ids, ids + ndocs);
        std::vector<uint64_t> hstart(ndocs + 1);
        hstart[0] = 0;
        for (uint64_t d = 0; d < ndocs; ++d) {
            if (doc_start[d + 1] < doc_start[d]) throw Error("doc_start must be non-decreasing");
            hstart[d + 1] = doc_start[d + 1] - first;
        }
        const Layout L = layout_of(hstart, ndocs);  // (throws the reference's capacity errors: nothing changed yet)
        bool committed = false;
        try {
            DevBuf d_start, d_ids;
            up.join();
            if (!uerr.empty()) throw Error(uerr);
            // (tried: the 16 MB of document tables on a helper thread beside the text — the runtime's pageable staging then competes
            //  with the chunk copies: 21.4 instead of 20.1 ms)
            upload_tables(ix, hstart, hid, ndocs, d_start, d_ids);
            committed = true;
            reset_unbuilt(ix);
            ix.host_upload_ms = wall_ms() - tu;
            commit_layout(ix, L);
            ix.ids.swap(hid);
            ix.doc_start.swap(hstart);
            ix.host_tables_valid = true;
            std::string().swap(ix.host_text);
            ix.host_text_valid = false;  // the column lives on the device (and with the caller)
            ix.d_text_owned = std::move(text);
            ix.d_text = ix.d_text_owned.as<uint8_t>();
            ix.text_padded = true;
            ix.d_doc_start = std::move(d_start);
            ix.d_ids = std::move(d_ids);
            build_suffix_array(ix);
        } catch (...) {
            if (committed) reset_unbuilt(ix);
            else (void)hipStreamSynchronize(ix.stream);
            throw;
        }
    });
}

/* cdb_build_views: the same for documents that are separate strings on the host (ptrs[d], lens[d]) — exactly what
 * string_index::add collects (index.cpp:174-177: ids.push_back(id); data.push_back(view)).  The shim's build() is this call. */
int cdb_build_views(cdb_index* h, const int64_t* ids, const char* const* ptrs, const uint64_t* lens, uint64_t ndocs) {
    reserve_join();
    if (!h || (ndocs && (!ids || !ptrs || !lens))) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        std::vector<int64_t> hid(ids, ids + ndocs);
        std::vector<uint64_t> hstart(ndocs + 1);
        hstart[0] = 0;
        for (uint64_t d = 0; d < ndocs; ++d) {
            if (lens[d] && !ptrs[d]) throw Error("cdb_build_views: null document");
            hstart[d + 1] = hstart[d] + lens[d];
        }
        const Layout L = layout_of(hstart, ndocs);  // (throws the reference's capacity errors: nothing changed yet)
        const uint64_t n = L.size;
        bool committed = false;
        try {
            DevBuf text, d_start, d_ids;
            text.alloc(n + TEXT_PAD);
            CDB_HIP(hipMemsetAsync((uint8_t*)text.p + n, 0, TEXT_PAD, ix.stream));
            const double tu = wall_ms();
            if (n) upload_views(text.p, ptrs, hstart.data(), ndocs, n, ix.stream, ix.device, upload_stream(ix));
            upload_tables(ix, hstart, hid, ndocs, d_start, d_ids);
            committed = true;
            reset_unbuilt(ix);
            ix.host_upload_ms = wall_ms() - tu;
            commit_layout(ix, L);
            ix.ids.swap(hid);
            ix.doc_start.swap(hstart);
            ix.host_tables_valid = true;
            std::string().swap(ix.host_text);
            ix.host_text_valid = false;  // the column lives on the device (and with the caller)
            ix.d_text_owned = std::move(text);
            ix.d_text = ix.d_text_owned.as<uint8_t>();
            ix.text_padded = true;
            ix.d_doc_start = std::move(d_start);
            ix.d_ids = std::move(d_ids);
            build_suffix_array(ix);
        } catch (...) {
            if (committed) reset_unbuilt(ix);
            else (void)hipStreamSynchronize(ix.stream);
            throw;
        }
    });
}

int cdb_build_resident(cdb_index* h, const void* d_text, const uint64_t* d_doc_start, const int64_t* d_ids,
                       uint64_t ndocs) {
    reserve_join();
    if (!h || !d_doc_start || (ndocs && !d_ids)) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        if (((uintptr_t)d_text & 15u) != 0) throw Error("device text must be 16-byte aligned");
        hipStream_t s = ix.stream;
        DevBuf d_start, d_id, d_out;
        d_start.alloc((ndocs + 1) * sizeof(uint64_t));
        d_id.alloc(std::max<uint64_t>(ndocs, 1) * sizeof(int64_t));
        d_out.alloc(2 * sizeof(uint64_t));
        CDB_HIP(hipMemsetAsync(d_out.p, 0, 2 * sizeof(uint64_t), s));
        const int grid = (int)std::min<uint64_t>(ceil_div(ndocs + 1, 256), 1024);
        hipLaunchKernelGGL(layout_kernel, dim3(grid), dim3(256), 0, s, d_doc_start, d_ids, ndocs, d_start.as<uint64_t>(),
                           d_id.as<int64_t>(), d_out.as<unsigned long long>());
        uint64_t out[2] = {0, 0}, first = 0, total = 0;
        CDB_HIP(hipMemcpyAsync(out, d_out.p, sizeof(out), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipMemcpyAsync(&first, d_doc_start, 8, hipMemcpyDeviceToHost, s));
        CDB_HIP(hipMemcpyAsync(&total, d_start.as<uint64_t>() + ndocs, 8, hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        if (out[1]) throw Error("doc_start must be non-decreasing");
        if (first != 0) throw Error("d_doc_start[0] must be 0");
        const Layout L = layout_from(ndocs, total, out[0]);  // bits / mask / size / entry width exactly as index.cpp:182-208
        try {  // (nothing can fail between here and the commit: the tables are already on the device)
            reset_unbuilt(ix);
            commit_layout(ix, L);
            ix.ids.clear();
            ix.doc_start.assign(1, 0);
            std::string().swap(ix.host_text);
            ix.host_tables_valid = false;  // tables and text live on the device (fetched back on demand)
            ix.host_text_valid = false;
            ix.d_text_owned.release();
            ix.d_text = static_cast<const uint8_t*>(d_text);
            ix.text_padded = false;
            ix.d_doc_start = std::move(d_start);
            ix.d_ids = std::move(d_id);
            build_suffix_array(ix);
        } catch (...) {
            reset_unbuilt(ix);
            throw;
        }
    });
}

void cdb_free(void* p) { host_free(p); }

namespace {
int query_batch_impl(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out,
                     cdb_hits* hits) {
    if (!h || !out || (npat && !offsets)) return CDB_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    if (hits) std::memset(hits, 0, sizeof(*hits));
    const int rc = guarded(h, [&] {
        Index& ix = h->ix;
        for (uint64_t j = 0; j < npat; ++j)
            if (offsets[j + 1] <= offsets[j]) throw Error("Empty keywords are not allowed");  // index.cpp:239-241
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        const double t0 = wall_ms();
        hipStream_t s = ix.stream;
        const uint64_t base = npat ? offsets[0] : 0;
        const uint64_t nbytes = npat ? offsets[npat] - base : 0;
        ix.q_pat.ensure(nbytes + 16);
        ix.q_offs.ensure((npat + 1) * 8);
        std::vector<uint64_t> rel(npat + 1);
        for (uint64_t j = 0; j <= npat; ++j) rel[j] = npat ? offsets[j] - base : 0;
        if (nbytes) CDB_HIP(hipMemcpyAsync(ix.q_pat.p, blob + base, nbytes, hipMemcpyHostToDevice, s));
        CDB_HIP(hipMemcpyAsync(ix.q_offs.p, rel.data(), (npat + 1) * 8, hipMemcpyHostToDevice, s));
        CDB_HIP(hipStreamSynchronize(s));
        const double t1 = wall_ms();
        const DeviceCsr r = query_batch_on_device(ix, ix.q_pat.as<uint8_t>(), ix.q_offs.as<uint64_t>(), npat, hits != nullptr);
        CDB_HIP(hipStreamSynchronize(s));
        const double t2 = wall_ms();
        out->npat = npat;
        out->nrows = r.nrows;
        out->nhits = r.nhits;
        out->row_ptr = (uint64_t*)host_alloc((npat + 1) * 8);
        out->ids = (int64_t*)host_alloc(r.nrows * 8);
        out->counts = (int64_t*)host_alloc(r.nrows * 8);
        if (hits) {
            hits->hit_ptr = (uint64_t*)host_alloc((r.nrows + 1) * 8, true);
            hits->offsets = (uint64_t*)host_alloc(r.nhits * 8);
        }
        CDB_HIP(hipMemcpyAsync(out->row_ptr, ix.q_rowptr.p, (npat + 1) * 8, hipMemcpyDeviceToHost, s));
        if (r.nrows) {
            CDB_HIP(hipMemcpyAsync(out->ids, ix.q_ids.p, r.nrows * 8, hipMemcpyDeviceToHost, s));
            CDB_HIP(hipMemcpyAsync(out->counts, ix.q_counts.p, r.nrows * 8, hipMemcpyDeviceToHost, s));
            if (hits) {
                CDB_HIP(hipMemcpyAsync(hits->hit_ptr, ix.q_hitptr.p, (r.nrows + 1) * 8, hipMemcpyDeviceToHost, s));
                CDB_HIP(hipMemcpyAsync(hits->offsets, ix.q_hitoff.p, r.nhits * 8, hipMemcpyDeviceToHost, s));
            }
        }
        CDB_HIP(hipStreamSynchronize(s));
        ix.qstats.query_ms = wall_ms() - t0;
        ix.qstats.upload_ms = t1 - t0;
        ix.qstats.device_ms = t2 - t1;
        ix.qstats.download_ms = wall_ms() - t2;
        ix.qstats.nhits = r.nhits;
        ix.qstats.nrows = r.nrows;
    });
    if (rc != CDB_OK) {  // a later allocation or copy failed: nothing half-filled leaves the library
        cdb_result_free(out);
        if (hits) cdb_hits_free(hits);
    }
    return rc;
}
}  // namespace

int cdb_query_batch(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out) {
    return query_batch_impl(h, blob, offsets, npat, out, nullptr);
}

int cdb_query_batch_offsets(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out,
                            cdb_hits* hits) {
    if (!hits) return CDB_E_INVALID;
    return query_batch_impl(h, blob, offsets, npat, out, hits);
}

void cdb_hits_free(cdb_hits* x) {
    if (!x) return;
    host_free(x->hit_ptr);
    host_free(x->offsets);
    std::memset(x, 0, sizeof(*x));
}

namespace {
int query_or_impl(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t nkw, int64_t** ids, int64_t** counts,
                  size_t* nrows, bool ranked, int64_t corr_lo, int64_t corr_hi, uint64_t limit) {
    if (!h || !ids || !counts || !nrows || (nkw && !offsets)) return CDB_E_INVALID;
    *ids = nullptr;
    *counts = nullptr;
    *nrows = 0;
    return guarded(h, [&] {
        Index& ix = h->ix;
        if (nkw == 0) throw Error("The constraint list cannot be empty");  // interface.cpp:75-77
        for (uint64_t j = 0; j < nkw; ++j)
            if (offsets[j + 1] <= offsets[j]) throw Error("Empty keywords are not allowed");
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        hipStream_t s = ix.stream;
        const uint64_t base = offsets[0], nbytes = offsets[nkw] - base;
        ix.q_pat.ensure(nbytes + 16);
        ix.q_offs.ensure((nkw + 1) * 8);
        std::vector<uint64_t> rel(nkw + 1);
        for (uint64_t j = 0; j <= nkw; ++j) rel[j] = offsets[j] - base;
        CDB_HIP(hipMemcpyAsync(ix.q_pat.p, blob + base, nbytes, hipMemcpyHostToDevice, s));
        CDB_HIP(hipMemcpyAsync(ix.q_offs.p, rel.data(), (nkw + 1) * 8, hipMemcpyHostToDevice, s));
        const DeviceCsr r = ranked ? query_ranked_on_device(ix, ix.q_pat.as<uint8_t>(), ix.q_offs.as<uint64_t>(), nkw, corr_lo, corr_hi, limit)
                                   : query_or_on_device(ix, ix.q_pat.as<uint8_t>(), ix.q_offs.as<uint64_t>(), nkw);
        int64_t* hi = (int64_t*)host_alloc(r.nrows * 8);
        int64_t* hc = nullptr;
        try {
            hc = (int64_t*)host_alloc(r.nrows * 8);
        } catch (...) {
            host_free(hi);
            throw;
        }
        try {
            if (r.nrows) {
                CDB_HIP(hipMemcpyAsync(hi, ix.q_ids.p, r.nrows * 8, hipMemcpyDeviceToHost, s));
                CDB_HIP(hipMemcpyAsync(hc, ix.q_counts.p, r.nrows * 8, hipMemcpyDeviceToHost, s));
                CDB_HIP(hipStreamSynchronize(s));
            }
        } catch (...) {
            host_free(hi);
            host_free(hc);
            throw;
        }
        *ids = hi;
        *counts = hc;
        *nrows = (size_t)r.nrows;
    });
}
}  // namespace

int cdb_query_or(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t nkw, int64_t** ids, int64_t** counts,
                 size_t* nrows) {
    return query_or_impl(h, blob, offsets, nkw, ids, counts, nrows, false, 0, 0, 0);
}

int cdb_query_ranked(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t nkw, int64_t corr_lo, int64_t corr_hi,
                     uint64_t limit, int64_t** ids, int64_t** counts, size_t* nrows) {
    return query_or_impl(h, blob, offsets, nkw, ids, counts, nrows, true, corr_lo, corr_hi, limit);
}

int cdb_query_and(const cdb_key_query* keys, int nkeys, int ranked, int64_t corr_lo, int64_t corr_hi, uint64_t limit, int64_t** ids,
                  int64_t** counts, size_t* nrows) {
    if (!keys || nkeys < 1 || !ids || !counts || !nrows) return CDB_E_INVALID;
    *ids = nullptr;
    *counts = nullptr;
    *nrows = 0;
    cdb_index* lead = nullptr;  // the first string key: its stream runs the merge, its handle carries the error
    for (int k = 0; k < nkeys; ++k)
        if (keys[k].index && !lead) lead = keys[k].index;
    if (!lead) return CDB_E_INVALID;
    return cdb::query_and_with_lead(lead, keys, nkeys, ranked, corr_lo, corr_hi, limit, ids, counts, nrows);
}
}  // extern "C"

// the merge behind cdb_query_and on `lead`'s device and stream (shards.hip: every key of a sharded AND arrives as a row
// list resolved by its shards, and the first shard of the first key lends its device)
int cdb::query_and_with_lead(cdb_index* lead, const cdb_key_query* keys, int nkeys, int ranked, int64_t corr_lo, int64_t corr_hi,
                             uint64_t limit, int64_t** ids, int64_t** counts, size_t* nrows) {
    {
    return guarded(lead, [&] {
        Index& ix = lead->ix;
        for (int k = 0; k < nkeys; ++k) {
            const cdb_key_query& q = keys[k];
            if (q.index) {
                if (q.index->ix.device != ix.device) throw Error("cdb_query_and: all string keys must live on one GPU");
                if (q.nkw == 0) throw Error("The constraint list cannot be empty");  // interface.cpp:75-77
                if (!q.offsets) throw Error("cdb_query_and: keyword offsets missing");
                for (uint64_t j = 0; j < q.nkw; ++j)
                    if (q.offsets[j + 1] <= q.offsets[j]) throw Error("Empty keywords are not allowed");
            } else if (q.nrows && (!q.ids || !q.counts)) {
                throw Error("cdb_query_and: row list missing");
            }
        }
        // every handle involved stays locked until the merge has read its rows (address order: no lock inversion)
        std::vector<cdb_index*> hs;
        for (int k = 0; k < nkeys; ++k)
            if (keys[k].index) hs.push_back(keys[k].index);
        std::sort(hs.begin(), hs.end());
        hs.erase(std::unique(hs.begin(), hs.end()), hs.end());
        std::vector<std::unique_lock<std::mutex>> locks;
        for (cdb_index* p : hs) locks.emplace_back(p->ix.mu);
        DeviceScope dscope(ix);
        hipStream_t s = ix.stream;
        std::vector<DeviceRows> lists;
        std::vector<DevBuf> held;  // copies of rows that would be overwritten by a later query on the same handle
        for (int k = 0; k < nkeys; ++k) {
            const cdb_key_query& q = keys[k];
            DevBuf di, dc;
            uint64_t n = 0;
            if (q.index) {
                Index& kx = q.index->ix;
                StreamScope kss(kx.stream);
                const uint64_t base = q.offsets[0], nbytes = q.offsets[q.nkw] - base;
                kx.q_pat.ensure(nbytes + 16);
                kx.q_offs.ensure((q.nkw + 1) * 8);
                std::vector<uint64_t> rel(q.nkw + 1);
                for (uint64_t j = 0; j <= q.nkw; ++j) rel[j] = q.offsets[j] - base;
                CDB_HIP(hipMemcpyAsync(kx.q_pat.p, q.blob + base, nbytes, hipMemcpyHostToDevice, kx.stream));
                CDB_HIP(hipMemcpyAsync(kx.q_offs.p, rel.data(), (q.nkw + 1) * 8, hipMemcpyHostToDevice, kx.stream));
                const DeviceCsr r = query_or_on_device(kx, kx.q_pat.as<uint8_t>(), kx.q_offs.as<uint64_t>(), q.nkw);  // (synchronises)
                n = r.nrows;
                di.alloc(std::max<uint64_t>(n, 1) * 8);
                dc.alloc(std::max<uint64_t>(n, 1) * 8);
                if (n) {
                    CDB_HIP(hipMemcpyAsync(di.p, kx.q_ids.p, n * 8, hipMemcpyDeviceToDevice, s));
                    CDB_HIP(hipMemcpyAsync(dc.p, kx.q_counts.p, n * 8, hipMemcpyDeviceToDevice, s));
                }
            } else {  // rows resolved elsewhere (numeric / bool keys: index.cpp:63-74,129-173 return (id, 0) rows)
                n = q.nrows;
                di.alloc(std::max<uint64_t>(n, 1) * 8);
                dc.alloc(std::max<uint64_t>(n, 1) * 8);
                if (n) {
                    CDB_HIP(hipMemcpyAsync(di.p, q.ids, n * 8, hipMemcpyHostToDevice, s));
                    CDB_HIP(hipMemcpyAsync(dc.p, q.counts, n * 8, hipMemcpyHostToDevice, s));
                }
            }
            lists.push_back(DeviceRows{di.as<int64_t>(), dc.as<int64_t>(), n});
            held.push_back(std::move(di));
            held.push_back(std::move(dc));
        }
        CDB_HIP(hipStreamSynchronize(s));  // (host rows are pageable)
        const DeviceCsr r = and_merge_on_device(ix, lists, ranked != 0, corr_lo, corr_hi, limit);
        int64_t* hi = (int64_t*)host_alloc(r.nrows * 8);
        int64_t* hc = nullptr;
        try {
            hc = (int64_t*)host_alloc(r.nrows * 8);
            if (r.nrows) {
                CDB_HIP(hipMemcpyAsync(hi, ix.q_ids.p, r.nrows * 8, hipMemcpyDeviceToHost, s));
                CDB_HIP(hipMemcpyAsync(hc, ix.q_counts.p, r.nrows * 8, hipMemcpyDeviceToHost, s));
                CDB_HIP(hipStreamSynchronize(s));
            }
        } catch (...) {
            host_free(hi);
            host_free(hc);
            throw;
        }
        *ids = hi;
        *counts = hc;
        *nrows = (size_t)r.nrows;
    });
    }
}

extern "C" {

int cdb_query_spans(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t nkw, cdb_spans* out) {
    if (!h || !out || (nkw && !offsets)) return CDB_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    const int rc = guarded(h, [&] {
        Index& ix = h->ix;
        std::vector<uint64_t> rel{0};
        std::string pat;
        for (uint64_t j = 0; j < nkw; ++j) {  // empty highlight keywords never match (ac_automaton::insert)
            if (offsets[j + 1] <= offsets[j]) continue;
            pat.append(blob + offsets[j], offsets[j + 1] - offsets[j]);
            rel.push_back(pat.size());
        }
        const uint64_t npat = rel.size() - 1;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        hipStream_t s = ix.stream;
        SpanResult r;
        if (npat) {
            ix.q_pat.ensure(pat.size() + 16);
            ix.q_offs.ensure((npat + 1) * 8);
            CDB_HIP(hipMemcpyAsync(ix.q_pat.p, pat.data(), pat.size(), hipMemcpyHostToDevice, s));
            CDB_HIP(hipMemcpyAsync(ix.q_offs.p, rel.data(), (npat + 1) * 8, hipMemcpyHostToDevice, s));
            r = query_spans_on_device(ix, ix.q_pat.as<uint8_t>(), ix.q_offs.as<uint64_t>(), npat, pat.size());
        }
        out->ndocs = r.ndocs;
        out->nspans = r.nspans;
        out->ids = (int64_t*)host_alloc(r.ndocs * 8);
        out->span_ptr = (uint64_t*)host_alloc((r.ndocs + 1) * 8, true);
        out->begin = (uint64_t*)host_alloc(r.nspans * 8);
        out->end = (uint64_t*)host_alloc(r.nspans * 8);
        if (r.nspans) {
            CDB_HIP(hipMemcpyAsync(out->ids, ix.q_ids.p, r.ndocs * 8, hipMemcpyDeviceToHost, s));
            CDB_HIP(hipMemcpyAsync(out->span_ptr, ix.q_rowptr.p, (r.ndocs + 1) * 8, hipMemcpyDeviceToHost, s));
            CDB_HIP(hipMemcpyAsync(out->begin, ix.q_keys0.p, r.nspans * 8, hipMemcpyDeviceToHost, s));
            CDB_HIP(hipMemcpyAsync(out->end, ix.q_keys1.p, r.nspans * 8, hipMemcpyDeviceToHost, s));
            CDB_HIP(hipStreamSynchronize(s));
        }
    });
    if (rc != CDB_OK) cdb_spans_free(out);
    return rc;
}

void cdb_spans_free(cdb_spans* r) {
    if (!r) return;
    host_free(r->ids);
    host_free(r->span_ptr);
    host_free(r->begin);
    host_free(r->end);
    std::memset(r, 0, sizeof(*r));
}

void cdb_result_free(cdb_result* r) {
    if (!r) return;
    host_free(r->row_ptr);
    host_free(r->ids);
    host_free(r->counts);
    std::memset(r, 0, sizeof(*r));
}

// Single-keyword queries arrive from many host threads at once (the reference serves them from an
// httplib pool under a shared lock, database.cpp:388) and one GPU round trip costs ~100 us, so
// concurrent callers are coalesced: the first caller becomes the leader and resolves everything that
// queued up as ONE batched GPU query, then the next batch, until the queue is empty; the others wait
// for their slice of the result.  No artificial delay is added for a lone caller.
namespace {
struct PendingQuery {
    const char* kw;
    size_t len;
    int64_t *ids = nullptr, *counts = nullptr;
    size_t rows = 0;
    int rc = CDB_OK;
    bool done = false;
};

void run_coalesced(cdb_index* h, std::vector<PendingQuery*>& batch) {
    if (batch.size() == 1) {  // a lone keyword: one wavefront, one launch (query.hip: q_single_kernel)
        PendingQuery* q = batch[0];
        int64_t *ids = nullptr, *counts = nullptr;
        size_t rows = 0;
        bool answered = false;
        const int rc1 = guarded(h, [&] {
            Index& ix = h->ix;
            std::lock_guard<std::mutex> g(ix.mu);
            DeviceScope dscope(ix);
            const double t0 = wall_ms();
            answered = query_single_on_device(ix, q->kw, q->len, &ids, &counts, &rows);
            if (answered) ix.qstats.query_ms = wall_ms() - t0;
        });
        if (rc1 != CDB_OK) {
            q->rc = rc1;
            return;
        }
        if (answered) {
            q->rows = rows;
            q->ids = ids;
            q->counts = counts;
            q->rc = CDB_OK;
            return;
        }
    }
    std::string blob;
    std::vector<uint64_t> offs{0};
    for (auto* q : batch) {
        blob.append(q->kw, q->len);
        offs.push_back(blob.size());
    }
    cdb_result r;
    const int rc = cdb_query_batch(h, blob.data(), offs.data(), batch.size(), &r);
    for (size_t j = 0; j < batch.size(); ++j) {
        PendingQuery* q = batch[j];
        q->rc = rc;
        if (rc != CDB_OK) continue;
        const uint64_t a = r.row_ptr[j], b = r.row_ptr[j + 1];
        q->rows = (size_t)(b - a);
        q->ids = (int64_t*)std::malloc(std::max<size_t>(q->rows, 1) * 8);
        q->counts = (int64_t*)std::malloc(std::max<size_t>(q->rows, 1) * 8);
        if (!q->ids || !q->counts) {
            q->rc = CDB_E_DEVICE;
            continue;
        }
        std::memcpy(q->ids, r.ids + a, q->rows * 8);
        std::memcpy(q->counts, r.counts + a, q->rows * 8);
    }
    if (rc == CDB_OK) cdb_result_free(&r);
}
}  // namespace

int cdb_query(cdb_index* h, const char* keyword, size_t len, int64_t** ids, int64_t** counts, size_t* nrows) {
    if (!h || !ids || !counts || !nrows) return CDB_E_INVALID;
    *ids = nullptr;
    *counts = nullptr;
    *nrows = 0;
    Index& ix = h->ix;
    if (len == 0) {  // index.cpp:239-241
        set_error(ix, "Empty keywords are not allowed");
        return CDB_E_INVALID;
    }
    PendingQuery me{keyword, len};
    if (!ix.coalesce_queries) {
        std::vector<PendingQuery*> one{&me};
        run_coalesced(h, one);
    } else {
        // Leader / follower: whoever finds no leader takes everything queued so far (its own query included),
        // resolves it as ONE batched GPU query and then gives the leadership up — a waiter whose query arrived
        // meanwhile takes over.  A leader therefore never serves more than the batch holding its own query (under
        // sustained load it would otherwise never return), and leadership cannot be stranded by an exception.
        std::unique_lock<std::mutex> lk(ix.qmu);
        ix.qpending.push_back(&me);
        while (!me.done) {
            if (ix.qleader) {
                ix.qcv.wait(lk, [&] { return me.done || !ix.qleader; });
                continue;
            }
            ix.qleader = true;
            std::vector<void*> taken;
            taken.swap(ix.qpending);
            lk.unlock();
            try {
                std::vector<PendingQuery*> batch;
                batch.reserve(taken.size());
                for (void* p : taken) batch.push_back(static_cast<PendingQuery*>(p));
                run_coalesced(h, batch);
            } catch (...) {  // (host allocation failure while assembling the batch)
                for (void* p : taken) {
                    PendingQuery* q = static_cast<PendingQuery*>(p);
                    std::free(q->ids);
                    std::free(q->counts);
                    q->ids = q->counts = nullptr;
                    q->rc = CDB_E_DEVICE;
                }
                set_error(ix, "out of host memory");
            }
            lk.lock();
            for (void* p : taken) static_cast<PendingQuery*>(p)->done = true;
            ix.qleader = false;
            ix.qcv.notify_all();
        }
    }
    if (me.rc != CDB_OK) {
        std::free(me.ids);
        std::free(me.counts);
        return me.rc;
    }
    *ids = me.ids;
    *counts = me.counts;
    *nrows = me.rows;
    return CDB_OK;
}

namespace {
int query_batch_device_impl(cdb_index* h, const void* d_blob, const uint64_t* d_offsets, uint64_t npat, cdb_device_result* out,
                            cdb_device_hits* hits) {
    if (!h || !out) return CDB_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    if (hits) std::memset(hits, 0, sizeof(*hits));
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        const double t0 = wall_ms();
        const DeviceCsr r = query_batch_on_device(ix, static_cast<const uint8_t*>(d_blob), d_offsets, npat, hits != nullptr);
        out->npat = npat;
        out->nrows = r.nrows;
        out->nhits = r.nhits;
        out->d_row_ptr = ix.q_rowptr.as<uint64_t>();
        out->d_ids = ix.q_ids.as<int64_t>();
        out->d_counts = ix.q_counts.as<int64_t>();
        if (hits) {
            hits->d_hit_ptr = ix.q_hitptr.as<uint64_t>();
            hits->d_offsets = ix.q_hitoff.as<uint64_t>();
        }
        ix.qstats.query_ms = wall_ms() - t0;
        ix.qstats.nhits = r.nhits;
        ix.qstats.nrows = r.nrows;
    });
}
}  // namespace

int cdb_query_batch_device(cdb_index* h, const void* d_blob, const uint64_t* d_offsets, uint64_t npat,
                           uint64_t blob_bytes, cdb_device_result* out) {
    (void)blob_bytes;
    return query_batch_device_impl(h, d_blob, d_offsets, npat, out, nullptr);
}

int cdb_query_batch_offsets_device(cdb_index* h, const void* d_blob, const uint64_t* d_offsets, uint64_t npat,
                                   uint64_t blob_bytes, cdb_device_result* out, cdb_device_hits* hits) {
    (void)blob_bytes;
    if (!hits) return CDB_E_INVALID;
    return query_batch_device_impl(h, d_blob, d_offsets, npat, out, hits);
}

uint64_t cdb_size(const cdb_index* h) { return h ? h->ix.size : 0; }
uint64_t cdb_bits(const cdb_index* h) { return h ? h->ix.bits : 0; }
uint64_t cdb_mask(const cdb_index* h) { return h ? h->ix.mask : 0; }
int cdb_sa_width(const cdb_index* h) { return h ? h->ix.width : 0; }

int cdb_sa_copy(cdb_index* h, void* host_out, uint64_t capacity_bytes) {
    if (!h || !host_out) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        const uint64_t need = ix.size * (uint64_t)ix.width;
        if (capacity_bytes < need) throw Error("cdb_sa_copy: buffer too small");
        if (need && ix.sa_packed) {  // the caller sees the reference's u64 entries (index.cpp:203-208), expanded chunk by chunk
            DevBuf chunk;
            const uint64_t per = std::min<uint64_t>(ix.size, 32ull << 20);
            chunk.alloc(per * 8);
            for (uint64_t first = 0; first < ix.size; first += per) {
                const uint64_t cnt = std::min<uint64_t>(per, ix.size - first);
                sa_expand(ix, first, cnt, chunk.as<uint64_t>());
                CDB_HIP(hipMemcpyAsync(static_cast<char*>(host_out) + first * 8, chunk.p, cnt * 8, hipMemcpyDeviceToHost, ix.stream));
                CDB_HIP(hipStreamSynchronize(ix.stream));
            }
        } else if (need) {
            CDB_HIP(hipMemcpyAsync(host_out, ix.d_sa.p, need, hipMemcpyDeviceToHost, ix.stream));
            CDB_HIP(hipStreamSynchronize(ix.stream));
        }
    });
}

int cdb_set_option(cdb_index* h, const char* name, int64_t value) {
    if (!h || !name) return CDB_E_INVALID;
    Index& ix = h->ix;
    // The resident query workgroup keeps the key arrays / symbol count it was launched with, while a request only carries
    // what query_single_launch encodes under the CURRENT options: every option that changes that encoding stops the
    // workgroup first (the next lone keyword starts a fresh one under the new settings).
    for (const char* o : {"fast_search", "key_directory", "single_query", "search_lanes", "keep_keys", "reference_compat"})
        if (!std::strcmp(name, o)) {
            std::lock_guard<std::mutex> g(ix.mu);
            query_resident_stop(ix);
        }
    if (!std::strcmp(name, "profile")) ix.prof.enabled = value != 0;
    else if (!std::strcmp(name, "reference_compat")) ix.reference_compat = value != 0;
    else if (!std::strcmp(name, "force_doubling")) ix.force_doubling = value != 0;
    else if (!std::strcmp(name, "initial_passes")) ix.initial_passes = (int)value;
    else if (!std::strcmp(name, "sort_variant")) ix.sort_variant = (int)value;
    else if (!std::strcmp(name, "keyhist3")) ix.keyhist3 = value != 0;
    else if (!std::strcmp(name, "key_symbols")) ix.key_symbols = (int)value;
    else if (!std::strcmp(name, "msd_first")) ix.msd_first = value != 0;
    else if (!std::strcmp(name, "msd_pair")) ix.msd_pair = value != 0;
    else if (!std::strcmp(name, "overlap_paircount")) ix.overlap_paircount = value != 0;
    else if (!std::strcmp(name, "key_directory")) { ix.key_directory = value != 0; ix.h_keydir.clear(); ix.keydir_tried = false; }
    else if (!std::strcmp(name, "fuse_records")) ix.fuse_records = value != 0;
    else if (!std::strcmp(name, "sweep_records")) ix.sweep_records = value != 0;
    else if (!std::strcmp(name, "partial_symbol")) ix.partial_symbol = value != 0;
    else if (!std::strcmp(name, "group_sort")) ix.group_sort = value != 0;
    else if (!std::strcmp(name, "list_rounds")) ix.list_rounds = value != 0;
    else if (!std::strcmp(name, "fuse_pairclass")) ix.fuse_pairclass = value != 0;
    else if (!std::strcmp(name, "group_sort_cap")) ix.group_sort_cap = (int)std::max<int64_t>(1, std::min<int64_t>(value, 1 << 20));
    else if (!std::strcmp(name, "vl_keys")) ix.vl_keys = value < 0 ? 0 : (value > 56 ? 56 : (int)value);
    else if (!std::strcmp(name, "pack_sa")) ix.pack_sa = value != 0;
    else if (!std::strcmp(name, "key_cost_model")) ix.key_cost_model = value != 0;
    else if (!std::strcmp(name, "gen_prebased")) ix.gen_prebased = value != 0;
    else if (!std::strcmp(name, "pack_entries")) ix.pack_entries = value != 0;
    else if (!std::strcmp(name, "segmented_sort")) ix.segmented_sort = value != 0;
    else if (!std::strcmp(name, "fold_root")) ix.fold_root = value != 0;
    else if (!std::strcmp(name, "fold_depth1")) ix.fold_depth1 = value != 0;
    else if (!std::strcmp(name, "flags_in_last_pass")) ix.flags_in_last_pass = value != 0;
    else if (!std::strcmp(name, "search_lanes")) ix.search_lanes = (value == 1 || value == 8) ? (int)value : 0;
    else if (!std::strcmp(name, "digit_bits")) ix.digit_bits = (int)value;
    else if (!std::strcmp(name, "fuse_keygen")) ix.fuse_keygen = value != 0;
    else if (!std::strcmp(name, "force_big_path")) ix.force_big_path = value != 0;
    else if (!std::strcmp(name, "debug_fail_build")) ix.debug_fail_build = value != 0;
    else if (!std::strcmp(name, "debug_no_segcap")) ix.debug_no_segcap = value != 0;  // test hook: "a bucket does not fit the record memory"
    else if (!std::strcmp(name, "debug_starve_group")) ix.debug_starve_group = (int)value;  // 1 = reported after the sorts, 2 = error flag up before the initial sort
    else if (!std::strcmp(name, "self_check")) ix.self_check = value < 0 ? 0 : value > 3 ? 3 : (int)value;  // 0 off, 1 sample, 2 every pair inline, 3 sample + proof after publish
    else if (!std::strcmp(name, "premap_generation")) ix.premap_generation = value != 0;
    else if (!std::strcmp(name, "proof_cancel")) {  // stops the order proof in flight (state 5); the next build / load starts a new one
        if (value) proof_stop(ix);
    }
    else if (!std::strcmp(name, "debug_damage_after_build")) ix.debug_damage_after_build = value < 0 ? 0 : (uint64_t)value;
    else if (!std::strcmp(name, "debug_fail_self_check")) ix.debug_fail_self_check = value != 0;
    else if (!std::strcmp(name, "plain_tile_order")) ix.rws.plain_order = value != 0;
    else if (!std::strcmp(name, "key_coding")) ix.key_coding = (int)value;
    else if (!std::strcmp(name, "narrow_keys")) ix.narrow_keys = value != 0;
    else if (!std::strcmp(name, "single_query")) ix.use_single_query = value != 0;
    else if (!std::strcmp(name, "resident_query")) {
        std::lock_guard<std::mutex> g(ix.mu);
        if (value != 1) query_resident_stop(ix);
        ix.resident_mode = value <= 0 ? 0 : value == 1 ? 1 : 2;  // 0 never, 1 always, 2 automatic (default)
        ix.resident_query = ix.resident_mode == 1;
        ix.single_streak = 0;
    }
    else if (!std::strcmp(name, "bucket_group_limit")) ix.bucket_group_limit = (uint64_t)value;
    else if (!std::strcmp(name, "query_hit_budget"))  // <= 2^31: one kernel launch addresses < 2^32 threads
        ix.query_hit_budget = value > 0 ? std::min<uint64_t>((uint64_t)value, 1ull << 31) : 1;
    else if (!std::strcmp(name, "coalesce_queries")) ix.coalesce_queries = value != 0;
    else if (!std::strcmp(name, "fast_search")) ix.use_fast_search = value != 0;
    else if (!std::strcmp(name, "wave_rows")) ix.use_wave_rows = value != 0;
    else if (!std::strcmp(name, "keep_keys")) ix.keep_keys = value != 0;
    else {
        set_error(ix, (std::string("unknown option: ") + name).c_str());
        return CDB_E_INVALID;
    }
    return CDB_OK;
}

int cdb_get_stat(const cdb_index* h, const char* name, double* value) {
    if (!h || !name || !value) return CDB_E_INVALID;
    const BuildStats& b = h->ix.bstats;
    const QueryStats& q = h->ix.qstats;
    struct { const char* n; double v; } tab[] = {
        {"build_ms", b.build_ms}, {"alloc_ms", b.alloc_ms}, {"free_ms", b.free_ms}, {"rounds", (double)b.rounds}, {"ext_rounds", (double)b.ext_rounds},
        {"dbl_rounds", (double)b.dbl_rounds}, {"unresolved_after_initial", (double)b.unresolved_initial},
        {"unresolved_max", (double)b.unresolved_max}, {"sort_passes", (double)b.sort_passes},
        {"sort_passes_skipped", (double)b.sort_passes_skipped}, {"isa_built", (double)b.isa_built}, {"fused_keygen", (double)b.fused_keygen}, {"dense_keys", (double)b.dense_keys}, {"key_layout", (double)b.key_layout}, {"bucketed", (double)b.bucketed}, {"bucket_groups", (double)b.bucket_groups}, {"segmented", (double)b.segmented}, {"fused_records", (double)b.fused_records}, {"sweep_records", (double)b.sweep_records}, {"vl_key_bits", (double)b.vl_key_bits}, {"partial_levels", (double)b.partial_levels}, {"list_rounds", (double)b.list_rounds}, {"pairclass_fused", (double)b.pairclass_fused}, {"group_sorts", (double)b.group_sorts}, {"group_sort_fallbacks", (double)b.group_sort_fallbacks}, {"vl_avg_len", b.vl_avg_len}, {"vl_rate", b.vl_rate}, {"vl_est_unresolved", b.vl_est_unresolved}, {"fixed_est_unresolved", b.fixed_est_unresolved}, {"gen_prebased", (double)b.gen_prebased}, {"root_folded", (double)b.root_folded}, {"flags_in_last_pass", (double)b.flags_in_last_pass}, {"msd_first", (double)b.msd_first}, {"bucket_low_digits", (double)b.bucket_low_digits}, {"key_directory_cells", h->ix.h_keydir.empty() ? 0.0 : (double)(h->ix.h_keydir.size() - 1)}, {"group_fallbacks", (double)h->ix.group_fallbacks}, {"dense_key_retries", (double)h->ix.dense_key_retries}, {"self_check_fallbacks", (double)h->ix.self_check_fallbacks}, {"sa_packed", h->ix.sa_packed ? 1.0 : 0.0}, {"sa_bytes_per_entry", h->ix.sa_packed ? 5.0 : (double)h->ix.width}, {"self_check_pairs", (double)h->ix.self_check_pairs}, {"self_check_ms", h->ix.self_check_ms}, {"self_check_coverage", h->ix.size > 1 ? (double)h->ix.self_check_pairs / (double)(h->ix.size - 1) : 0.0},
        {"order_proved", (h->ix.proof.state.load() == 2 || h->ix.proof.state.load() == 3 || (h->ix.self_check == 2 && h->ix.width != 0)) ? 1.0 : 0.0},
        {"proof_state", (double)h->ix.proof.state.load()}, {"proof_ms", h->ix.proof.ms}, {"proof_repair_ms", h->ix.proof.repair_ms},
        {"proof_pairs", (double)h->ix.proof.pairs}, {"proof_bad_pairs", (double)h->ix.proof.found[0]}, {"proof_invalid_entries", (double)h->ix.proof.found[1]}, {"proof_skipped_pairs", (double)h->ix.proof.skipped}, {"proof_mixed_pairs", (double)h->ix.proof.mixed},
        {"proof_runs", (double)h->ix.proof.runs}, {"pool_big_mallocs", (double)DevPool::get().big_mallocs()}, {"premap_ms", h->ix.proof.premap_ms}, {"premap_bytes", (double)h->ix.proof.premap_bytes},
        {"key_symbols", (double)b.key_symbols}, {"symbol_bits", (double)b.symbol_bits},
        {"alphabet", (double)b.alphabet}, {"digit_bits", (double)b.digit_bits}, {"final_depth", (double)b.final_depth}, {"compat_rotations", (double)b.compat_rotations},
        {"compat_depth", (double)b.compat_depth},
        {"host_upload_ms", h->ix.host_upload_ms}, {"host_free_ms", h->ix.host_free_ms},
        {"resident_answers", (double)h->ix.res_answers}, {"launched_answers", (double)h->ix.launched_answers}, {"resident_mode", (double)h->ix.resident_mode},
        {"query_ms", q.query_ms}, {"query_upload_ms", q.upload_ms}, {"query_device_ms", q.device_ms}, {"query_download_ms", q.download_ms}, {"query_hits", (double)q.nhits}, {"query_rows", (double)q.nrows},
    };
    for (auto& e : tab)
        if (!std::strcmp(e.n, name)) {
            *value = e.v;
            return CDB_OK;
        }
    return CDB_E_INVALID;
}

int cdb_profile_get(cdb_index* h, const char* kernel, double* total_ms, uint64_t* launches, uint64_t* bytes) {
    if (!h || !kernel) return CDB_E_INVALID;
    std::lock_guard<std::mutex> g(h->ix.mu);
    auto it = h->ix.prof.recs.find(kernel);
    if (it == h->ix.prof.recs.end()) return CDB_E_INVALID;
    if (total_ms) *total_ms = it->second.ms;
    if (launches) *launches = it->second.launches;
    if (bytes) *bytes = it->second.bytes;
    return CDB_OK;
}

int cdb_profile_dump(cdb_index* h, char* buf, size_t cap) {
    if (!h || !buf || cap == 0) return CDB_E_INVALID;
    std::lock_guard<std::mutex> g(h->ix.mu);
    std::string s;
    for (auto& kv : h->ix.prof.recs) {
        char line[256];
        std::snprintf(line, sizeof(line), "%s %.6f %llu %llu\n", kv.first.c_str(), kv.second.ms,
                      (unsigned long long)kv.second.launches, (unsigned long long)kv.second.bytes);
        s += line;
    }
    std::strncpy(buf, s.c_str(), cap - 1);
    buf[cap - 1] = 0;
    return CDB_OK;
}

int cdb_debug_verify(cdb_index* h, uint64_t out[5]) {
    if (!h || !out) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        if (ix.width == 0) throw Error("index has not been built");
        verify_suffix_array(ix, out);
    });
}

int cdb_debug_self_check(cdb_index* h, int full, uint64_t out[2]) {
    if (!h || !out) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        if (ix.width == 0) throw Error("index has not been built");
        spot_check_suffix_array(ix, full ? 0u : (uint32_t)std::min<uint64_t>(1u << 15, ix.size > 1 ? ix.size - 1 : 1), out);
    });
}

int cdb_proof_wait(cdb_index* h, double timeout_ms) {
    if (!h) return CDB_E_INVALID;
    // (no lock: the state is atomic, and a repair in progress holds ix.mu itself)
    const double t0 = wall_ms();
    for (;;) {
        const int st = h->ix.proof.state.load();
        if (st != 1 && !h->ix.proof.busy.load(std::memory_order_acquire)) return st;
        if (timeout_ms >= 0 && wall_ms() - t0 >= timeout_ms) return st;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

int cdb_debug_verify_reference(cdb_index* h, uint64_t out[4]) {
    if (!h || !out) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        DeviceScope dscope(ix);
        if (ix.width == 0) throw Error("index has not been built");
        verify_reference_order(ix, out);
    });
}

int cdb_debug_radix_sort(int device, void* d_keys, void* d_vals, uint64_t n, int val_bytes, int key_bits,
                         int variant, double* onesweep_ms, int* passes) {
    if (!d_keys || (val_bytes != 0 && val_bytes != 4 && val_bytes != 8) || key_bits < 1 || key_bits > 64)
        return CDB_E_INVALID;
    try {
        if (device >= 0) CDB_HIP(hipSetDevice(device));
        hipStream_t s;
        CDB_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        {
            StreamScope sscope(s);
            RadixWorkspace ws;
            Profiler prof;
            prof.enabled = true;
            DevBuf k1, v1;
            k1.alloc(n * 8);
            if (val_bytes) v1.alloc(n * val_bytes);
            SortStats st;
            int sel = 0;
            uint64_t* k0 = static_cast<uint64_t*>(d_keys);
            if (val_bytes == 4)
                sel = radix_sort<uint64_t, uint32_t>(s, ws, prof, k0, k1.as<uint64_t>(), static_cast<uint32_t*>(d_vals),
                                                     v1.as<uint32_t>(), n, 0, key_bits, &st, variant);
            else if (val_bytes == 8)
                sel = radix_sort<uint64_t, uint64_t>(s, ws, prof, k0, k1.as<uint64_t>(), static_cast<uint64_t*>(d_vals),
                                                     v1.as<uint64_t>(), n, 0, key_bits, &st, variant);
            else
                sel = radix_sort<uint64_t, NoVal>(s, ws, prof, k0, k1.as<uint64_t>(), (NoVal*)nullptr, (NoVal*)nullptr, n,
                                                  0, key_bits, &st, variant);
            if (sel == 1) {
                CDB_HIP(hipMemcpyAsync(d_keys, k1.p, n * 8, hipMemcpyDeviceToDevice, s));
                if (val_bytes) CDB_HIP(hipMemcpyAsync(d_vals, v1.p, n * val_bytes, hipMemcpyDeviceToDevice, s));
            }
            radix_check_error(s, ws);
            CDB_HIP(hipStreamSynchronize(s));
            prof.resolve();
            double ms = 0;
            for (auto& kv : prof.recs)
                if (kv.first.rfind("rs_onesweep", 0) == 0) ms += kv.second.ms;  // every tile size
            if (onesweep_ms) *onesweep_ms = ms;
            if (passes) *passes = st.passes_run;
        }
        DevPool::get().retire_stream(s);
        (void)hipStreamDestroy(s);
        return CDB_OK;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "cdb_debug_radix_sort: %s\n", e.what());
        return CDB_E_DEVICE;
    }
}

int cdb_debug_query_latency(cdb_index* h, const char* blob, const uint64_t* offsets, size_t nkw, int reps, double* us_out) {
    if (!h || !blob || !offsets || !us_out || reps < 1) return CDB_E_INVALID;
    std::vector<double> t((size_t)reps);
    for (size_t k = 0; k < nkw; ++k) {
        const char* kw = blob + offsets[k];
        const size_t len = (size_t)(offsets[k + 1] - offsets[k]);
        for (int r = 0; r < reps; ++r) {
            int64_t *ids = nullptr, *counts = nullptr;
            size_t rows = 0;
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = cdb_query(h, kw, len, &ids, &counts, &rows);
            t[(size_t)r] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rc != CDB_OK) return rc;
            cdb_free(ids);
            cdb_free(counts);
        }
        std::sort(t.begin(), t.end());
        us_out[k] = t[t.size() / 2];
    }
    return CDB_OK;
}

int cdb_layout_rule(uint64_t ndocs, uint64_t longest, uint64_t* bits, uint64_t* mask, int* width, int* off_bits, char* err,
                    size_t err_cap) {
    try {
        const Layout L = layout_from(ndocs, 0, longest);
        if (bits) *bits = L.bits;
        if (mask) *mask = L.mask;
        if (width) *width = L.width;
        if (off_bits) *off_bits = L.off_bits;
        return CDB_OK;
    } catch (const std::exception& e) {
        if (err && err_cap) {
            std::strncpy(err, e.what(), err_cap - 1);
            err[err_cap - 1] = 0;
        }
        return CDB_E_INVALID;
    }
}

void cdb_release_cached_memory(void) {
    reserve_join();
    DevPool::get().trim();
    HostPool::get().trim();
}

void cdb_set_cache_limit(uint64_t bytes) { DevPool::get().set_limit((size_t)bytes); }
void cdb_memory_stats(uint64_t* in_use_bytes, uint64_t* peak_bytes, uint64_t* cached_bytes) {
    size_t a = 0, b = 0, c = 0;
    DevPool::get().stats(a, b, c);
    if (in_use_bytes) *in_use_bytes = a;
    if (peak_bytes) *peak_bytes = b;
    if (cached_bytes) *cached_bytes = c;
}
void cdb_memory_reset_peak(void) { DevPool::get().reset_peak(); }

uint64_t cdb_cached_memory_bytes(void) { return (uint64_t)DevPool::get().cached_bytes(); }

void cdb_profile_reset(cdb_index* h) {
    if (!h) return;
    std::lock_guard<std::mutex> g(h->ix.mu);
    h->ix.prof.reset();
}

}  // extern "C"
