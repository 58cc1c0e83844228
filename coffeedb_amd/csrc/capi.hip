// capi.hip — the C ABI of libcoffeedb_gpu.so (include/coffeedb_gpu.h).  Host logic only: staging of
// cdb_add, the reference's width rule, uploads/downloads and error translation.  All device work is in
// sa_build.hip / query.hip / radix_sort.h.
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>

#include "../../include/coffeedb_gpu.h"
#include "index_impl.h"

using namespace cdb;

struct cdb_index {
    Index ix;
};

namespace {

double wall_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

template <typename F>
int guarded(cdb_index* h, F&& f) {
    try {
        f();
        return CDB_OK;
    } catch (const Error& e) {
        h->ix.err = e.what();
        const bool dev = std::strncmp(e.what(), "HIP error", 9) == 0;
        const bool internal = std::strstr(e.what(), "internal") != nullptr;
        return dev ? CDB_E_DEVICE : (internal ? CDB_E_INTERNAL : CDB_E_INVALID);
    } catch (const std::bad_alloc&) {
        h->ix.err = "out of host memory";
        return CDB_E_DEVICE;
    } catch (const std::exception& e) {
        h->ix.err = e.what();
        return CDB_E_INTERNAL;
    }
}

// bits / mask / size / width exactly as string_index::build does (reference src/index.cpp:182-208)
void compute_layout(Index& ix) {
    const uint64_t ndocs = ix.ids.size();
    uint64_t size = 0, mask1 = 1, mask2 = 1;
    while (mask1 < ndocs) mask1 = (mask1 << 1) + 1;
    for (uint64_t d = 0; d < ndocs; ++d) {
        const uint64_t len = ix.doc_start[d + 1] - ix.doc_start[d];
        size += len;
        while (mask2 < len) mask2 = (mask2 << 1) + 1;
    }
    const int bits1 = __builtin_popcountll(mask1), bits2 = __builtin_popcountll(mask2);
    if (bits1 + bits2 > 64) throw Error("The amount of data exceeds the maximum range that CoffeeDB can handle");
    if (bits1 > 32) throw Error("The number of objects exceeds the maximum range that CoffeeDB can handle");
    ix.size = size;
    ix.mask = mask1;
    ix.bits = (uint64_t)bits1;
    ix.width = bits1 + bits2 <= 32 ? 4 : 8;
    ix.off_bits = bits2;
    ix.ndocs = ndocs;
}

void upload_tables(Index& ix) {
    hipStream_t s = ix.stream;
    ix.d_doc_start.alloc((ix.ndocs + 1) * sizeof(uint64_t));
    CDB_HIP(hipMemcpyAsync(ix.d_doc_start.p, ix.doc_start.data(), (ix.ndocs + 1) * sizeof(uint64_t),
                           hipMemcpyHostToDevice, s));
    ix.d_ids.alloc(std::max<uint64_t>(ix.ndocs, 1) * sizeof(int64_t));
    if (ix.ndocs)
        CDB_HIP(hipMemcpyAsync(ix.d_ids.p, ix.ids.data(), ix.ndocs * sizeof(int64_t), hipMemcpyHostToDevice, s));
}

void set_device(Index& ix) { CDB_HIP(hipSetDevice(ix.device)); }

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
}  // namespace cdb

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
    if (h->ix.stream) {
        (void)hipStreamSynchronize(h->ix.stream);
        (void)hipStreamDestroy(h->ix.stream);
    }
    if (h->ix.h_single) (void)hipHostFree(h->ix.h_single);
    delete h;
}

const char* cdb_last_error(const cdb_index* h) { return h ? h->ix.err.c_str() : "null handle"; }

int cdb_add(cdb_index* h, int64_t id, const char* value, size_t len) {
    if (!h || (!value && len)) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        ix.ids.push_back(id);
        ix.host_text.append(value, len);
        ix.doc_start.push_back(ix.host_text.size());
    });
}

int cdb_add_bulk(cdb_index* h, const int64_t* ids, const char* blob, const uint64_t* doc_start, uint64_t ndocs) {
    if (!h || (ndocs && (!ids || !doc_start))) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        if (!ndocs) return;
        const uint64_t base = ix.host_text.size();
        ix.host_text.append(blob + doc_start[0], doc_start[ndocs] - doc_start[0]);
        ix.ids.reserve(ix.ids.size() + ndocs);
        ix.doc_start.reserve(ix.doc_start.size() + ndocs);
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
        h->ix.err = "malformed raw record";
        return CDB_E_INVALID;
    }
    if (r == 0) return CDB_OK;  // the object has no string value under this key
    return cdb_add(h, id, v, vl);
}

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
        set_device(ix);
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
        dump(ix.d_sa.p, ix.size * (uint64_t)ix.width);
        if (!ok) throw Error(std::string("Cannot write file: ") + path);
    });
}

int cdb_load(cdb_index* h, const char* path) {
    if (!h || !path) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        set_device(ix);
        FILE* fp = std::fopen(path, "rb");
        if (!fp) throw Error(std::string("Cannot open file: ") + path);
        struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{fp};
        SaveHeader hd{};
        if (std::fread(&hd, sizeof(hd), 1, fp) != 1 || hd.magic != SAVE_MAGIC || (hd.width != 4 && hd.width != 8))
            throw Error(std::string("Not a saved index: ") + path);
        ix.ids.resize(hd.ndocs);
        ix.doc_start.resize(hd.ndocs + 1);
        bool ok = hd.ndocs == 0 || std::fread(ix.ids.data(), 8, hd.ndocs, fp) == hd.ndocs;
        ok = ok && std::fread(ix.doc_start.data(), 8, hd.ndocs + 1, fp) == hd.ndocs + 1;
        if (!ok || ix.doc_start[hd.ndocs] != hd.size) throw Error(std::string("Truncated index file: ") + path);
        ix.size = hd.size; ix.ndocs = hd.ndocs; ix.bits = hd.bits; ix.mask = hd.mask; ix.width = (int)hd.width;
        ix.reference_compat = hd.compat != 0;
        {
            uint64_t mask2 = 1;
            for (uint64_t d = 0; d < hd.ndocs; ++d)
                while (mask2 < ix.doc_start[d + 1] - ix.doc_start[d]) mask2 = (mask2 << 1) + 1;
            ix.off_bits = __builtin_popcountll(mask2);
        }
        ix.sa_sorted = hd.sorted != 0;  // a reference-compat ordering keeps the reference's exact probe sequence
        ix.pivot_levels = 0;
        ix.drop_keys();
        ix.host_text.clear();
        ix.d_text_owned.alloc(ix.size + TEXT_PAD);
        CDB_HIP(hipMemsetAsync((uint8_t*)ix.d_text_owned.p + ix.size, 0, TEXT_PAD, ix.stream));
        ix.d_text = ix.d_text_owned.as<uint8_t>();
        ix.text_padded = true;
        ix.d_sa.alloc(std::max<uint64_t>(ix.size * hd.width, 16));
        std::vector<char> buf(std::min<uint64_t>(std::max<uint64_t>(ix.size * hd.width, 1), 256ull << 20));
        auto fill = [&](void* dptr, uint64_t bytes) {
            for (uint64_t o = 0; o < bytes; o += buf.size()) {
                const uint64_t c = std::min<uint64_t>(buf.size(), bytes - o);
                if (std::fread(buf.data(), 1, c, fp) != c) throw Error(std::string("Truncated index file: ") + path);
                CDB_HIP(hipMemcpyAsync(static_cast<char*>(dptr) + o, buf.data(), c, hipMemcpyHostToDevice, ix.stream));
                CDB_HIP(hipStreamSynchronize(ix.stream));
            }
        };
        fill(ix.d_text_owned.p, ix.size);
        fill(ix.d_sa.p, ix.size * hd.width);
        upload_tables(ix);
        CDB_HIP(hipStreamSynchronize(ix.stream));
    });
}

int cdb_build(cdb_index* h) {
    if (!h) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        set_device(ix);
        compute_layout(ix);
        const uint64_t n = ix.size;
        ix.d_text_owned.alloc(n + TEXT_PAD);
        CDB_HIP(hipMemsetAsync((uint8_t*)ix.d_text_owned.p + n, 0, TEXT_PAD, ix.stream));
        if (n) CDB_HIP(hipMemcpyAsync(ix.d_text_owned.p, ix.host_text.data(), n, hipMemcpyHostToDevice, ix.stream));
        ix.d_text = ix.d_text_owned.as<uint8_t>();
        ix.text_padded = true;
        upload_tables(ix);
        build_suffix_array(ix);
    });
}

int cdb_build_device(cdb_index* h, const void* d_text, const uint64_t* doc_start, const int64_t* ids, uint64_t ndocs) {
    if (!h || (ndocs && (!doc_start || !ids))) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        set_device(ix);
        if (((uintptr_t)d_text & 15u) != 0) throw Error("device text must be 16-byte aligned");
        ix.ids.assign(ids, ids + ndocs);
        ix.doc_start.resize(ndocs + 1);
        ix.doc_start[0] = 0;
        for (uint64_t d = 0; d < ndocs; ++d) {
            if (doc_start[d + 1] < doc_start[d]) throw Error("doc_start must be non-decreasing");
            ix.doc_start[d + 1] = doc_start[d + 1] - doc_start[0];
        }
        ix.host_text.clear();
        compute_layout(ix);
        ix.d_text_owned.release();
        ix.d_text = static_cast<const uint8_t*>(d_text) + doc_start[0];
        if (doc_start[0] & 15u) throw Error("first document must start 16-byte aligned");
        ix.text_padded = false;
        upload_tables(ix);
        build_suffix_array(ix);
    });
}

int cdb_build_resident(cdb_index* h, const void* d_text, const uint64_t* d_doc_start, const int64_t* d_ids,
                       uint64_t ndocs) {
    if (!h || !d_doc_start || (ndocs && !d_ids)) return CDB_E_INVALID;
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        set_device(ix);
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
        // bits / mask / size / entry width exactly as index.cpp:182-208
        uint64_t mask1 = 1, mask2 = 1;
        while (mask1 < ndocs) mask1 = (mask1 << 1) + 1;
        while (mask2 < out[0]) mask2 = (mask2 << 1) + 1;
        const int bits1 = __builtin_popcountll(mask1), bits2 = __builtin_popcountll(mask2);
        if (bits1 + bits2 > 64) throw Error("The amount of data exceeds the maximum range that CoffeeDB can handle");
        if (bits1 > 32) throw Error("The number of objects exceeds the maximum range that CoffeeDB can handle");
        ix.size = total;
        ix.mask = mask1;
        ix.bits = (uint64_t)bits1;
        ix.width = bits1 + bits2 <= 32 ? 4 : 8;
        ix.off_bits = bits2;
        ix.ndocs = ndocs;
        ix.ids.clear();
        ix.doc_start.assign(1, 0);
        ix.host_text.clear();
        ix.host_tables_valid = false;
        ix.d_text_owned.release();
        ix.d_text = static_cast<const uint8_t*>(d_text);
        ix.text_padded = false;
        ix.d_doc_start = std::move(d_start);
        ix.d_ids = std::move(d_id);
        build_suffix_array(ix);
    });
}

void cdb_free(void* p) { host_free(p); }

namespace {
int query_batch_impl(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out,
                     cdb_hits* hits) {
    if (!h || !out || (npat && !offsets)) return CDB_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    if (hits) std::memset(hits, 0, sizeof(*hits));
    return guarded(h, [&] {
        Index& ix = h->ix;
        for (uint64_t j = 0; j < npat; ++j)
            if (offsets[j + 1] <= offsets[j]) throw Error("Empty keywords are not allowed");  // index.cpp:239-241
        std::lock_guard<std::mutex> g(ix.mu);
        set_device(ix);
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
        set_device(ix);
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
        if (r.nrows) {
            CDB_HIP(hipMemcpyAsync(hi, ix.q_ids.p, r.nrows * 8, hipMemcpyDeviceToHost, s));
            CDB_HIP(hipMemcpyAsync(hc, ix.q_counts.p, r.nrows * 8, hipMemcpyDeviceToHost, s));
            CDB_HIP(hipStreamSynchronize(s));
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

int cdb_query_spans(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t nkw, cdb_spans* out) {
    if (!h || !out || (nkw && !offsets)) return CDB_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    return guarded(h, [&] {
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
        set_device(ix);
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
            set_device(ix);
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
        std::lock_guard<std::mutex> g(ix.qmu);
        ix.err = "Empty keywords are not allowed";
        return CDB_E_INVALID;
    }
    PendingQuery me{keyword, len};
    if (!ix.coalesce_queries) {
        std::vector<PendingQuery*> one{&me};
        run_coalesced(h, one);
    } else {
        std::unique_lock<std::mutex> lk(ix.qmu);
        ix.qpending.push_back(&me);
        if (!ix.qleader) {
            ix.qleader = true;
            while (!ix.qpending.empty()) {
                std::vector<void*> taken;
                taken.swap(ix.qpending);
                lk.unlock();
                std::vector<PendingQuery*> batch;
                for (void* p : taken) batch.push_back(static_cast<PendingQuery*>(p));
                run_coalesced(h, batch);
                lk.lock();
                for (auto* q : batch) q->done = true;
                ix.qcv.notify_all();
            }
            ix.qleader = false;
        } else {
            ix.qcv.wait(lk, [&] { return me.done; });
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

int cdb_query_batch_device(cdb_index* h, const void* d_blob, const uint64_t* d_offsets, uint64_t npat,
                           uint64_t blob_bytes, cdb_device_result* out) {
    if (!h || !out) return CDB_E_INVALID;
    (void)blob_bytes;
    std::memset(out, 0, sizeof(*out));
    return guarded(h, [&] {
        Index& ix = h->ix;
        std::lock_guard<std::mutex> g(ix.mu);
        set_device(ix);
        const double t0 = wall_ms();
        const DeviceCsr r = query_batch_on_device(ix, static_cast<const uint8_t*>(d_blob), d_offsets, npat);
        out->npat = npat;
        out->nrows = r.nrows;
        out->nhits = r.nhits;
        out->d_row_ptr = ix.q_rowptr.as<uint64_t>();
        out->d_ids = ix.q_ids.as<int64_t>();
        out->d_counts = ix.q_counts.as<int64_t>();
        ix.qstats.query_ms = wall_ms() - t0;
        ix.qstats.nhits = r.nhits;
        ix.qstats.nrows = r.nrows;
    });
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
        set_device(ix);
        const uint64_t need = ix.size * (uint64_t)ix.width;
        if (capacity_bytes < need) throw Error("cdb_sa_copy: buffer too small");
        if (need) {
            CDB_HIP(hipMemcpyAsync(host_out, ix.d_sa.p, need, hipMemcpyDeviceToHost, ix.stream));
            CDB_HIP(hipStreamSynchronize(ix.stream));
        }
    });
}

int cdb_set_option(cdb_index* h, const char* name, int64_t value) {
    if (!h || !name) return CDB_E_INVALID;
    Index& ix = h->ix;
    if (!std::strcmp(name, "profile")) ix.prof.enabled = value != 0;
    else if (!std::strcmp(name, "reference_compat")) ix.reference_compat = value != 0;
    else if (!std::strcmp(name, "force_doubling")) ix.force_doubling = value != 0;
    else if (!std::strcmp(name, "initial_passes")) ix.initial_passes = (int)value;
    else if (!std::strcmp(name, "sort_variant")) ix.sort_variant = (int)value;
    else if (!std::strcmp(name, "digit_bits")) ix.digit_bits = (int)value;
    else if (!std::strcmp(name, "fuse_keygen")) ix.fuse_keygen = value != 0;
    else if (!std::strcmp(name, "force_big_path")) ix.force_big_path = value != 0;
    else if (!std::strcmp(name, "key_coding")) ix.key_coding = (int)value;
    else if (!std::strcmp(name, "narrow_keys")) ix.narrow_keys = value != 0;
    else if (!std::strcmp(name, "single_query")) ix.use_single_query = value != 0;
    else if (!std::strcmp(name, "bucket_group_limit")) ix.bucket_group_limit = (uint64_t)value;
    else if (!std::strcmp(name, "query_hit_budget"))  // <= 2^31: one kernel launch addresses < 2^32 threads
        ix.query_hit_budget = value > 0 ? std::min<uint64_t>((uint64_t)value, 1ull << 31) : 1;
    else if (!std::strcmp(name, "coalesce_queries")) ix.coalesce_queries = value != 0;
    else if (!std::strcmp(name, "fast_search")) ix.use_fast_search = value != 0;
    else if (!std::strcmp(name, "wave_rows")) ix.use_wave_rows = value != 0;
    else if (!std::strcmp(name, "keep_keys")) ix.keep_keys = value != 0;
    else {
        ix.err = std::string("unknown option: ") + name;
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
        {"sort_passes_skipped", (double)b.sort_passes_skipped}, {"isa_built", (double)b.isa_built}, {"fused_keygen", (double)b.fused_keygen}, {"dense_keys", (double)b.dense_keys}, {"key_layout", (double)b.key_layout}, {"bucketed", (double)b.bucketed}, {"bucket_groups", (double)b.bucket_groups},
        {"key_symbols", (double)b.key_symbols}, {"symbol_bits", (double)b.symbol_bits},
        {"alphabet", (double)b.alphabet}, {"digit_bits", (double)b.digit_bits}, {"final_depth", (double)b.final_depth}, {"compat_rotations", (double)b.compat_rotations},
        {"compat_depth", (double)b.compat_depth},
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
        set_device(ix);
        if (ix.width == 0) throw Error("index has not been built");
        verify_suffix_array(ix, out);
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
        (void)hipStreamDestroy(s);
        return CDB_OK;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "cdb_debug_radix_sort: %s\n", e.what());
        return CDB_E_DEVICE;
    }
}

void cdb_release_cached_memory(void) {
    DevPool::get().trim();
    HostPool::get().trim();
}

void cdb_set_cache_limit(uint64_t bytes) { DevPool::get().set_limit((size_t)bytes); }

uint64_t cdb_cached_memory_bytes(void) { return (uint64_t)DevPool::get().cached_bytes(); }

void cdb_profile_reset(cdb_index* h) {
    if (!h) return;
    std::lock_guard<std::mutex> g(h->ix.mu);
    h->ix.prof.reset();
}

}  // extern "C"
