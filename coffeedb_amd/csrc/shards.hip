// shards.hip — the string index across several GPUs, behind the C ABI.
//
// Suffixes never cross document boundaries (reference src/index.h:61-65) and a result row belongs to exactly one
// document (src/index.cpp:317-321), so a column splits into document-aligned byte ranges with one independent
// suffix array per GPU.  Every shard answers the whole pattern batch for its documents; concatenating the rows of
// a pattern in shard order is already ascending in document index, so the merge is an all-gatherv plus a
// placement — no reduction:
//     1. all-gather of the per-pattern row counts (u32 x patterns),
//     2. one exclusive scan over the (shard, pattern) counts = where every shard's rows of every pattern sit in the
//        gathered stream, and the merged row_ptr,
//     3. all-gatherv of the (id, count) rows — grouped ncclBroadcast, one root per shard (xGMI is point to point
//        and every peer is one hop away: a direct exchange, not a ring),
//     4. a placement kernel moves the rows from shard-major to pattern-major order.
// Two ways in: cdb_shards_* — ONE process owning G devices (what the CoffeeDB shim uses: database.cpp is a single
// process), one host thread per shard; cdb_comm_* — one process per GPU (bench.py under torchrun), each rank
// merging its own shard.  Both run the same merge over RCCL (loaded at run time: single-GPU users need no RCCL);
// shards that share a device (tests on a one-GPU box) exchange through device-to-device copies instead, because
// RCCL refuses two ranks on one GPU.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <shared_mutex>
#include <thread>

#include "../../include/coffeedb_gpu.h"
#include "index_impl.h"
#include "scan.h"

using namespace cdb;

namespace {

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;

    static RcclApi& get() {
        static RcclApi api;
        static std::once_flag once;
        std::call_once(once, [] {
            // (a process that already loaded an RCCL — PyTorch bundles one — gets that one back by soname)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (api.lib) break;
            }
            if (!api.lib) return;
            auto sym = [&](const char* n) { return dlsym(api.lib, n); };
            api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
            api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
            api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
            api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
            api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
            api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommInitAll || !api.CommDestroy || !api.AllGather || !api.Broadcast ||
                !api.GroupStart || !api.GroupEnd) {
                dlclose(api.lib);
                api.lib = nullptr;
            }
        });
        return api;
    }
    bool ok() const { return lib != nullptr; }
};

#define CDB_NCCL(expr)                                                                                    \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) {                                                                          \
            const RcclApi& a_ = RcclApi::get();                                                           \
            throw ::cdb::Error(std::string("HIP error in " #expr " (RCCL): ") +                           \
                               (a_.GetErrorString ? a_.GetErrorString(r_) : "unknown"));                  \
        }                                                                                                 \
    } while (0)

// ---- exchange between the shards of one merge ---------------------------------------------------------------------
struct Transport {
    virtual ~Transport() = default;
    // recv[q * bytes .. ) = rank q's `bytes` bytes of send, for every q
    virtual void all_gather(int rank, const void* send, void* recv, size_t bytes, hipStream_t s) = 0;
    // recv[off[q] * 8 .. ) = rank q's cnt[q] 8-byte elements of send (send holds cnt[rank] elements)
    virtual void all_gather_v(int rank, const void* send, void* recv, const uint64_t* cnt, const uint64_t* off, hipStream_t s) = 0;
    virtual const char* name() const = 0;
};

// one RCCL communicator per rank (ranks may live in one process — cdb_shards — or in one process each — cdb_comm)
struct RcclTransport : Transport {
    int world;
    std::vector<ncclComm_t> comms;  // indexed by LOCAL rank slot
    std::vector<int> local_rank;    // global rank of every local slot
    explicit RcclTransport(int w) : world(w) {}
    ~RcclTransport() override {
        for (ncclComm_t c : comms)
            if (c) (void)RcclApi::get().CommDestroy(c);
    }
    ncclComm_t comm_of(int rank) const {
        for (size_t i = 0; i < local_rank.size(); ++i)
            if (local_rank[i] == rank) return comms[i];
        throw Error("internal: rank is not local to this communicator");
    }
    void all_gather(int rank, const void* send, void* recv, size_t bytes, hipStream_t s) override {
        CDB_NCCL(RcclApi::get().AllGather(send, recv, bytes, ncclUint8, comm_of(rank), s));
    }
    void all_gather_v(int rank, const void* send, void* recv, const uint64_t* cnt, const uint64_t* off, hipStream_t s) override {
        // all-gatherv = one broadcast per root inside a group (every rank knows every count, so empty roots are
        // skipped consistently)
        const RcclApi& a = RcclApi::get();
        ncclComm_t c = comm_of(rank);
        CDB_NCCL(a.GroupStart());
        for (int q = 0; q < world; ++q) {
            if (!cnt[q]) continue;
            char* dst = static_cast<char*>(recv) + off[q] * 8;
            ncclResult_t r = a.Broadcast(q == rank ? send : (const void*)dst, dst, cnt[q], ncclUint64, q, c, s);
            if (r != ncclSuccess) {
                (void)a.GroupEnd();
                CDB_NCCL(r);
            }
        }
        CDB_NCCL(a.GroupEnd());
    }
    const char* name() const override { return "rccl"; }
};

// shards of ONE process that share devices: a board of published buffers, device-to-device copies
struct LocalTransport : Transport {
    int world;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    std::vector<const void*> send;
    explicit LocalTransport(int w) : world(w), send(w, nullptr) {}
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = gen;
        if (++arrived == world) {
            arrived = 0;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
    }
    void exchange(int rank, const void* snd, void* recv, const uint64_t* bytes, const uint64_t* off_bytes, hipStream_t s) {
        CDB_HIP(hipStreamSynchronize(s));  // this rank's buffer is complete
        {
            std::lock_guard<std::mutex> g(mu);
            send[rank] = snd;
        }
        barrier();  // every buffer is published and complete
        for (int q = 0; q < world; ++q)
            if (bytes[q])
                CDB_HIP(hipMemcpyAsync(static_cast<char*>(recv) + off_bytes[q], send[q], bytes[q], hipMemcpyDeviceToDevice, s));
        CDB_HIP(hipStreamSynchronize(s));
        barrier();  // every rank has read every buffer: they may be reused
    }
    void all_gather(int rank, const void* snd, void* recv, size_t bytes, hipStream_t s) override {
        std::vector<uint64_t> b(world, bytes), o(world);
        for (int q = 0; q < world; ++q) o[q] = (uint64_t)q * bytes;
        exchange(rank, snd, recv, b.data(), o.data(), s);
    }
    void all_gather_v(int rank, const void* snd, void* recv, const uint64_t* cnt, const uint64_t* off, hipStream_t s) override {
        std::vector<uint64_t> b(world), o(world);
        for (int q = 0; q < world; ++q) {
            b[q] = cnt[q] * 8;
            o[q] = off[q] * 8;
        }
        exchange(rank, snd, recv, b.data(), o.data(), s);
    }
    const char* name() const override { return "device copies"; }
};

// ---- merge kernels ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sh_counts_kernel(const uint64_t* __restrict__ row_ptr, uint64_t npat, uint32_t* __restrict__ cnt) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < npat) cnt[j] = (uint32_t)(row_ptr[j + 1] - row_ptr[j]);
}
struct FlatCntIn {  // (shard, pattern) counts in shard-major order = the order of the gathered row stream
    const uint32_t* c;
    __device__ __forceinline__ uint64_t operator()(uint64_t i) const { return c[i]; }
};
struct FlatPtrOut {
    uint64_t* p;
    uint64_t n;
    __device__ __forceinline__ void operator()(uint64_t i, uint64_t ex, uint64_t in) const {
        p[i] = ex;
        if (i + 1 == n) p[n] = in;
    }
};
struct ColSumIn {  // rows of pattern j over all shards
    const uint32_t* c;
    uint64_t npat;
    int world;
    __device__ __forceinline__ uint64_t operator()(uint64_t j) const {
        uint64_t t = 0;
        for (int q = 0; q < world; ++q) t += c[(uint64_t)q * npat + j];
        return t;
    }
};
__global__ __launch_bounds__(64) void sh_rank_offsets_kernel(const uint64_t* __restrict__ flat_ptr, uint64_t npat, int world,
                                                             uint64_t* __restrict__ out /*[world + 1]*/) {
    for (int q = threadIdx.x; q <= world; q += 64) out[q] = flat_ptr[(uint64_t)q * npat];  // (any number of ranks)
}
// merged row i: its pattern (search in the merged row_ptr), then the shard whose rows of that pattern cover it
__global__ __launch_bounds__(256) void sh_place_kernel(const uint64_t* __restrict__ g_row_ptr, const uint32_t* __restrict__ all_cnt,
                                                       const uint64_t* __restrict__ flat_ptr, uint64_t npat, int world, uint64_t total,
                                                       const int64_t* __restrict__ in_ids, const int64_t* __restrict__ in_cnt,
                                                       int64_t* __restrict__ out_ids, int64_t* __restrict__ out_cnt) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        uint64_t lo = 0, hi = npat - 1;  // largest j with g_row_ptr[j] <= i
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo + 1) / 2;
            if (g_row_ptr[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const uint64_t j = lo;
        uint64_t k = i - g_row_ptr[j];
        int q = 0;
        for (; q < world - 1; ++q) {
            const uint64_t c = all_cnt[(uint64_t)q * npat + j];
            if (k < c) break;
            k -= c;
        }
        const uint64_t src = flat_ptr[(uint64_t)q * npat + j] + k;
        out_ids[i] = in_ids[src];
        out_cnt[i] = in_cnt[src];
    }
}

// ---- one rank of a merge ------------------------------------------------------------------------------------------
struct MergeRank {
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::shared_ptr<Transport> tr;
    DevBuf cnt, all_cnt, flat_ptr, g_row_ptr, offs, st_ids, st_cnt, out_ids, out_cnt, partials;
    std::mutex err_mu;
    std::string err;
    ~MergeRank() {
        if (own_stream && stream) {
            (void)hipSetDevice(device);
            (void)hipStreamSynchronize(stream);
            for (DevBuf* b : {&cnt, &all_cnt, &flat_ptr, &g_row_ptr, &offs, &st_ids, &st_cnt, &out_ids, &out_cnt, &partials}) b->release();
            DevPool::get().retire_stream(stream);
            (void)hipStreamDestroy(stream);
        }
    }
};

// collective over all ranks of mr.tr; local = this shard's device CSR; merged arrays live in mr
void merge_core(MergeRank& mr, const cdb_device_result& local, cdb_device_result& merged) {
    CDB_HIP(hipSetDevice(mr.device));
    StreamScope ss(mr.stream);
    hipStream_t s = mr.stream;
    const uint64_t npat = local.npat;
    const int G = mr.world;
    std::memset(&merged, 0, sizeof(merged));
    merged.npat = npat;
    mr.g_row_ptr.ensure((npat + 1) * 8);
    if (npat == 0) {
        CDB_HIP(hipMemsetAsync(mr.g_row_ptr.p, 0, 8, s));
        CDB_HIP(hipStreamSynchronize(s));
        merged.d_row_ptr = mr.g_row_ptr.as<uint64_t>();
        return;
    }
    mr.cnt.ensure(npat * 4);
    mr.all_cnt.ensure((size_t)G * npat * 4);
    mr.flat_ptr.ensure(((size_t)G * npat + 1) * 8);
    mr.offs.ensure((G + 2) * 8);
    hipLaunchKernelGGL(sh_counts_kernel, dim3((unsigned)ceil_div(npat, 256)), dim3(256), 0, s, local.d_row_ptr, npat, mr.cnt.as<uint32_t>());
    mr.tr->all_gather(mr.rank, mr.cnt.p, mr.all_cnt.p, npat * 4, s);
    // where every (shard, pattern) group starts in the gathered stream, and the merged row_ptr
    FlatCntIn fin{mr.all_cnt.as<uint32_t>()};
    scan_totals_device<uint64_t>(s, mr.partials, fin, (uint64_t)G * npat, OpAdd{}, (uint64_t)0);
    scan_apply<uint64_t>(s, mr.partials, fin, (uint64_t)G * npat, OpAdd{}, (uint64_t)0, FlatPtrOut{mr.flat_ptr.as<uint64_t>(), (uint64_t)G * npat});
    ColSumIn cin{mr.all_cnt.as<uint32_t>(), npat, G};
    scan_totals_device<uint64_t>(s, mr.partials, cin, npat, OpAdd{}, (uint64_t)0);
    scan_apply<uint64_t>(s, mr.partials, cin, npat, OpAdd{}, (uint64_t)0, FlatPtrOut{mr.g_row_ptr.as<uint64_t>(), npat});
    hipLaunchKernelGGL(sh_rank_offsets_kernel, dim3(1), dim3(64), 0, s, (const uint64_t*)mr.flat_ptr.as<uint64_t>(), npat, G,
                       mr.offs.as<uint64_t>());
    std::vector<uint64_t> off(G + 1), cntv(G);
    CDB_HIP(hipMemcpyAsync(off.data(), mr.offs.p, (G + 1) * 8, hipMemcpyDeviceToHost, s));
    CDB_HIP(hipGetLastError());
    CDB_HIP(hipStreamSynchronize(s));
    for (int q = 0; q < G; ++q) cntv[q] = off[q + 1] - off[q];
    const uint64_t total = off[G];
    if (cntv[mr.rank] != local.nrows) throw Error("internal: shard row count does not match its row_ptr");
    merged.nrows = total;
    merged.d_row_ptr = mr.g_row_ptr.as<uint64_t>();
    if (total == 0) return;
    mr.st_ids.ensure(total * 8);
    mr.st_cnt.ensure(total * 8);
    mr.out_ids.ensure(total * 8);
    mr.out_cnt.ensure(total * 8);
    mr.tr->all_gather_v(mr.rank, local.d_ids, mr.st_ids.p, cntv.data(), off.data(), s);
    mr.tr->all_gather_v(mr.rank, local.d_counts, mr.st_cnt.p, cntv.data(), off.data(), s);
    const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(total, 256), 1u << 20);
    hipLaunchKernelGGL(sh_place_kernel, dim3(grid), dim3(256), 0, s, (const uint64_t*)mr.g_row_ptr.as<uint64_t>(),
                       (const uint32_t*)mr.all_cnt.as<uint32_t>(), (const uint64_t*)mr.flat_ptr.as<uint64_t>(), npat, G, total,
                       (const int64_t*)mr.st_ids.as<int64_t>(), (const int64_t*)mr.st_cnt.as<int64_t>(), mr.out_ids.as<int64_t>(),
                       mr.out_cnt.as<int64_t>());
    CDB_HIP(hipGetLastError());
    CDB_HIP(hipStreamSynchronize(s));
    merged.d_ids = mr.out_ids.as<int64_t>();
    merged.d_counts = mr.out_cnt.as<int64_t>();
}

// Counts-only merge for host (or rank-local) consumers ("otherwise each GPU D2H's its slice"): the ranks
// exchange nothing but their per-pattern row counts; every rank learns the merged row_ptr and where ITS rows of every
// pattern start in the merged row stream.  No rank ever holds another rank's rows — at C4 (10^7 patterns x 8 shards)
// the full all-gatherv would put ~10^9 rows on every GPU.
__global__ __launch_bounds__(256) void sh_base_kernel(const uint64_t* __restrict__ g_row_ptr, const uint32_t* __restrict__ all_cnt,
                                                      uint64_t npat, int rank, uint64_t* __restrict__ base) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= npat) return;
    uint64_t b = g_row_ptr[j];
    for (int q = 0; q < rank; ++q) b += all_cnt[(uint64_t)q * npat + j];
    base[j] = b;
}

void merge_counts_core(MergeRank& mr, const cdb_device_result& local, cdb_shard_slice& out) {
    CDB_HIP(hipSetDevice(mr.device));
    StreamScope ss(mr.stream);
    hipStream_t s = mr.stream;
    const uint64_t npat = local.npat;
    const int G = mr.world;
    std::memset(&out, 0, sizeof(out));
    out.npat = npat;
    out.nrows_local = local.nrows;
    mr.g_row_ptr.ensure((npat + 1) * 8);
    if (npat == 0) {
        CDB_HIP(hipMemsetAsync(mr.g_row_ptr.p, 0, 8, s));
        CDB_HIP(hipStreamSynchronize(s));
        out.d_row_ptr = mr.g_row_ptr.as<uint64_t>();
        return;
    }
    mr.cnt.ensure(npat * 4);
    mr.all_cnt.ensure((size_t)G * npat * 4);
    mr.flat_ptr.ensure(npat * 8);  // (here: this rank's row bases)
    hipLaunchKernelGGL(sh_counts_kernel, dim3((unsigned)ceil_div(npat, 256)), dim3(256), 0, s, local.d_row_ptr, npat, mr.cnt.as<uint32_t>());
    mr.tr->all_gather(mr.rank, mr.cnt.p, mr.all_cnt.p, npat * 4, s);
    ColSumIn cin{mr.all_cnt.as<uint32_t>(), npat, G};
    scan_totals_device<uint64_t>(s, mr.partials, cin, npat, OpAdd{}, (uint64_t)0);
    scan_apply<uint64_t>(s, mr.partials, cin, npat, OpAdd{}, (uint64_t)0, FlatPtrOut{mr.g_row_ptr.as<uint64_t>(), npat});
    hipLaunchKernelGGL(sh_base_kernel, dim3((unsigned)ceil_div(npat, 256)), dim3(256), 0, s, (const uint64_t*)mr.g_row_ptr.as<uint64_t>(),
                       (const uint32_t*)mr.all_cnt.as<uint32_t>(), npat, mr.rank, mr.flat_ptr.as<uint64_t>());
    uint64_t total = 0;
    CDB_HIP(hipMemcpyAsync(&total, mr.g_row_ptr.as<uint64_t>() + npat, 8, hipMemcpyDeviceToHost, s));
    CDB_HIP(hipGetLastError());
    CDB_HIP(hipStreamSynchronize(s));
    out.nrows_total = total;
    out.d_row_ptr = mr.g_row_ptr.as<uint64_t>();
    out.d_row_base = mr.flat_ptr.as<uint64_t>();
}

template <typename T, typename F>
int guarded_on(T* obj, F&& f) {
    ForegroundCall fg;  // (order proofs yield to calls in flight: common.h)
    try {
        f();
        return CDB_OK;
    } catch (const Error& e) {
        std::lock_guard<std::mutex> g(obj->err_mu);
        obj->err = e.what();
        const bool dev = std::strncmp(e.what(), "HIP error", 9) == 0;
        return dev ? CDB_E_DEVICE : (std::strstr(e.what(), "internal") ? CDB_E_INTERNAL : CDB_E_INVALID);
    } catch (const std::bad_alloc&) {
        std::lock_guard<std::mutex> g(obj->err_mu);
        obj->err = "out of host memory";
        return CDB_E_DEVICE;
    } catch (const std::exception& e) {
        std::lock_guard<std::mutex> g(obj->err_mu);
        obj->err = e.what();
        return CDB_E_INTERNAL;
    }
}

// doc-aligned split into `parts` contiguous ranges balanced by bytes: docs [b[r], b[r+1])
std::vector<uint64_t> shard_bounds(const std::vector<uint64_t>& doc_start, int parts) {
    const uint64_t nd = doc_start.size() - 1, total = doc_start[nd];
    std::vector<uint64_t> b{0};
    for (int r = 1; r < parts; ++r) {
        const unsigned __int128 target128 = (unsigned __int128)total * r / parts;
        const uint64_t target = (uint64_t)target128;
        uint64_t d = std::lower_bound(doc_start.begin(), doc_start.end(), target) - doc_start.begin();
        d = std::min<uint64_t>(std::max<uint64_t>(d, b.back()), nd);
        b.push_back(d);
    }
    b.push_back(nd);
    return b;
}

}  // namespace

// =====================================================================================================================
// one process per GPU
// =====================================================================================================================
struct cdb_comm {
    MergeRank mr;
};

extern "C" {

int cdb_comm_unique_id(void* id128) {
    if (!id128) return CDB_E_INVALID;
    const RcclApi& a = RcclApi::get();
    if (!a.ok()) return CDB_E_DEVICE;
    ncclUniqueId id;
    if (a.GetUniqueId(&id) != ncclSuccess) return CDB_E_DEVICE;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return CDB_OK;
}

int cdb_comm_create(cdb_comm** out, const void* id128, int rank, int world, int device) {
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return CDB_E_INVALID;
    *out = nullptr;
    const RcclApi& a = RcclApi::get();
    if (!a.ok()) return CDB_E_DEVICE;
    cdb_comm* c = new (std::nothrow) cdb_comm();
    if (!c) return CDB_E_DEVICE;
    try {
        if (device < 0) CDB_HIP(hipGetDevice(&device));
        CDB_HIP(hipSetDevice(device));
        c->mr.rank = rank;
        c->mr.world = world;
        c->mr.device = device;
        CDB_HIP(hipStreamCreateWithFlags(&c->mr.stream, hipStreamNonBlocking));
        c->mr.own_stream = true;
        auto tr = std::make_shared<RcclTransport>(world);
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        ncclComm_t comm = nullptr;
        CDB_NCCL(a.CommInitRank(&comm, world, id, rank));
        tr->comms.push_back(comm);
        tr->local_rank.push_back(rank);
        c->mr.tr = tr;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "cdb_comm_create: %s\n", e.what());
        delete c;
        return CDB_E_DEVICE;
    }
    *out = c;
    return CDB_OK;
}

// the ranks of ONE process (a host thread per rank calls the collectives): RCCL over distinct devices, device-to-device copies when
// ranks share a device — e.g. BASELINE config 3's four 8 GiB shards co-resident on one MI355X (tests/test_gpu_fullsize.py)
int cdb_comm_create_group(cdb_comm** out, int world, const int* devices) {
    if (!out || !devices || world < 1) return CDB_E_INVALID;
    for (int i = 0; i < world; ++i) out[i] = nullptr;
    try {
        bool distinct = true;
        for (int i = 0; i < world; ++i)
            for (int j = 0; j < i; ++j)
                if (devices[i] == devices[j]) distinct = false;
        const char* force = std::getenv("CDB_SHARD_TRANSPORT");
        std::shared_ptr<Transport> tr;
        if (world > 1 && distinct && RcclApi::get().ok() && !(force && std::string(force) == "copy")) {
            auto r = std::make_shared<RcclTransport>(world);
            r->comms.assign(world, nullptr);
            CDB_NCCL(RcclApi::get().CommInitAll(r->comms.data(), world, devices));
            for (int i = 0; i < world; ++i) r->local_rank.push_back(i);
            tr = r;
        } else {
            tr = std::make_shared<LocalTransport>(world);
        }
        for (int i = 0; i < world; ++i) {
            cdb_comm* c = new cdb_comm();
            out[i] = c;
            c->mr.rank = i;
            c->mr.world = world;
            c->mr.device = devices[i];
            c->mr.tr = tr;
            CDB_HIP(hipSetDevice(devices[i]));
            CDB_HIP(hipStreamCreateWithFlags(&c->mr.stream, hipStreamNonBlocking));
            c->mr.own_stream = true;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "cdb_comm_create_group: %s\n", e.what());
        for (int i = 0; i < world; ++i) {
            delete out[i];
            out[i] = nullptr;
        }
        return CDB_E_DEVICE;
    }
    return CDB_OK;
}

void cdb_comm_destroy(cdb_comm* c) { delete c; }

const char* cdb_comm_last_error(const cdb_comm* c) {
    if (!c) return "null handle";
    static thread_local std::string copy;
    std::lock_guard<std::mutex> g(const_cast<cdb_comm*>(c)->mr.err_mu);
    copy = c->mr.err;
    return copy.c_str();
}

int cdb_comm_merge(cdb_comm* c, const cdb_device_result* local, cdb_device_result* merged) {
    if (!c || !local || !merged) return CDB_E_INVALID;
    return guarded_on(&c->mr, [&] { merge_core(c->mr, *local, *merged); });
}

int cdb_comm_merge_counts(cdb_comm* c, const cdb_device_result* local, cdb_shard_slice* out) {
    if (!c || !local || !out) return CDB_E_INVALID;
    return guarded_on(&c->mr, [&] { merge_counts_core(c->mr, *local, *out); });
}

int cdb_comm_world(const cdb_comm* c) { return c ? c->mr.world : 0; }
const char* cdb_comm_transport(const cdb_comm* c) { return (c && c->mr.tr) ? c->mr.tr->name() : "none"; }

}  // extern "C"

// =====================================================================================================================
// one process, G devices
// =====================================================================================================================
// Locking: `state` guards WHICH shard handles exist — builds and loads replace them under the exclusive
// lock, every query holds it shared for the whole call, so a query never sees a destroyed handle or a half-built set.
// Device work of one shard is serialised by that shard's own ix.mu; code that holds several of them at once (the
// lone-keyword fan-out, the device merge) takes them in ascending shard order.
// Shared / exclusive lock that PREFERS the writer: glibc's rwlock lets new readers in while a writer waits, so
// two client threads with overlapping queries could keep a finished build from ever taking over.  Here a waiting writer
// closes the door for new readers; the ones inside finish, the generation swap runs, the door opens.  (No thread takes
// the shared side twice: nothing below nests two shared sections.)
class StateLock {
    std::mutex m;
    std::condition_variable cv;
    int readers = 0, writers_waiting = 0;
    bool writer = false;

public:
    void lock_shared() {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !writer && writers_waiting == 0; });
        ++readers;
    }
    void unlock_shared() {
        std::unique_lock<std::mutex> l(m);
        if (--readers == 0) cv.notify_all();
    }
    void lock() {
        std::unique_lock<std::mutex> l(m);
        ++writers_waiting;
        cv.wait(l, [&] { return !writer && readers == 0; });
        --writers_waiting;
        writer = true;
    }
    void unlock() {
        std::unique_lock<std::mutex> l(m);
        writer = false;
        cv.notify_all();
    }
};

struct cdb_shards {
    std::vector<int> devices;
    std::vector<cdb_index*> shard;                  // one handle per device slot
    std::vector<std::unique_ptr<MergeRank>> ranks;  // merge state of the shards in use
    std::shared_ptr<Transport> tr;
    int used = 0;                                   // shards holding documents after the last build
    std::vector<uint64_t> bounds;                   // docs [bounds[i], bounds[i+1]) live on shard i
    // host staging of the whole column (cdb_shards_add*): the ONE host copy — builds upload views of it (cdb_build_view)
    std::vector<int64_t> ids;
    std::vector<uint64_t> doc_start{0};
    std::string text;
    bool staging_valid = true;               // false after cdb_shards_load until the column is fetched back from the shards
    uint64_t max_shard_bytes = 12ull << 30;  // a shard beyond this is split although fewer devices would do
    bool use_all = false;                    // always spread over every device (bench / tests)
    bool device_merge = false;               // cdb_shards_query_batch merges on the devices (RCCL all-gatherv) instead of the host
    bool replace_in_place = false;           // build: destroy the serving shards FIRST (a column whose old + new arrays do not
                                             // fit together; the reference keeps both, database.cpp:276-280)
    std::vector<std::pair<std::string, int64_t>> options;  // guarded by opt_mu
    std::mutex opt_mu;
    StateLock state;
    std::mutex staging_mu;                   // add* / build read and write the staged column
    std::mutex err_mu;
    std::string err;
};

namespace {

// runs f(i) for i in [0, n) on n host threads (each shard has its own device, stream and locks); rethrows the first error
template <typename F>
void parallel_shards(int n, F&& f) {
    if (n == 1) {
        f(0);
        return;
    }
    std::vector<std::thread> th;
    std::mutex emu;
    std::string first;
    bool failed = false;
    for (int i = 0; i < n; ++i)
        th.emplace_back([&, i] {
            try {
                f(i);
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> g(emu);
                if (!failed) first = e.what();
                failed = true;
            }
        });
    for (auto& t : th) t.join();
    if (failed) throw Error(first);
}

void check_handle(cdb_index* p, int rc) {
    if (rc != CDB_OK) throw Error(cdb_last_error(p));
}
void check_shard(cdb_shards* h, int i, int rc) { check_handle(h->shard[i], rc); }

// merge ranks and their transport for the first `used` handles of `shard`
void make_merge(const std::vector<int>& devices, const std::vector<cdb_index*>& shard, int G,
                std::vector<std::unique_ptr<MergeRank>>& ranks, std::shared_ptr<Transport>& tr_out) {
    ranks.clear();
    tr_out.reset();
    if (G <= 1) return;
    bool distinct = true;
    for (int i = 0; i < G; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) distinct = false;
    const char* force = std::getenv("CDB_SHARD_TRANSPORT");
    const bool want_rccl = distinct && RcclApi::get().ok() && !(force && std::string(force) == "copy");
    if (want_rccl) {
        auto tr = std::make_shared<RcclTransport>(G);
        tr->comms.assign(G, nullptr);
        CDB_NCCL(RcclApi::get().CommInitAll(tr->comms.data(), G, devices.data()));
        for (int i = 0; i < G; ++i) tr->local_rank.push_back(i);
        tr_out = tr;
    } else {
        tr_out = std::make_shared<LocalTransport>(G);
    }
    for (int i = 0; i < G; ++i) {
        auto mr = std::make_unique<MergeRank>();
        mr->rank = i;
        mr->world = G;
        mr->device = devices[i];
        mr->stream = shard[i]->ix.stream;  // the shard's own stream: its query results are complete in stream order
        mr->tr = tr_out;
        ranks.push_back(std::move(mr));
    }
}

// fresh handles for every device slot, with the options set so far
std::vector<cdb_index*> fresh_handles(cdb_shards* h, size_t* options_seen = nullptr) {
    std::vector<cdb_index*> fresh;
    std::vector<std::pair<std::string, int64_t>> options;
    {
        std::lock_guard<std::mutex> g(h->opt_mu);
        options = h->options;  // snapshot: cdb_shards_set_option may append while a build runs (install() replays the rest)
    }
    if (options_seen) *options_seen = options.size();
    try {
        for (size_t i = 0; i < h->devices.size(); ++i) {
            cdb_index* p = nullptr;
            if (cdb_create(&p, h->devices[i]) != CDB_OK) throw Error("HIP error: cannot create a shard handle");
            fresh.push_back(p);
            for (auto& kv : options) (void)cdb_set_option(p, kv.first.c_str(), kv.second);
        }
    } catch (...) {
        for (cdb_index* p : fresh) cdb_destroy(p);
        throw;
    }
    return fresh;
}

// a new generation of shards takes over (exclusive lock held by the caller); the old handles are destroyed
void install(cdb_shards* h, std::vector<cdb_index*>& fresh, int used, std::vector<uint64_t>& bounds,
             std::vector<std::unique_ptr<MergeRank>>& ranks, std::shared_ptr<Transport>& tr, size_t options_seen = ~size_t(0)) {
    {   // options set while this generation was being built reached the OLD handles only: replay them on the new ones
        std::lock_guard<std::mutex> g(h->opt_mu);
        for (size_t k = options_seen; k < h->options.size(); ++k)
            for (cdb_index* p : fresh) (void)cdb_set_option(p, h->options[k].first.c_str(), h->options[k].second);
    }
    h->ranks.swap(ranks);   // (the old merge ranks borrow the old handles' streams: they go first)
    ranks.clear();
    h->tr.swap(tr);
    tr.reset();
    h->shard.swap(fresh);
    h->used = used;
    h->bounds.swap(bounds);
    for (cdb_index* p : fresh) cdb_destroy(p);
    fresh.clear();
}

// the column back on the host after cdb_shards_load (the shards hold it on their devices)
void fetch_staging(cdb_shards* h) {
    if (h->staging_valid) return;
    std::vector<int64_t> ids;
    std::vector<uint64_t> ds{0};
    std::string text;
    for (int i = 0; i < h->used; ++i) {
        Index& ix = h->shard[i]->ix;
        ensure_host_staging(ix);
        ids.insert(ids.end(), ix.ids.begin(), ix.ids.end());
        const uint64_t base = text.size();
        text.append(ix.host_text);
        for (size_t d = 1; d < ix.doc_start.size(); ++d) ds.push_back(base + ix.doc_start[d]);
        std::string().swap(ix.host_text);  // (one host copy: the shard keeps its device copy)
        ix.host_text_valid = false;
    }
    h->ids.swap(ids);
    h->doc_start.swap(ds);
    h->text.swap(text);
    h->staging_valid = true;
}

// f(t, j0, j1) over [0, n) split into contiguous ranges, on up to T host threads (small n: one)
template <typename F>
void parallel_ranges(uint64_t n, int T, F&& f) {
    T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)T, n >> 16));
    if (T == 1) {
        f(0, (uint64_t)0, n);
        return;
    }
    parallel_shards(T, [&](int t) { f(t, n * t / T, n * (t + 1) / T); });
}

int out_rows(const std::vector<std::pair<int64_t, int64_t>>& rows, int64_t** ids, int64_t** counts, size_t* nrows) {
    int64_t* oi = (int64_t*)std::malloc(std::max<size_t>(rows.size(), 1) * 8);
    int64_t* oc = (int64_t*)std::malloc(std::max<size_t>(rows.size(), 1) * 8);
    if (!oi || !oc) {
        std::free(oi);
        std::free(oc);
        throw std::bad_alloc();
    }
    for (size_t r = 0; r < rows.size(); ++r) {
        oi[r] = rows[r].first;
        oc[r] = rows[r].second;
    }
    *ids = oi;
    *counts = oc;
    *nrows = rows.size();
    return 0;
}

}  // namespace

extern "C" {

int cdb_shards_create(cdb_shards** out, const int* devices, int ndev) {
    if (!out || ndev < 1 || !devices) return CDB_E_INVALID;
    *out = nullptr;
    cdb_shards* h = new (std::nothrow) cdb_shards();
    if (!h) return CDB_E_DEVICE;
    for (int i = 0; i < ndev; ++i) {
        cdb_index* ix = nullptr;
        const int rc = cdb_create(&ix, devices[i]);
        if (rc != CDB_OK) {
            for (cdb_index* p : h->shard) cdb_destroy(p);
            delete h;
            return rc;
        }
        h->shard.push_back(ix);
        h->devices.push_back(ix->ix.device);
    }
    *out = h;
    return CDB_OK;
}

void cdb_shards_destroy(cdb_shards* h) {
    if (!h) return;
    h->ranks.clear();
    h->tr.reset();
    for (cdb_index* p : h->shard) cdb_destroy(p);
    delete h;
}

const char* cdb_shards_last_error(const cdb_shards* h) {
    if (!h) return "null handle";
    static thread_local std::string copy;
    std::lock_guard<std::mutex> g(const_cast<cdb_shards*>(h)->err_mu);
    copy = h->err;
    return copy.c_str();
}

int cdb_shards_add(cdb_shards* h, int64_t id, const char* value, size_t len) {
    if (!h || (!value && len)) return CDB_E_INVALID;
    return guarded_on(h, [&] {
        std::lock_guard<std::mutex> g(h->staging_mu);  // (lock order everywhere: staging_mu, then state)
        {
            std::shared_lock<StateLock> st(h->state);
            fetch_staging(h);
        }
        h->text.append(value, len);
        h->ids.push_back(id);
        h->doc_start.push_back(h->text.size());
    });
}

int cdb_shards_add_bulk(cdb_shards* h, const int64_t* ids, const char* blob, const uint64_t* doc_start, uint64_t ndocs) {
    if (!h || (ndocs && (!ids || !doc_start))) return CDB_E_INVALID;
    return guarded_on(h, [&] {
        if (!ndocs) return;
        for (uint64_t d = 0; d < ndocs; ++d)
            if (doc_start[d + 1] < doc_start[d]) throw Error("doc_start must be non-decreasing");
        std::lock_guard<std::mutex> g(h->staging_mu);
        {
            std::shared_lock<StateLock> st(h->state);
            fetch_staging(h);
        }
        h->ids.reserve(h->ids.size() + ndocs);
        h->doc_start.reserve(h->doc_start.size() + ndocs);
        const uint64_t base = h->text.size();
        h->text.append(blob + doc_start[0], doc_start[ndocs] - doc_start[0]);
        for (uint64_t d = 0; d < ndocs; ++d) {
            h->ids.push_back(ids[d]);
            h->doc_start.push_back(base + doc_start[d + 1] - doc_start[0]);
        }
    });
}

// raw-file ingest (database.cpp:170-275) into the sharded column: cdb_add_raw_dir's reader, all or nothing
int cdb_shards_add_raw_dir(cdb_shards* h, const char* dir, const char* key, uint64_t* records, uint64_t* added) {
    if (!h || !dir || !key) return CDB_E_INVALID;
    if (records) *records = 0;
    if (added) *added = 0;
    return guarded_on(h, [&] {
        std::lock_guard<std::mutex> g(h->staging_mu);
        {
            std::shared_lock<StateLock> st(h->state);
            fetch_staging(h);
        }
        uint64_t nrec = 0, nadd = 0;
        read_raw_dir(dir, key, h->ids, h->doc_start, h->text, nrec, nadd);
        if (records) *records = nrec;
        if (added) *added = nadd;
    });
}

int cdb_shards_set_option(cdb_shards* h, const char* name, int64_t value) {
    if (!h || !name) return CDB_E_INVALID;
    if (!std::strcmp(name, "max_shard_bytes")) {
        h->max_shard_bytes = value > 0 ? (uint64_t)value : 1;
        return CDB_OK;
    }
    if (!std::strcmp(name, "use_all_devices")) {
        h->use_all = value != 0;
        return CDB_OK;
    }
    if (!std::strcmp(name, "device_merge")) {
        h->device_merge = value != 0;
        return CDB_OK;
    }
    if (!std::strcmp(name, "replace_in_place")) {
        h->replace_in_place = value != 0;
        return CDB_OK;
    }
    std::shared_lock<StateLock> st(h->state);
    for (cdb_index* p : h->shard) {
        const int rc = cdb_set_option(p, name, value);
        if (rc != CDB_OK) {
            std::lock_guard<std::mutex> g(h->err_mu);
            h->err = cdb_last_error(p);
            return rc;
        }
    }
    {
        std::lock_guard<std::mutex> g(h->opt_mu);
        h->options.emplace_back(name, value);  // (replayed on the fresh handles of every rebuild)
    }
    return CDB_OK;
}

// index.cpp:178-236 over several GPUs.  Transactional: a NEW generation of shard handles is built beside
// the serving one — database.cpp:276-280 keeps the old index answering until the new one is complete — and takes over
// under the exclusive lock only when every shard succeeded; a failure leaves the old generation untouched.  The shards
// upload VIEWS of the one staged column (cdb_build_view): no second host copy per shard.
int cdb_shards_build(cdb_shards* h) {
    if (!h) return CDB_E_INVALID;
    return guarded_on(h, [&] {
        std::lock_guard<std::mutex> sg(h->staging_mu);  // (the column must not move while the shards upload it)
        {
            std::shared_lock<StateLock> st(h->state);
            fetch_staging(h);
        }
        const int G = (int)h->devices.size();
        const uint64_t nd = h->ids.size(), total = h->doc_start[nd];
        // shard only when the column exceeds what one GPU should hold (north star), unless told to spread anyway
        int used = h->use_all ? G : (int)std::min<uint64_t>((uint64_t)G, std::max<uint64_t>(1, ceil_div(total, h->max_shard_bytes)));
        used = std::max(1, std::min<int>(used, (int)std::max<uint64_t>(nd, 1)));
        std::vector<uint64_t> b = shard_bounds(h->doc_start, used);
        if (h->replace_in_place) {  // the caller accepts an unbuilt window: the old arrays go first
            std::unique_lock<StateLock> st(h->state);
            std::vector<cdb_index*> empty = fresh_handles(h);
            std::vector<uint64_t> nb{0};
            std::vector<std::unique_ptr<MergeRank>> nr;
            std::shared_ptr<Transport> nt;
            install(h, empty, 0, nb, nr, nt);
        }
        size_t opts_seen = 0;
        std::vector<cdb_index*> fresh = fresh_handles(h, &opts_seen);
        std::vector<std::unique_ptr<MergeRank>> ranks;
        std::shared_ptr<Transport> tr;
        try {
            parallel_shards(used, [&](int i) {
                const uint64_t d0 = b[i], d1 = b[i + 1];
                check_handle(fresh[i], cdb_build_view(fresh[i], h->ids.data() + d0, h->text.data(), h->doc_start.data() + d0, d1 - d0));
            });
            make_merge(h->devices, fresh, used, ranks, tr);
        } catch (...) {
            ranks.clear();
            tr.reset();
            for (cdb_index* p : fresh) cdb_destroy(p);
            throw;
        }
        std::unique_lock<StateLock> st(h->state);
        install(h, fresh, used, b, ranks, tr, opts_seen);
    });
}

// The same from the caller's own strings (string_index's views): nothing is staged in the handle, every shard gathers its
// documents straight into its pinned upload chunks.  A later cdb_shards_add fetches the column back from the shards.
int cdb_shards_build_views(cdb_shards* h, const int64_t* ids, const char* const* ptrs, const uint64_t* lens, uint64_t ndocs) {
    if (!h || (ndocs && (!ids || !ptrs || !lens))) return CDB_E_INVALID;
    return guarded_on(h, [&] {
        std::lock_guard<std::mutex> sg(h->staging_mu);
        std::vector<uint64_t> ds(ndocs + 1, 0);
        for (uint64_t d = 0; d < ndocs; ++d) ds[d + 1] = ds[d] + lens[d];
        const int G = (int)h->devices.size();
        const uint64_t total = ds[ndocs];
        int used = h->use_all ? G : (int)std::min<uint64_t>((uint64_t)G, std::max<uint64_t>(1, ceil_div(total, h->max_shard_bytes)));
        used = std::max(1, std::min<int>(used, (int)std::max<uint64_t>(ndocs, 1)));
        std::vector<uint64_t> b = shard_bounds(ds, used);
        size_t opts_seen = 0;
        std::vector<cdb_index*> fresh = fresh_handles(h, &opts_seen);
        std::vector<std::unique_ptr<MergeRank>> ranks;
        std::shared_ptr<Transport> tr;
        try {
            parallel_shards(used, [&](int i) {
                const uint64_t d0 = b[i], d1 = b[i + 1];
                check_handle(fresh[i], cdb_build_views(fresh[i], ids + d0, ptrs + d0, lens + d0, d1 - d0));
            });
            make_merge(h->devices, fresh, used, ranks, tr);
        } catch (...) {
            ranks.clear();
            tr.reset();
            for (cdb_index* p : fresh) cdb_destroy(p);
            throw;
        }
        std::unique_lock<StateLock> st(h->state);
        install(h, fresh, used, b, ranks, tr, opts_seen);
        h->ids.clear();
        h->doc_start.assign(1, 0);
        std::string().swap(h->text);
        h->staging_valid = false;  // (the column lives on the devices and with the caller)
    });
}

// f4 over shards: `path` holds the shard count and document bounds, `path.<i>` shard i's own file (cdb_save)
namespace {
constexpr uint64_t SHARDS_MAGIC = 0x3130485344424443ull;  // "CDBDSH01"
}
int cdb_shards_save(cdb_shards* h, const char* path) {
    if (!h || !path) return CDB_E_INVALID;
    return guarded_on(h, [&] {
        std::shared_lock<StateLock> st(h->state);
        if (h->used < 1) throw Error("index has not been built");
        for (int i = 0; i < h->used; ++i) check_shard(h, i, cdb_save(h->shard[i], (std::string(path) + "." + std::to_string(i)).c_str()));
        FILE* fp = std::fopen(path, "wb");
        if (!fp) throw Error(std::string("Cannot open file: ") + path);
        struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{fp};
        const uint64_t hd[2] = {SHARDS_MAGIC, (uint64_t)h->used};
        bool ok = std::fwrite(hd, 8, 2, fp) == 2;
        ok = ok && std::fwrite(h->bounds.data(), 8, (size_t)h->used + 1, fp) == (size_t)h->used + 1;
        if (!ok) throw Error(std::string("Cannot write file: ") + path);
    });
}

int cdb_shards_load(cdb_shards* h, const char* path) {
    if (!h || !path) return CDB_E_INVALID;
    return guarded_on(h, [&] {
        std::lock_guard<std::mutex> sg(h->staging_mu);
        uint64_t hd[2] = {0, 0};
        std::vector<uint64_t> b;
        {
            FILE* fp = std::fopen(path, "rb");
            if (!fp) throw Error(std::string("Cannot open file: ") + path);
            struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{fp};
            if (std::fread(hd, 8, 2, fp) != 2 || hd[0] != SHARDS_MAGIC || hd[1] < 1 || hd[1] > 4096)
                throw Error(std::string("Not a saved sharded index: ") + path);
            b.resize(hd[1] + 1);
            if (std::fread(b.data(), 8, b.size(), fp) != b.size()) throw Error(std::string("Truncated index file: ") + path);
        }
        const int used = (int)hd[1];
        if (used > (int)h->devices.size()) throw Error("saved index has more shards than this handle has devices");
        size_t opts_seen = 0;
        std::vector<cdb_index*> fresh = fresh_handles(h, &opts_seen);
        std::vector<std::unique_ptr<MergeRank>> ranks;
        std::shared_ptr<Transport> tr;
        try {
            parallel_shards(used, [&](int i) {
                check_handle(fresh[i], cdb_load(fresh[i], (std::string(path) + "." + std::to_string(i)).c_str()));
                if (fresh[i]->ix.ndocs != b[i + 1] - b[i]) throw Error(std::string("Corrupt index file (shard bounds): ") + path);
            });
            make_merge(h->devices, fresh, used, ranks, tr);
        } catch (...) {
            ranks.clear();
            tr.reset();
            for (cdb_index* p : fresh) cdb_destroy(p);
            throw;
        }
        std::unique_lock<StateLock> st(h->state);
        install(h, fresh, used, b, ranks, tr, opts_seen);
        h->ids.clear();
        h->doc_start.assign(1, 0);
        std::string().swap(h->text);
        h->staging_valid = false;  // the column lives on the devices (fetched back by the next add / build)
    });
}

int cdb_shards_count(const cdb_shards* h) { return h ? h->used : 0; }
cdb_index* cdb_shards_get(cdb_shards* h, int i) { return (h && i >= 0 && i < (int)h->shard.size()) ? h->shard[i] : nullptr; }
uint64_t cdb_shards_first_doc(const cdb_shards* h, int i) {
    return (h && i >= 0 && i < (int)h->bounds.size()) ? h->bounds[i] : 0;
}
const char* cdb_shards_transport(const cdb_shards* h) { return (h && h->tr) ? h->tr->name() : "none"; }

// One keyword (database.cpp:392): every shard's lone-keyword kernel is in flight before the first answer is awaited
// (one launch + one poll per shard instead of G complete round trips); shards in order = ascending document index.
int cdb_shards_query(cdb_shards* h, const char* keyword, size_t len, int64_t** ids, int64_t** counts, size_t* nrows) {
    if (!h || !ids || !counts || !nrows) return CDB_E_INVALID;
    *ids = nullptr;
    *counts = nullptr;
    *nrows = 0;
    return guarded_on(h, [&] {
        if (len == 0) throw Error("Empty keywords are not allowed");  // index.cpp:239-241
        std::shared_lock<StateLock> st(h->state);
        const int G = std::max(h->used, 1);
        struct Part {
            int64_t *ids = nullptr, *counts = nullptr;
            size_t n = 0;
            SingleLaunch state = SingleLaunch::NotApplicable;
        };
        std::vector<Part> part(G);
        struct Cleanup {
            std::vector<Part>& p;
            ~Cleanup() { for (auto& x : p) { cdb_free(x.ids); cdb_free(x.counts); } }
        } cleanup{part};
        {
            std::vector<std::unique_lock<std::mutex>> locks;  // ascending shard order
            std::vector<char> in_flight(G, 0);                // launched, answer not collected yet
            try {
                for (int i = 0; i < G; ++i) {
                    Index& ix = h->shard[i]->ix;
                    locks.emplace_back(ix.mu);
                    CDB_HIP(hipSetDevice(ix.device));
                    StreamScope ss(ix.stream);
                    part[i].state = query_single_launch(ix, keyword, len);
                    in_flight[i] = part[i].state == SingleLaunch::Launched;
                }
                for (int i = 0; i < G; ++i) {
                    Index& ix = h->shard[i]->ix;
                    if (part[i].state == SingleLaunch::Absent) query_single_empty(ix, &part[i].ids, &part[i].counts, &part[i].n);
                    else if (part[i].state == SingleLaunch::Launched) {
                        CDB_HIP(hipSetDevice(ix.device));
                        StreamScope ss(ix.stream);
                        in_flight[i] = 0;  // (collect waits for the answer or throws after the stream was synchronised)
                        if (!query_single_collect(ix, &part[i].ids, &part[i].counts, &part[i].n)) part[i].state = SingleLaunch::NotApplicable;
                    }
                }
            } catch (...) {
                // kernels of the shards already launched would answer LATER into their host-mapped result blocks — over the
                // next query's "not answered yet" mark.  They finish (or the resident workgroup leaves) before the locks go.
                for (int i = 0; i < (int)locks.size(); ++i) {
                    Index& ix = h->shard[i]->ix;
                    (void)hipSetDevice(ix.device);
                    if (ix.resident_query) query_resident_stop(ix);
                    else if (in_flight[i]) (void)hipStreamSynchronize(ix.stream);
                }
                throw;
            }
        }
        std::vector<std::pair<int64_t, int64_t>> rows;
        for (int i = 0; i < G; ++i) {
            if (part[i].state == SingleLaunch::NotApplicable)  // long hit lists, options: the shard's ordinary path
                check_shard(h, i, cdb_query(h->shard[i], keyword, len, &part[i].ids, &part[i].counts, &part[i].n));
            for (size_t r = 0; r < part[i].n; ++r) rows.emplace_back(part[i].ids[r], part[i].counts[r]);
        }
        out_rows(rows, ids, counts, nrows);
    });
}

// The per-key operations work shard by shard: documents — hence object ids — are disjoint across shards, so the union over
// shards is a concatenation, ordered afterwards the way the single-GPU call orders it.  The shards work concurrently.
namespace {
void shards_or_rows(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t nkw, bool ranked, int64_t lo, int64_t hi,
                    uint64_t limit, std::vector<std::pair<int64_t, int64_t>>& rows) {
    std::shared_lock<StateLock> st(h->state);
    const int G = std::max(h->used, 1);
    std::vector<std::vector<std::pair<int64_t, int64_t>>> part(G);
    parallel_shards(G, [&](int i) {
        int64_t *pi = nullptr, *pc = nullptr;
        size_t n = 0;
        // (every shard's own top `limit` rows contain its share of the global top `limit`)
        check_shard(h, i, ranked ? cdb_query_ranked(h->shard[i], blob, offsets, nkw, lo, hi, limit, &pi, &pc, &n)
                                 : cdb_query_or(h->shard[i], blob, offsets, nkw, &pi, &pc, &n));
        part[i].reserve(n);
        for (size_t r = 0; r < n; ++r) part[i].emplace_back(pi[r], pc[r]);
        cdb_free(pi);
        cdb_free(pc);
    });
    for (int i = 0; i < G; ++i) rows.insert(rows.end(), part[i].begin(), part[i].end());
    if (h->used > 1) {
        if (ranked) {
            std::sort(rows.begin(), rows.end(), [](const auto& a, const auto& b) { return a.second != b.second ? a.second > b.second : a.first < b.first; });
            if (limit && rows.size() > limit) rows.resize(limit);
        } else {
            std::sort(rows.begin(), rows.end());
        }
    }
}
int shards_or_impl(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t nkw, bool ranked, int64_t lo, int64_t hi,
                   uint64_t limit, int64_t** ids, int64_t** counts, size_t* nrows) {
    if (!h || !ids || !counts || !nrows || (nkw && !offsets)) return CDB_E_INVALID;
    *ids = nullptr;
    *counts = nullptr;
    *nrows = 0;
    return guarded_on(h, [&] {
        std::vector<std::pair<int64_t, int64_t>> rows;
        shards_or_rows(h, blob, offsets, nkw, ranked, lo, hi, limit, rows);
        out_rows(rows, ids, counts, nrows);
    });
}
}  // namespace

int cdb_shards_query_or(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t nkw, int64_t** ids, int64_t** counts,
                        size_t* nrows) {
    return shards_or_impl(h, blob, offsets, nkw, false, 0, 0, 0, ids, counts, nrows);
}

int cdb_shards_query_ranked(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t nkw, int64_t corr_lo, int64_t corr_hi,
                            uint64_t limit, int64_t** ids, int64_t** counts, size_t* nrows) {
    return shards_or_impl(h, blob, offsets, nkw, true, corr_lo, corr_hi, limit, ids, counts, nrows);
}

// AND across keys (interface.cpp:114-146) when string keys are sharded columns.  Different columns are cut at different
// documents (the cut balances bytes), so the intersection cannot be taken shard by shard: every sharded key is resolved
// with its OR over its own shards (rows ascending by id), then all row lists meet in ONE device merge — the same
// and_merge_on_device as cdb_query_and — on the first shard of the first sharded key.
int cdb_shards_query_and(const cdb_shards_key_query* keys, int nkeys, int ranked, int64_t corr_lo, int64_t corr_hi, uint64_t limit,
                         int64_t** ids, int64_t** counts, size_t* nrows) {
    if (!keys || nkeys < 1 || !ids || !counts || !nrows) return CDB_E_INVALID;
    *ids = nullptr;
    *counts = nullptr;
    *nrows = 0;
    cdb_shards* lead = nullptr;
    for (int k = 0; k < nkeys; ++k)
        if (keys[k].shards && !lead) lead = keys[k].shards;
    if (!lead) return CDB_E_INVALID;
    return guarded_on(lead, [&] {
        std::vector<std::vector<int64_t>> hi(nkeys), hc(nkeys);
        std::vector<cdb_key_query> flat(nkeys);
        for (int k = 0; k < nkeys; ++k) {
            const cdb_shards_key_query& q = keys[k];
            std::memset(&flat[k], 0, sizeof(cdb_key_query));
            if (q.shards) {
                if (q.nkw == 0) throw Error("The constraint list cannot be empty");  // interface.cpp:75-77
                if (!q.offsets) throw Error("cdb_shards_query_and: keyword offsets missing");
                for (uint64_t j = 0; j < q.nkw; ++j)
                    if (q.offsets[j + 1] <= q.offsets[j]) throw Error("Empty keywords are not allowed");
                std::vector<std::pair<int64_t, int64_t>> rows;
                shards_or_rows(q.shards, q.blob, q.offsets, q.nkw, false, 0, 0, 0, rows);
                hi[k].reserve(rows.size());
                hc[k].reserve(rows.size());
                for (auto& r : rows) {
                    hi[k].push_back(r.first);
                    hc[k].push_back(r.second);
                }
                flat[k].ids = hi[k].data();
                flat[k].counts = hc[k].data();
                flat[k].nrows = hi[k].size();
            } else {
                if (q.nrows && (!q.ids || !q.counts)) throw Error("cdb_shards_query_and: row list missing");
                flat[k].ids = q.ids;
                flat[k].counts = q.counts;
                flat[k].nrows = q.nrows;
            }
        }
        std::shared_lock<StateLock> st(lead->state);
        cdb_index* dev = lead->shard[0];
        check_handle(dev, query_and_with_lead(dev, flat.data(), nkeys, ranked, corr_lo, corr_hi, limit, ids, counts, nrows));
    });
}

int cdb_shards_query_spans(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t nkw, cdb_spans* out) {
    if (!h || !out || (nkw && !offsets)) return CDB_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    const int rc = guarded_on(h, [&] {
        std::shared_lock<StateLock> st(h->state);
        std::vector<cdb_spans> parts(std::max(h->used, 1));
        for (auto& x : parts) std::memset(&x, 0, sizeof(cdb_spans));
        struct Cleanup {
            std::vector<cdb_spans>& p;
            ~Cleanup() { for (auto& x : p) cdb_spans_free(&x); }
        } cleanup{parts};
        parallel_shards((int)parts.size(), [&](int i) { check_shard(h, i, cdb_query_spans(h->shard[i], blob, offsets, nkw, &parts[i])); });
        uint64_t nd = 0, ns = 0;
        for (const cdb_spans& p : parts) {  // shard order = ascending document index
            nd += p.ndocs;
            ns += p.nspans;
        }
        out->ndocs = nd;
        out->nspans = ns;
        out->ids = (int64_t*)host_alloc(nd * 8);
        out->span_ptr = (uint64_t*)host_alloc((nd + 1) * 8, true);
        out->begin = (uint64_t*)host_alloc(ns * 8);
        out->end = (uint64_t*)host_alloc(ns * 8);
        uint64_t d0 = 0, s0 = 0;
        for (const cdb_spans& p : parts) {
            for (uint64_t d = 0; d < p.ndocs; ++d) {
                out->ids[d0 + d] = p.ids[d];
                out->span_ptr[d0 + d] = s0 + p.span_ptr[d];
            }
            if (p.nspans) {
                std::memcpy(out->begin + s0, p.begin, p.nspans * 8);
                std::memcpy(out->end + s0, p.end, p.nspans * 8);
            }
            d0 += p.ndocs;
            s0 += p.nspans;
        }
        out->span_ptr[nd] = ns;
    });
    if (rc != CDB_OK) cdb_spans_free(out);
    return rc;
}

namespace {
// Host merge of the shards' CSR answers (the consumer of the C ABI is the host: SURVEY §8e — every GPU hands over its own
// slice, nothing is replicated on the devices).  Every shard answers the whole batch through its own host entry point
// (patterns up and rows down over its own PCIe link, concurrently); the rows of a pattern are then the shards' rows in
// shard order.  The interleave runs over pattern ranges on several host threads.
void shards_batch_host(cdb_shards* h, int G, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out, cdb_hits* hits) {
    std::vector<cdb_result> part(G);
    std::vector<cdb_hits> hpart(G);
    for (int i = 0; i < G; ++i) {
        std::memset(&part[i], 0, sizeof(cdb_result));
        std::memset(&hpart[i], 0, sizeof(cdb_hits));
    }
    struct Cleanup {
        std::vector<cdb_result>& p;
        std::vector<cdb_hits>& q;
        ~Cleanup() {
            for (auto& x : p) cdb_result_free(&x);
            for (auto& x : q) cdb_hits_free(&x);
        }
    } cleanup{part, hpart};
    parallel_shards(G, [&](int i) {
        check_shard(h, i, hits ? cdb_query_batch_offsets(h->shard[i], blob, offsets, npat, &part[i], &hpart[i])
                               : cdb_query_batch(h->shard[i], blob, offsets, npat, &part[i]));
    });
    uint64_t nrows = 0, nhits = 0;
    for (int i = 0; i < G; ++i) {
        nrows += part[i].nrows;
        nhits += part[i].nhits;
    }
    out->npat = npat;
    out->nrows = nrows;
    out->nhits = nhits;
    out->row_ptr = (uint64_t*)host_alloc((npat + 1) * 8);
    out->ids = (int64_t*)host_alloc(nrows * 8);
    out->counts = (int64_t*)host_alloc(nrows * 8);
    if (hits) {
        hits->hit_ptr = (uint64_t*)host_alloc((nrows + 1) * 8);
        hits->offsets = (uint64_t*)host_alloc(nhits * 8);
    }
    // first row (and first hit) of every pattern range = the rows (hits) the shards hold in front of it
    const int T = std::max(G, 4);
    parallel_ranges(npat, T, [&](int, uint64_t j0, uint64_t j1) {
        uint64_t row = 0, hit = 0;
        for (int i = 0; i < G; ++i) {
            row += part[i].row_ptr[j0];
            if (hits) hit += hpart[i].hit_ptr[part[i].row_ptr[j0]];
        }
        for (uint64_t j = j0; j < j1; ++j) {
            out->row_ptr[j] = row;
            for (int i = 0; i < G; ++i) {
                const uint64_t a = part[i].row_ptr[j], b = part[i].row_ptr[j + 1];
                if (a == b) continue;
                std::memcpy(out->ids + row, part[i].ids + a, (b - a) * 8);
                std::memcpy(out->counts + row, part[i].counts + a, (b - a) * 8);
                if (hits) {
                    const uint64_t ha = hpart[i].hit_ptr[a], hb = hpart[i].hit_ptr[b];
                    for (uint64_t r = a; r < b; ++r) hits->hit_ptr[row + (r - a)] = hit + (hpart[i].hit_ptr[r] - ha);
                    std::memcpy(hits->offsets + hit, hpart[i].offsets + ha, (hb - ha) * 8);
                    hit += hb - ha;
                }
                row += b - a;
            }
        }
    });
    out->row_ptr[npat] = nrows;
    if (hits) hits->hit_ptr[nrows] = nhits;
}

// ... or merged on the devices (option device_merge: the RCCL all-gatherv path, for consumers that keep the rows in HBM;
// every shard ends up holding the merged CSR, shard 0 hands it to the host)
void shards_batch_device(cdb_shards* h, int G, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out) {
    const uint64_t base = npat ? offsets[0] : 0, nbytes = npat ? offsets[npat] - base : 0;
    std::vector<uint64_t> rel(npat + 1);
    for (uint64_t j = 0; j <= npat; ++j) rel[j] = npat ? offsets[j] - base : 0;
    std::vector<cdb_device_result> local(G), merged(G);
    // every shard stays locked from its query to the end of the merge (the merge reads the shard's result
    // buffers, which a concurrent query on that shard would overwrite); ascending shard order
    std::vector<std::unique_lock<std::mutex>> locks;
    for (int i = 0; i < G; ++i) locks.emplace_back(h->shard[i]->ix.mu);
    parallel_shards(G, [&](int i) {
        Index& ix = h->shard[i]->ix;
        CDB_HIP(hipSetDevice(ix.device));
        StreamScope ss(ix.stream);
        hipStream_t s = ix.stream;
        ix.q_pat.ensure(nbytes + 16);
        ix.q_offs.ensure((npat + 1) * 8);
        if (nbytes) CDB_HIP(hipMemcpyAsync(ix.q_pat.p, blob + base, nbytes, hipMemcpyHostToDevice, s));
        CDB_HIP(hipMemcpyAsync(ix.q_offs.p, rel.data(), (npat + 1) * 8, hipMemcpyHostToDevice, s));
        const DeviceCsr r = query_batch_on_device(ix, ix.q_pat.as<uint8_t>(), ix.q_offs.as<uint64_t>(), npat);
        local[i] = cdb_device_result{npat, r.nrows, r.nhits, ix.q_rowptr.as<uint64_t>(), ix.q_ids.as<int64_t>(), ix.q_counts.as<int64_t>()};
    });
    uint64_t hits = 0;
    for (int i = 0; i < G; ++i) hits += local[i].nhits;
    // the collective merge is a phase of its own: a shard that failed above never leaves the others waiting
    parallel_shards(G, [&](int i) { merge_core(*h->ranks[i], local[i], merged[i]); });
    Index& ix0 = h->shard[0]->ix;
    CDB_HIP(hipSetDevice(ix0.device));
    hipStream_t s = ix0.stream;
    const cdb_device_result& m = merged[0];
    out->npat = npat;
    out->nrows = m.nrows;
    out->nhits = hits;
    out->row_ptr = (uint64_t*)host_alloc((npat + 1) * 8);
    out->ids = (int64_t*)host_alloc(m.nrows * 8);
    out->counts = (int64_t*)host_alloc(m.nrows * 8);
    CDB_HIP(hipMemcpyAsync(out->row_ptr, m.d_row_ptr, (npat + 1) * 8, hipMemcpyDeviceToHost, s));
    if (m.nrows) {
        CDB_HIP(hipMemcpyAsync(out->ids, m.d_ids, m.nrows * 8, hipMemcpyDeviceToHost, s));
        CDB_HIP(hipMemcpyAsync(out->counts, m.d_counts, m.nrows * 8, hipMemcpyDeviceToHost, s));
    }
    CDB_HIP(hipStreamSynchronize(s));
}

int shards_batch_impl(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out, cdb_hits* hits) {
    if (!h || !out || (npat && !offsets)) return CDB_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    if (hits) std::memset(hits, 0, sizeof(*hits));
    const int rc = guarded_on(h, [&] {
        for (uint64_t j = 0; j < npat; ++j)
            if (offsets[j + 1] <= offsets[j]) throw Error("Empty keywords are not allowed");  // index.cpp:239-241
        std::shared_lock<StateLock> st(h->state);
        const int G = std::max(h->used, 1);
        if (G == 1) {
            check_shard(h, 0, hits ? cdb_query_batch_offsets(h->shard[0], blob, offsets, npat, out, hits)
                                   : cdb_query_batch(h->shard[0], blob, offsets, npat, out));
            return;
        }
        if (h->device_merge && !hits) shards_batch_device(h, G, blob, offsets, npat, out);
        else shards_batch_host(h, G, blob, offsets, npat, out, hits);
    });
    if (rc != CDB_OK) {
        cdb_result_free(out);
        if (hits) cdb_hits_free(hits);
    }
    return rc;
}
}  // namespace

int cdb_shards_query_batch(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out) {
    return shards_batch_impl(h, blob, offsets, npat, out, nullptr);
}

int cdb_shards_query_batch_offsets(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out,
                                   cdb_hits* hits) {
    if (!hits) return CDB_E_INVALID;
    return shards_batch_impl(h, blob, offsets, npat, out, hits);
}

}  // extern "C"
