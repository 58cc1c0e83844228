// verify.hip — size-independent checks of a built suffix array, computed on the GPU by code that shares
// nothing with the build (plain adjacent-suffix comparison).  Test hook behind cdb_debug_verify: lets
// the full-size configurations (1 GiB and up, where no CPU oracle finishes) assert
//   * sortedness        : suffix(sa[i-1]) <= suffix(sa[i]) in unsigned byte order, shorter first
//   * canonical ties     : equal suffixes ascend by document index (SURVEY.md Q1 canonical form)
//   * permutation        : every entry is a valid (doc, off) and the wrapped sum of all entries equals
//                          the closed-form sum over all (doc, off) pairs — with strict tie order this
//                          rules out duplicates and omissions.
#include <chrono>
#include <thread>

#include "index_impl.h"

namespace cdb {
namespace {
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename V>
__global__ __launch_bounds__(256) void sa_verify_kernel(typename SaOf<V>::ptr sa, uint64_t n,
                                                        const uint8_t* __restrict__ text,
                                                        const uint64_t* __restrict__ doc_start, uint64_t ndocs,
                                                        int bits, uint64_t mask, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long s_acc[4];
    if (threadIdx.x < 4) s_acc[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long inv = 0, tie = 0, sum = 0, bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;  // grid-stride: n may exceed 2^32 threads
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const auto eb = sa[i];
        const uint64_t db = (uint64_t)eb & mask, ob = (uint64_t)eb >> bits;
        sum += (unsigned long long)eb;
        if (db >= ndocs || ob >= doc_start[db + 1] - doc_start[db]) {
            bad += 1;
        } else if (i > 0) {
            const auto ea = sa[i - 1];
            const uint64_t da = (uint64_t)ea & mask, oa = (uint64_t)ea >> bits;
            if (da < ndocs && oa < doc_start[da + 1] - doc_start[da]) {
                const uint8_t* pa = text + doc_start[da] + oa;
                const uint8_t* pb = text + doc_start[db] + ob;
                const uint64_t la = doc_start[da + 1] - doc_start[da] - oa, lb = doc_start[db + 1] - doc_start[db] - ob;
                const uint64_t len = la < lb ? la : lb;
                int c = 0;
                for (uint64_t k = 0; k < len; ++k) {
                    const uint8_t x = pa[k], y = pb[k];
                    if (x != y) {
                        c = x < y ? -1 : 1;
                        break;
                    }
                }
                if (c == 0) c = la < lb ? -1 : (la > lb ? 1 : 0);
                if (c > 0) inv += 1;
                if (c == 0 && da >= db) tie += 1;
            }
        }
    }
    if (inv) atomicAdd(&s_acc[0], inv);
    if (tie) atomicAdd(&s_acc[1], tie);
    atomicAdd(&s_acc[2], sum);
    if (bad) atomicAdd(&s_acc[3], bad);
    __syncthreads();
    if (threadIdx.x < 4 && s_acc[threadIdx.x]) atomicAdd(&out[threadIdx.x], s_acc[threadIdx.x]);
}

template <typename V>
__global__ __launch_bounds__(256) void sa_entry_check_kernel(typename SaOf<V>::ptr sa, uint64_t n,
                                                             const uint64_t* __restrict__ doc_start, uint64_t ndocs, int bits,
                                                             uint64_t mask, unsigned long long* __restrict__ bad) {
    unsigned long long b = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const auto e = sa[i];
        const uint64_t d = (uint64_t)e & mask, o = (uint64_t)e >> bits;
        if (d >= ndocs || o >= doc_start[d + 1] - doc_start[d]) b += 1;
    }
    if (b) atomicAdd(bad, b);
}

// Spot check behind every build (option self_check, default on): `samples` pseudo-random adjacent pairs of the finished
// array are compared the plain way.  A stable LSD sort whose ranking went wrong anywhere leaves inversions scattered
// over the whole array, so a few ten thousand pairs notice what the per-device self-test of the one-atomic ranking
// (radix_sort.h) could miss; the cost is a few microseconds.  Rules as in sa_verify_reference_kernel, minus the one
// that needs bucket sizes: equal suffixes ascend by document; a suffix that is a prefix of its neighbour comes first;
// first differing bytes of the same sign class ascend; a byte >= 0x80 against one < 0x80 must ascend only when the array
// is in plain unsigned order (`plain`), otherwise that pair is skipped (the reference's order depends on bucket sizes).
template <typename V>
__global__ __launch_bounds__(256) void sa_spot_check_kernel(typename SaOf<V>::ptr sa, uint64_t n,
                                                            const uint8_t* __restrict__ text,
                                                            const uint64_t* __restrict__ doc_start, uint64_t ndocs, int bits,
                                                            uint64_t mask, uint32_t samples, uint64_t seed, bool plain,
                                                            unsigned long long* __restrict__ out) {
    // samples == 0: the FULL sweep (option self_check = 2) — every adjacent pair, grid-stride; a proof instead of a sample
    if (n < 2) return;
    const uint64_t t0 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t stride = samples ? ~0ull : (uint64_t)gridDim.x * 256;
    unsigned long long nbad = 0, ninvalid = 0;
    for (uint64_t t = t0; t < (samples ? (uint64_t)samples : n - 1); t += stride) {
    uint64_t i = t + 1;
    if (samples) {
        uint64_t h = seed + 0x9E3779B97F4A7C15ull * (t + 1);
        h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
        h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
        h ^= h >> 31;
        i = 1 + h % (n - 1);
    }
    const auto ea = sa[i - 1], eb = sa[i];
    const uint64_t da = (uint64_t)ea & mask, oa = (uint64_t)ea >> bits, db = (uint64_t)eb & mask, ob = (uint64_t)eb >> bits;
    if (da >= ndocs || db >= ndocs || oa >= doc_start[da + 1] - doc_start[da] || ob >= doc_start[db + 1] - doc_start[db]) {
        ninvalid += 1;
        if (stride == ~0ull) break;
        continue;
    }
    const uint8_t* pa = text + doc_start[da] + oa;
    const uint8_t* pb = text + doc_start[db] + ob;
    const uint64_t la = doc_start[da + 1] - doc_start[da] - oa, lb = doc_start[db + 1] - doc_start[db] - ob;
    const uint64_t len = la < lb ? la : lb;
    const uint64_t cap = len < 4096 ? len : 4096;  // (long common prefixes — duplicate documents — are taken on trust)
    uint64_t l = 0;
    while (l < cap && pa[l] == pb[l]) ++l;
    bool bad = false;
    if (l == len) bad = la > lb || (la == lb && da >= db);
    else if (l < cap) bad = pa[l] > pb[l] && (plain || ((pa[l] ^ pb[l]) & 0x80u) == 0);
    if (bad) nbad += 1;
    if (stride == ~0ull) break;  // (sampled form: one pair per thread)
    }
    if (nbad) atomicAdd(&out[0], nbad);
    if (ninvalid) atomicAdd(&out[1], ninvalid);
}

// ---- the REFERENCE's order --------------------------------------------------------------
// For text with bytes >= 0x80 the reference's array is not globally sorted: a radix node (bucket of more than
// chuck_size = max(4096, n / 256) suffixes, index.cpp:96-126,218) lays its children out by
// `(int)char - CHAR_MIN + 1` of a SIGNED char (index.h:66-73): [end of document][0x80..0xFF][0x00..0x7F]; buckets of
// at most chuck_size suffixes are std::sort-ed by unsigned string_view order (index.cpp:86-95).  Checked here pair
// by pair, by code that shares nothing with apply_reference_order: for adjacent entries with common prefix length l,
//   * both end at l                      -> equal suffixes: ascending document (canonical tie order);
//   * exactly one ends at l              -> it comes first (either order);
//   * next bytes of the same sign class  -> ascending (signed and unsigned order agree);
//   * one byte >= 0x80, the other < 0x80 -> the bucket of their common l-prefix decides: more than chuck_size
//     suffixes share that prefix <=> it was a radix node <=> the byte >= 0x80 comes first; otherwise the byte < 0x80.
// Buckets are contiguous in the reference's order (its permutation only moves whole child ranges), so the size
// of the l-prefix bucket is found by galloping outwards from the pair until the prefix no longer matches.
template <typename V>
struct RefOrderCtx {
    typename SaOf<V>::ptr sa;
    uint64_t n;
    const uint8_t* text;
    const uint64_t* doc_start;
    int bits;
    uint64_t mask;
    uint64_t ndocs = ~0ull;  // (set by callers that may meet entries naming no document: such an entry is an empty suffix here)
    __device__ __forceinline__ void suffix(uint64_t i, const uint8_t*& p, uint64_t& len) const {
        const auto e = sa[i];
        const uint64_t d = (uint64_t)e & mask, o = (uint64_t)e >> bits;
        if (d >= ndocs) {
            p = text;
            len = 0;
            return;
        }
        const uint64_t ds = doc_start[d], dl = doc_start[d + 1] - ds;
        p = text + ds + (o < dl ? o : dl);
        len = o < dl ? dl - o : 0;
    }
    // does entry i share the first l bytes of q (a suffix known to have >= l bytes)?
    __device__ __forceinline__ bool shares(uint64_t i, const uint8_t* q, uint64_t l) const {
        const uint8_t* p;
        uint64_t len;
        suffix(i, p, len);
        if (len < l) return false;
        for (uint64_t k = 0; k < l; ++k)
            if (p[k] != q[k]) return false;
        return true;
    }
};

// is the bucket of the l-prefix that entries i - 1 and i share a radix node of the reference (more than `chuck` suffixes)?  pa =
// suffix of entry i - 1 (>= l bytes).  Counted outwards from the pair, capped at chuck + 1.
template <typename V>
__device__ __forceinline__ bool ref_bucket_is_node(const RefOrderCtx<V>& c, uint64_t i, const uint8_t* pa, uint64_t l, uint64_t chuck) {
    uint64_t cnt = 2;
    {   // backwards from i - 1
        uint64_t good = 0, step = 1, lim_b = i - 1;  // entries i-1-good .. i-1 share the prefix
        uint64_t badp = lim_b + 1;                   // first distance known not to share (or beyond the array)
        while (good + step <= lim_b && good + step <= chuck) {
            if (c.shares(i - 1 - (good + step), pa, l)) { good += step; step <<= 1; }
            else { badp = good + step; break; }
        }
        if (badp > good + step && good + step > lim_b) badp = lim_b + 1;
        if (good < chuck) {
            uint64_t hi = badp < chuck + 1 ? badp : chuck + 1;  // answer in [good, hi)
            while (good + 1 < hi) {
                const uint64_t mid = good + (hi - good) / 2;
                if (mid <= lim_b && c.shares(i - 1 - mid, pa, l)) good = mid; else hi = mid;
            }
        }
        cnt += good;
    }
    if (cnt <= chuck) {  // forwards from i
        uint64_t good = 0, step = 1, lim_f = c.n - 1 - i;
        uint64_t badp = lim_f + 1;
        while (good + step <= lim_f && good + step <= chuck) {
            if (c.shares(i + good + step, pa, l)) { good += step; step <<= 1; }
            else { badp = good + step; break; }
        }
        if (good < chuck) {
            uint64_t hi = badp < chuck + 1 ? badp : chuck + 1;
            while (good + 1 < hi) {
                const uint64_t mid = good + (hi - good) / 2;
                if (mid <= lim_f && c.shares(i + mid, pa, l)) good = mid; else hi = mid;
            }
        }
        cnt += good;
    }
    return cnt > chuck;
}

// The FULL sweep (self_check = 2) by the same rules, laid out for throughput: every lane fetches ONE suffix — its entry, the
// document bounds and the first 16 bytes as two big-endian words — and gets its left neighbour's through a lane shuffle, so a
// pair costs one random 64-byte sector instead of two and is decided by two 64-bit compares; only pairs that agree on
// 16 bytes walk on byte by byte (and the first lane of a wavefront fetches its neighbour itself).
struct SfxHead {
    uint64_t w0, w1;   // first 16 bytes, big-endian, bytes behind the suffix's end are 0
    uint64_t len, doc, pos;
    uint32_t ok;
};
template <typename V>
__device__ __forceinline__ SfxHead sfx_head(typename SaOf<V>::ptr sa, uint64_t i, uint64_t n, const uint8_t* __restrict__ text,
                                            const uint64_t* __restrict__ doc_start, uint64_t ndocs, int bits, uint64_t mask) {
    SfxHead h{0, 0, 0, 0, 0, 0u};
    const auto e = sa[i];
    const uint64_t d = (uint64_t)e & mask, o = (uint64_t)e >> bits;
    if (d >= ndocs) return h;
    const uint64_t ds = doc_start[d], de = doc_start[d + 1];
    if (o >= de - ds) return h;
    h.ok = 1u;
    h.doc = d;
    h.pos = ds + o;
    h.len = de - h.pos;
    typedef uint64_t __attribute__((aligned(1))) u64u;
    uint64_t a = 0, b = 0;
    if (h.pos + 16 <= n) {
        a = __builtin_bswap64(*reinterpret_cast<const u64u*>(text + h.pos));
        b = __builtin_bswap64(*reinterpret_cast<const u64u*>(text + h.pos + 8));
    } else {
        for (uint64_t k = 0; k < 16 && h.pos + k < n; ++k) {
            const uint64_t c = text[h.pos + k];
            if (k < 8) a |= c << (56 - 8 * k); else b |= c << (56 - 8 * (k - 8));
        }
    }
    if (h.len < 16) {  // bytes behind the end of the suffix (the next document's) do not count
        if (h.len <= 8) {
            b = 0;
            a = h.len == 8 ? a : (a & ~(~0ull >> (8 * h.len)));
        } else {
            b &= ~(~0ull >> (8 * (h.len - 8)));
        }
    }
    h.w0 = a;
    h.w1 = b;
    return h;
}
template <typename V>
__global__ __launch_bounds__(256) void sa_full_check_kernel(typename SaOf<V>::ptr sa, uint64_t n, const uint8_t* __restrict__ text,
                                                            const uint64_t* __restrict__ doc_start, uint64_t ndocs, int bits,
                                                            uint64_t mask, bool plain, unsigned long long* __restrict__ out,
                                                            uint64_t first, uint64_t end,
                                                            unsigned long long* __restrict__ out_mixed = nullptr, uint64_t chuck = 0) {
    // Reference-compat order of text with bytes >= 0x80 (plain = false): a pair whose first differing bytes lie on different sides of
    // 0x80 ("mixed") is right in EITHER order, depending on the size of the bucket the two suffixes share (rules above RefOrderCtx).
    // chuck = 0: such pairs pass (the sample behind a build); chuck = the reference's chuck_size: the lane that meets one finds the
    // bucket size by galloping over the array and judges it (the order proof).  out_mixed (optional) counts them.
    // entries [first, end) with the pair (first - 1, first) included: slices of one sweep add up to every adjacent pair
    const int lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    unsigned long long nbad = 0, ninvalid = 0, nskip = 0;
    for (uint64_t base = first + (uint64_t)blockIdx.x * 256; base < end; base += stride) {  // (uniform trip count: the shuffles need every lane)
        const uint64_t i = base + threadIdx.x;
        const bool valid = i < end;
        SfxHead b = valid ? sfx_head<V>(sa, i, n, text, doc_start, ndocs, bits, mask) : SfxHead{0, 0, 0, 0, 0, 0u};
        if (valid && !b.ok) ninvalid += 1;
        SfxHead a;
        a.w0 = __shfl_up(b.w0, 1);
        a.w1 = __shfl_up(b.w1, 1);
        a.len = __shfl_up(b.len, 1);
        a.doc = __shfl_up(b.doc, 1);
        a.pos = __shfl_up(b.pos, 1);
        a.ok = __shfl_up(b.ok, 1);
        if (lane == 0 && valid && i > 0) a = sfx_head<V>(sa, i - 1, n, text, doc_start, ndocs, bits, mask);
        if (!valid || i == 0 || !a.ok || !b.ok) continue;
        const uint64_t len = a.len < b.len ? a.len : b.len;
        bool bad = false;
        if (a.w0 != b.w0 || a.w1 != b.w1) {
            // first differing byte (inside min(len, 16) bytes unless one suffix ends first: its padding is 0 there)
            const uint64_t x = a.w0 != b.w0 ? a.w0 : a.w1, y = a.w0 != b.w0 ? b.w0 : b.w1;
            const uint32_t byte = (uint32_t)__builtin_clzll(x ^ y) >> 3;
            const uint64_t l = (a.w0 != b.w0 ? 0u : 8u) + byte;
            if (l >= len) {  // one is a prefix of the other: the shorter one must come first
                bad = a.len > b.len;
            } else {
                const uint32_t ca = (uint32_t)(x >> (56 - 8 * byte)) & 0xFFu, cb = (uint32_t)(y >> (56 - 8 * byte)) & 0xFFu;
                bad = ca > cb && (plain || ((ca ^ cb) & 0x80u) == 0);
                if (!plain && ((ca ^ cb) & 0x80u)) {
                    nskip += 1;
                    if (chuck) {
                        const RefOrderCtx<V> c{sa, n, text, doc_start, bits, mask, ndocs};
                        bad = (ca >= 0x80u) != ref_bucket_is_node(c, i, text + a.pos, l, chuck);
                    }
                }
            }
        } else if (len <= 16) {  // equal through the end of the shorter one
            bad = a.len > b.len || (a.len == b.len && a.doc >= b.doc);
        } else {  // 16 equal bytes: walk on (long common prefixes — duplicate documents — are taken on trust beyond 4096)
            const uint8_t* pa = text + a.pos;
            const uint8_t* pb = text + b.pos;
            const uint64_t cap = len < 4096 ? len : 4096;
            uint64_t l = 16;
            while (l < cap && pa[l] == pb[l]) ++l;
            if (l == len) bad = a.len > b.len || (a.len == b.len && a.doc >= b.doc);
            else if (l < cap) {
                bad = pa[l] > pb[l] && (plain || ((pa[l] ^ pb[l]) & 0x80u) == 0);
                if (!plain && ((pa[l] ^ pb[l]) & 0x80u)) {
                    nskip += 1;
                    if (chuck) {
                        const RefOrderCtx<V> c{sa, n, text, doc_start, bits, mask, ndocs};
                        bad = (pa[l] >= 0x80u) != ref_bucket_is_node(c, i, pa, l, chuck);
                    }
                }
            }
        }
        if (bad) nbad += 1;
    }
    if (nbad) atomicAdd(&out[0], nbad);
    if (ninvalid) atomicAdd(&out[1], ninvalid);
    if (nskip && out_mixed) atomicAdd(out_mixed, nskip);
}

template <typename V>
__global__ __launch_bounds__(256) void sa_verify_reference_kernel(RefOrderCtx<V> c, uint64_t chuck, unsigned long long* __restrict__ out,
                                                                  uint64_t first = 0, uint64_t end = ~0ull) {
    // pairs (i - 1, i) for i in [max(first, 1), min(end, n)): slices of one sweep add up to every adjacent pair
    unsigned long long bad = 0, mixed = 0, big = 0, tie = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    if (end > c.n) end = c.n;
    for (uint64_t i = first + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < end; i += stride) {
        if (i == 0) continue;
        const uint8_t *pa, *pb;
        uint64_t la, lb;
        c.suffix(i - 1, pa, la);
        c.suffix(i, pb, lb);
        const uint64_t lim = la < lb ? la : lb;
        uint64_t l = 0;
        while (l < lim && pa[l] == pb[l]) ++l;
        if (l == la && l == lb) {  // equal suffixes
            if (((uint64_t)c.sa[i - 1] & c.mask) >= ((uint64_t)c.sa[i] & c.mask)) tie += 1;
            continue;
        }
        if (l == la) continue;          // the shorter one first: right in both orders
        if (l == lb) { bad += 1; continue; }
        const uint8_t x = pa[l], y = pb[l];
        if ((x >= 0x80) == (y >= 0x80)) {
            if (x > y) bad += 1;
            continue;
        }
        mixed += 1;
        const bool node = ref_bucket_is_node(c, i, pa, l, chuck);  // a radix node of the reference: signed child order
        if (node) big += 1;
        const bool high_first = x >= 0x80;
        if (high_first != node) bad += 1;
    }
    if (bad) atomicAdd(&out[0], bad);
    if (mixed) atomicAdd(&out[1], mixed);
    if (big) atomicAdd(&out[2], big);
    if (tie) atomicAdd(&out[3], tie);
}

}  // namespace

void verify_reference_order(Index& ix, uint64_t out[4]) {
    hipStream_t s = ix.stream;
    DevBuf d_out;
    d_out.alloc(4 * sizeof(uint64_t));
    CDB_HIP(hipMemsetAsync(d_out.p, 0, 4 * sizeof(uint64_t), s));
    const uint64_t chuck = std::max<uint64_t>(4096, ix.size / 256);  // index.cpp:218
    if (ix.size > 1) {
        const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(ix.size, 256), 1u << 20);
        sa_dispatch(ix, [&](auto tag) {
            using T = decltype(tag);
            RefOrderCtx<T> c{ix.sa_view<T>(), ix.size, ix.d_text, ix.d_doc_start.as<uint64_t>(), (int)ix.bits, ix.mask};
            hipLaunchKernelGGL((sa_verify_reference_kernel<T>), dim3(grid), dim3(256), 0, s, c, chuck, d_out.as<unsigned long long>());
        });
    }
    CDB_HIP(hipMemcpyAsync(out, d_out.p, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
}

uint64_t count_invalid_entries(hipStream_t s, const void* d_sa, int width, uint64_t n, const uint64_t* d_doc_start, uint64_t ndocs,
                               int bits, uint64_t mask) {
    if (n == 0) return 0;
    DevBuf d_bad;
    d_bad.alloc(sizeof(uint64_t));
    CDB_HIP(hipMemsetAsync(d_bad.p, 0, sizeof(uint64_t), s));
    const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(n, 256), 1u << 16);
    if (width == 4)
        hipLaunchKernelGGL((sa_entry_check_kernel<uint32_t>), dim3(grid), dim3(256), 0, s, static_cast<const uint32_t*>(d_sa), n,
                           d_doc_start, ndocs, bits, mask, d_bad.as<unsigned long long>());
    else
        hipLaunchKernelGGL((sa_entry_check_kernel<uint64_t>), dim3(grid), dim3(256), 0, s, static_cast<const uint64_t*>(d_sa), n,
                           d_doc_start, ndocs, bits, mask, d_bad.as<unsigned long long>());
    uint64_t bad = 0;
    CDB_HIP(hipMemcpyAsync(&bad, d_bad.p, sizeof(bad), hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
    return bad;
}

// pairs out of order / invalid entries among `samples` random adjacent pairs, or among ALL of them (samples = 0); see
// sa_spot_check_kernel
void spot_check_suffix_array(Index& ix, uint32_t samples, uint64_t out[2]) {
    out[0] = out[1] = 0;
    if (ix.size < 2 || ix.width == 0) return;
    hipStream_t s = ix.stream;
    DevBuf d_out;
    d_out.alloc(2 * sizeof(uint64_t));
    CDB_HIP(hipMemsetAsync(d_out.p, 0, 2 * sizeof(uint64_t), s));
    // samples == 0: every adjacent pair (grid-stride sweep)
    const unsigned grid = samples ? (unsigned)ceil_div(samples, 256) : (unsigned)std::min<uint64_t>(ceil_div(ix.size - 1, 256), 1u << 16);
    if (!samples) {
        sa_dispatch(ix, [&](auto tag) {
            using T = decltype(tag);
            const unsigned g2 = (unsigned)std::min<uint64_t>(ceil_div(ix.size, 256), 1u << 16);
            hipLaunchKernelGGL((sa_full_check_kernel<T>), dim3(g2), dim3(256), 0, s, ix.sa_view<T>(), ix.size, ix.d_text,
                               (const uint64_t*)ix.d_doc_start.as<uint64_t>(), ix.ndocs, (int)ix.bits, ix.mask, ix.sa_sorted,
                               d_out.as<unsigned long long>(), (uint64_t)0, ix.size, (unsigned long long*)nullptr,
                               // (every pair means every pair: the bucket-size-dependent ones of a reference-compat order too)
                               ix.sa_sorted ? (uint64_t)0 : std::max<uint64_t>(4096, ix.size / 256));
        });
    } else
    sa_dispatch(ix, [&](auto tag) {
        using T = decltype(tag);
        hipLaunchKernelGGL((sa_spot_check_kernel<T>), dim3(grid), dim3(256), 0, s, ix.sa_view<T>(), ix.size, ix.d_text,
                           (const uint64_t*)ix.d_doc_start.as<uint64_t>(), ix.ndocs, (int)ix.bits, ix.mask, samples,
                           ix.size * 0x9E3779B97F4A7C15ull + ix.ndocs, ix.sa_sorted, d_out.as<unsigned long long>());
    });
    CDB_HIP(hipMemcpyAsync(out, d_out.p, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
}

// ---- proof after publish (index_impl.h: Index::Proof; option self_check = 3) ---------------------------------------------
namespace {
template <typename W>
__global__ void sa_debug_swap_kernel(W* a, uint64_t k) {
    const W t = a[k];
    a[k] = a[k + 1];
    a[k + 1] = t;
}
constexpr uint64_t PROOF_SLICE = 1ull << 24;  // entries per launch: ~0.5 ms — what a build that wants the arrays waits for at most,
                                              // and the longest a library call that arrives mid-slice shares the memory system with it
constexpr double PROOF_MAX_WAIT_MS = 25.0;    // under sustained load one slice runs at least this often (a duty cycle of ~2 %)

// the sweep itself: false = cancelled.  Reads the arrays as they were when the thread started (nothing changes them before
// proof_stop); its launches and the 16-byte result copies are the only work on proof.stream.
bool proof_sweep(Index& ix, uint64_t found[3]) {
    Index::Proof& pf = ix.proof;
    found[0] = found[1] = found[2] = 0;
    CDB_HIP(hipMemsetAsync(pf.d_out, 0, 3 * sizeof(uint64_t), pf.stream));
    for (uint64_t first = 0; first < ix.size; first += PROOF_SLICE) {
        if (pf.cancel.load(std::memory_order_acquire)) return false;
        // slices run in the gaps between library calls (common.h: foreground_calls): beside a batched search the sweep costs the
        // search several times its own duration
        for (const double tw = now_ms(); foreground_calls().load(std::memory_order_acquire) > 0 && now_ms() - tw < PROOF_MAX_WAIT_MS;) {
            if (pf.cancel.load(std::memory_order_acquire)) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        const uint64_t end = std::min<uint64_t>(ix.size, first + PROOF_SLICE);
        sa_dispatch(ix, [&](auto tag) {
            using T = decltype(tag);
            const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(end - first, 256), 1u << 14);
            hipLaunchKernelGGL((sa_full_check_kernel<T>), dim3(grid), dim3(256), 0, pf.stream, ix.sa_view<T>(), ix.size, ix.d_text,
                               (const uint64_t*)ix.d_doc_start.as<uint64_t>(), ix.ndocs, (int)ix.bits, ix.mask, ix.sa_sorted,
                               static_cast<unsigned long long*>(pf.d_out), first, end, static_cast<unsigned long long*>(pf.d_out) + 2,
                               ix.sa_sorted ? (uint64_t)0 : std::max<uint64_t>(4096, ix.size / 256) /* index.cpp:218 */);
        });
        CDB_HIP(hipStreamSynchronize(pf.stream));
    }
    CDB_HIP(hipMemcpyAsync(found, pf.d_out, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, pf.stream));
    CDB_HIP(hipStreamSynchronize(pf.stream));
    // (reference-compat order of text with bytes >= 0x80: the pairs whose order depends on the size of the reference's radix buckets
    //  — 8 % of the pairs of synthetic UTF-8 — were judged in place, sa_full_check_kernel with chuck > 0; none is left out)
    pf.mixed = found[2];
    found[2] = 0;
    return true;
}

// the blocks a rebuild beside this index will miss in the cache: mapped now, on this thread (DevPool::premap)
void premap_next_generation(Index& ix) {
    Index::Proof& pf = ix.proof;
    const double t0 = now_ms();
    const std::vector<size_t> sizes = retained_block_sizes(ix);
    pf.premap_bytes = sizes.empty() ? 0 : DevPool::get().premap(sizes, ix.device);
    pf.premap_ms = now_ms() - t0;
    if (getenv("CDB_BUILD_TRACE") && pf.premap_bytes)
        std::fprintf(stderr, "[premap] %.1f GB mapped for the next generation in %.1f ms\n", (double)pf.premap_bytes / 1e9, pf.premap_ms);
}

void proof_thread(Index* pix) {
    Index& ix = *pix;
    Index::Proof& pf = ix.proof;
    struct Idle {
        Index::Proof& pf;
        ~Idle() { pf.busy.store(false, std::memory_order_release); }
    } idle{pf};
    try {
        CDB_HIP(hipSetDevice(ix.device));
        if (ix.premap_generation && !pf.cancel.load(std::memory_order_acquire)) premap_next_generation(ix);
        if (!pf.want_proof) return;
        const double t0 = now_ms();
        uint64_t found[3] = {0, 0, 0};
        if (!proof_sweep(ix, found)) {
            pf.state.store(5);
            return;
        }
        pf.ms = now_ms() - t0;
        pf.pairs = ix.size - 1;
        pf.found[0] = found[0];
        pf.found[1] = found[1];
        pf.skipped = found[2];
        if (found[0] == 0 && found[1] == 0) {
            pf.state.store(2);
            return;
        }
        // ---- damage: replace the array.  ix.mu keeps the queries out meanwhile; a caller that holds it and waits for this thread
        // (proof_stop from a build that replaces the arrays anyway) is seen through the cancel flag
        std::unique_lock<std::mutex> lk(ix.mu, std::defer_lock);
        while (!lk.try_lock()) {
            if (pf.cancel.load(std::memory_order_acquire)) {
                pf.state.store(5);
                return;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        if (pf.cancel.load(std::memory_order_acquire)) {
            pf.state.store(5);
            return;
        }
        const double t1 = now_ms();
        if (getenv("CDB_BUILD_TRACE"))
            std::fprintf(stderr, "[proof] n=%llu: %llu pairs out of order, %llu invalid entries -> rebuilding with the ballot ranking\n",
                         (unsigned long long)ix.size, (unsigned long long)found[0], (unsigned long long)found[1]);
        StreamScope ss(ix.stream);
        const bool hooked = ix.debug_damage_after_build != 0;
        ix.debug_damage_after_build = 0;
        // (the test hook leaves the device alone; so does a damaged FILE — its array was not sorted here)
        if (!hooked && !pf.of_loaded_file && rs_atomic_rank_ok(ix.stream)) rs_atomic_rank_disable(ix.device);
        pf.of_loaded_file = false;
        ix.self_check_fallbacks += 1;
        const int level = ix.self_check;
        ix.self_check = 2;  // the replacement proves itself before it is served
        ix.proof_in_repair = true;
        struct Restore {
            Index& ix;
            int level;
            ~Restore() {
                ix.self_check = level;
                ix.proof_in_repair = false;
            }
        } restore{ix, level};
        query_resident_stop(ix);
        (void)hipStreamSynchronize(ix.stream);
        ix.release_sa();
        ix.drop_keys();
        ix.d_pivots.release();
        ix.pivot_levels = 0;
        ix.q_spec_cap = 0;
        try {
            build_suffix_array(ix);
            pf.repair_ms = now_ms() - t1;
            pf.state.store(3);
        } catch (const std::exception& e) {
            // (build_suffix_array left the handle "never built": queries answer {} instead of reading a wrong array)
            std::lock_guard<std::mutex> g(ix.err_mu);
            ix.err = std::string("order proof failed and the rebuild did not succeed: ") + e.what();
            pf.state.store(4);
        }
    } catch (...) {
        (void)hipGetLastError();
        if (pf.want_proof) pf.state.store(6);
    }
}
}  // namespace

// Helper threads still at work when the process ends (a handle nobody destroyed: an interpreter shutting down) are cancelled and
// joined BEFORE the block caches and the HIP runtime go away: the registry is constructed after the pools, so it is destroyed first.
namespace {
struct ProofRegistry {
    std::mutex mu;
    std::vector<Index*> live;
    ~ProofRegistry() {
        std::vector<Index*> v;
        {
            std::lock_guard<std::mutex> g(mu);
            v.swap(live);
        }
        for (Index* ix : v) {
            ix->proof.cancel.store(true, std::memory_order_release);
            if (ix->proof.th.joinable() && ix->proof.th.get_id() != std::this_thread::get_id()) ix->proof.th.join();
        }
    }
};
ProofRegistry& proof_registry() {
    (void)DevPool::get();
    (void)HostPool::get();
    static ProofRegistry r;
    return r;
}
}  // namespace
void proof_forget(Index& ix) {  // (cdb_destroy: the handle goes away)
    proof_stop(ix);
    ProofRegistry& r = proof_registry();
    std::lock_guard<std::mutex> g(r.mu);
    r.live.erase(std::remove(r.live.begin(), r.live.end(), &ix), r.live.end());
}

void proof_stop(Index& ix) {
    Index::Proof& pf = ix.proof;
    if (!pf.th.joinable() || pf.th.get_id() == std::this_thread::get_id()) return;
    pf.cancel.store(true, std::memory_order_release);
    pf.th.join();
    pf.cancel.store(false, std::memory_order_release);
}

void proof_start(Index& ix) {
    Index::Proof& pf = ix.proof;
    if (ix.proof_in_repair) return;
    proof_stop(ix);
    pf.state.store(0);
    pf.want_proof = ix.self_check >= 3;
    if (ix.width == 0) return;
    if (ix.size < 2) {
        if (pf.want_proof) pf.state.store(2);  // (nothing to compare)
        return;
    }
    if (!pf.want_proof && !ix.premap_generation) return;
    try {
        if (!pf.stream) {
            int lo = 0, hi = 0;
            CDB_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));  // (lo = the numerically largest = least urgent)
            CDB_HIP(hipStreamCreateWithPriority(&pf.stream, hipStreamNonBlocking, lo));
        }
        if (!pf.d_out) CDB_HIP(hipMalloc(&pf.d_out, 8 * sizeof(uint64_t)));
        if (pf.want_proof) {
            pf.runs += 1;
            pf.state.store(1);
        }
        {
            ProofRegistry& r = proof_registry();
            std::lock_guard<std::mutex> g(r.mu);
            if (std::find(r.live.begin(), r.live.end(), &ix) == r.live.end()) r.live.push_back(&ix);
        }
        pf.busy.store(true, std::memory_order_release);
        pf.th = std::thread(proof_thread, &ix);
    } catch (...) {
        (void)hipGetLastError();
        pf.busy.store(false, std::memory_order_release);
        if (pf.want_proof) pf.state.store(6);
    }
}

void debug_swap_entries(Index& ix, uint64_t k) {
    if (k + 1 >= ix.size) return;
    hipStream_t s = ix.stream;
    if (ix.sa_packed) {
        hipLaunchKernelGGL((sa_debug_swap_kernel<uint32_t>), dim3(1), dim3(1), 0, s, ix.d_sa.as<uint32_t>(), k);
        hipLaunchKernelGGL((sa_debug_swap_kernel<uint8_t>), dim3(1), dim3(1), 0, s, ix.d_sa_hi.as<uint8_t>(), k);
    } else if (ix.width == 8) {
        hipLaunchKernelGGL((sa_debug_swap_kernel<uint64_t>), dim3(1), dim3(1), 0, s, ix.d_sa.as<uint64_t>(), k);
    } else {
        hipLaunchKernelGGL((sa_debug_swap_kernel<uint32_t>), dim3(1), dim3(1), 0, s, ix.d_sa.as<uint32_t>(), k);
    }
    CDB_HIP(hipStreamSynchronize(s));
}

// ---- packed storage (index_impl.h: Sa40) <-> the reference's u64 entries ---------------------------------------------
namespace {
__global__ __launch_bounds__(256) void sa_expand_kernel(Sa40 sa, uint64_t first, uint64_t cnt, uint64_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += stride) out[i] = sa[first + i];
}
__global__ __launch_bounds__(256) void sa_pack_kernel(const uint64_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ lo,
                                                      uint8_t* __restrict__ hi) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint64_t e = in[i];
        lo[i] = (uint32_t)e;
        hi[i] = (uint8_t)(e >> 32);
    }
}
}  // namespace

// entries [first, first + cnt) of a packed index as u64 in d_out (cdb_sa_copy, cdb_save)
void sa_expand(Index& ix, uint64_t first, uint64_t cnt, uint64_t* d_out) {
    if (!cnt) return;
    hipLaunchKernelGGL(sa_expand_kernel, dim3((unsigned)std::min<uint64_t>(ceil_div(cnt, 256), 1u << 16)), dim3(256), 0, ix.stream,
                       ix.sa_view<Packed40>(), first, cnt, d_out);
}
// ix.d_sa holds size u64 entries below 2^40 (cdb_load; builds that could not write the packed form themselves): store them
// packed.  Costs one sweep (8 B read, 5 B written per entry); the u64 block goes back to the pool.
void sa_pack_inplace(Index& ix) {
    if (ix.sa_packed || ix.width != 8 || !ix.size) return;
    DevBuf lo, hi;
    lo.alloc(ix.size * sizeof(uint32_t));
    hi.alloc(ix.size);
    hipLaunchKernelGGL(sa_pack_kernel, dim3((unsigned)std::min<uint64_t>(ceil_div(ix.size, 256), 1u << 16)), dim3(256), 0, ix.stream,
                       (const uint64_t*)ix.d_sa.as<uint64_t>(), ix.size, lo.as<uint32_t>(), hi.as<uint8_t>());
    CDB_HIP(hipStreamSynchronize(ix.stream));
    ix.d_sa = std::move(lo);
    ix.d_sa_hi = std::move(hi);
    ix.sa_packed = true;
}

// cnt u64 entries in d_in -> packed storage at [first, first + cnt) of (lo, hi): cdb_load packs chunk by chunk while it reads the
// file, so the plain 8-byte array never exists on the device (peak 5 n + one chunk instead of 13 n)
void sa_pack_chunk(hipStream_t s, const uint64_t* d_in, uint64_t cnt, uint32_t* lo, uint8_t* hi, uint64_t first) {
    if (!cnt) return;
    hipLaunchKernelGGL(sa_pack_kernel, dim3((unsigned)std::min<uint64_t>(ceil_div(cnt, 256), 1u << 16)), dim3(256), 0, s, d_in, cnt, lo + first,
                       hi + first);
}

void verify_suffix_array(Index& ix, uint64_t out[5]) {
    hipStream_t s = ix.stream;
    DevBuf d_out;
    d_out.alloc(4 * sizeof(uint64_t));
    CDB_HIP(hipMemsetAsync(d_out.p, 0, 4 * sizeof(uint64_t), s));
    if (ix.size) {
        const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(ix.size, 256), 1u << 22);
        sa_dispatch(ix, [&](auto tag) {
            using T = decltype(tag);
            hipLaunchKernelGGL((sa_verify_kernel<T>), dim3(grid), dim3(256), 0, s, ix.sa_view<T>(), ix.size, ix.d_text,
                               (const uint64_t*)ix.d_doc_start.as<uint64_t>(), ix.ndocs, (int)ix.bits, ix.mask, d_out.as<unsigned long long>());
        });
    }
    CDB_HIP(hipMemcpyAsync(out, d_out.p, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
    // closed-form wrapped sum of all (off << bits) | doc entries
    ensure_host_tables(ix);
    uint64_t expect = 0;
    for (uint64_t d = 0; d < ix.ndocs; ++d) {
        const uint64_t len = ix.doc_start[d + 1] - ix.doc_start[d];
        const uint64_t tri = (len & 1) ? len * ((len - 1) / 2) : (len / 2) * (len - 1);  // len(len-1)/2 mod 2^64
        expect += (tri << ix.bits) + len * d;
    }
    out[4] = expect;
}

}  // namespace cdb
