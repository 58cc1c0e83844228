// query.hip — batched substring match on the GPU suffix array.
//
// Replaces string_index::query (reference /root/reference/src/index.cpp:237-326) for a whole batch of
// keywords at once:
//   a9/a10  two binary searches per keyword with the reference's exact probe sequence
//           (index.cpp:260-287: lower bound saturating at size-1, then the prefix upper bound) —
//           the same midpoints are visited, so results agree with the reference on any SA it could
//           have produced, including the not-globally-sorted one of SURVEY.md Q2;
//   a11     hit gather (doc = entry & mask) + sort by document: one stable radix sort of
//           (pattern id ∘ doc) keys for the whole batch instead of one sort per keyword;
//   a12     run-length encoding into CSR rows (ids[doc], count), ascending document index.
#include <cstring>

#include <chrono>

#include "index_impl.h"
#include "scan.h"

namespace cdb {
namespace {

__device__ __forceinline__ uint64_t load_be8(const uint8_t* p) {
    // 8 text bytes as a big-endian integer: integer order == unsigned lexicographic order
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}

// three-way compare of keyword k[0..m) against suffix s[0..sl) on their common length, then the
// reference's two predicates are derived from (c, m, sl):
//   keyword <= suffix  <=>  c < 0 || (c == 0 && m <= sl)          (index.cpp:267, string_view <=)
//   suffix starts with keyword <=> sl >= m && c == 0               (index.cpp:280)
__device__ __forceinline__ int cmp_common(const uint8_t* __restrict__ k, uint64_t m, const uint8_t* __restrict__ s,
                                          uint64_t sl) {
    const uint64_t len = m < sl ? m : sl;
    uint64_t i = 0;
    for (; i + 8 <= len; i += 8) {
        const uint64_t a = load_be8(k + i), b = load_be8(s + i);
        if (a != b) return a < b ? -1 : 1;
    }
    for (; i < len; ++i) {
        const uint8_t a = k[i], b = s[i];
        if (a != b) return a < b ? -1 : 1;
    }
    return 0;
}

template <typename V>
__global__ __launch_bounds__(256) void q_search_kernel(typename SaOf<V>::ptr sa, uint64_t n,
                                                       const uint8_t* __restrict__ text,
                                                       const uint64_t* __restrict__ doc_start, int bits, uint64_t mask,
                                                       const uint8_t* __restrict__ blob,
                                                       const uint64_t* __restrict__ offs, uint64_t npat,
                                                       int64_t* __restrict__ left_out, uint64_t* __restrict__ hits_out) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= npat) return;
    const uint8_t* k = blob + offs[j];
    const uint64_t m = offs[j + 1] - offs[j];
    if (m == 0 || offs[j + 1] < offs[j]) {  // empty pattern (device batches are not pre-validated): no rows
        left_out[j] = 0;
        hits_out[j] = 0;
        return;
    }
    int64_t L = 0, R = (int64_t)n - 1;
    while (L < R) {
        const int64_t M = L + (R - L) / 2;
        const auto e = sa[M];
        const uint64_t d = (uint64_t)e & mask, off = (uint64_t)e >> bits;
        const uint64_t b = doc_start[d] + off, sl = doc_start[d + 1] - b;
        const int c = cmp_common(k, m, text + b, sl);
        if (c < 0 || (c == 0 && m <= sl)) R = M; else L = M + 1;
    }
    const int64_t left = L;
    L = left - 1;
    R = (int64_t)n - 1;
    while (L < R) {
        const int64_t M = L + (R - L + 1) / 2;
        const auto e = sa[M];
        const uint64_t d = (uint64_t)e & mask, off = (uint64_t)e >> bits;
        const uint64_t b = doc_start[d] + off, sl = doc_start[d + 1] - b;
        const bool pref = sl >= m && cmp_common(k, m, text + b, sl) == 0;
        if (pref) L = M; else R = M - 1;
    }
    const int64_t right = L + 1;
    left_out[j] = left;
    hits_out[j] = right > left ? (uint64_t)(right - left) : 0ull;
}

// ---- fast search for globally sorted suffix arrays ---------------------------------------------------
// The reference's lower-bound loop visits the same midpoints for every keyword on its first levels
// (L = 0, R = n-1, M = L + (R-L)/2, ...).  The first PIVOT_LEVELS levels of that tree — 2^levels - 1
// suffixes — are summarised once per index as (first 16 bytes big-endian, suffix length) and staged in
// LDS, so those levels cost no global memory access at all; only a keyword that agrees with a pivot on
// all 16 bytes and is longer than 16 falls back to the text.  The prefix upper bound gallops from the
// lower bound (1, 2, 4, ... then bisects) instead of bisecting [left-1, n-1]: hit ranges are short, so
// this is a handful of probes instead of log2(n).  Both searches return exactly what the reference's
// loops return whenever the array is sorted; the reference-compat orderings of text with bytes >= 0x80
// are not, and keep using q_search_kernel (the reference's own probe sequence).
constexpr int PIVOT_LEVELS = 11;
constexpr int PIVOT_NODES = (1 << PIVOT_LEVELS) - 1;
struct Pivot {
    uint64_t hi, lo;  // bytes 0..7 and 8..15 of the suffix, big-endian, zero padded
    uint64_t sl;      // suffix length
};

template <typename V>
__global__ __launch_bounds__(256) void q_pivots_kernel(typename SaOf<V>::ptr sa, uint64_t n,
                                                       const uint8_t* __restrict__ text,
                                                       const uint64_t* __restrict__ doc_start, int bits, uint64_t mask,
                                                       int levels, Pivot* __restrict__ piv) {
    const uint32_t id = blockIdx.x * 256 + threadIdx.x + 1;  // heap order: root 1, children 2id (R = M), 2id+1 (L = M+1)
    if (id >= (1u << levels)) return;
    int depth = 31 - __clz(id);
    int64_t L = 0, R = (int64_t)n - 1;
    bool dead = false;
    for (int b = depth - 1; b >= 0; --b) {
        if (L >= R) { dead = true; break; }
        const int64_t M = L + (R - L) / 2;
        if ((id >> b) & 1u) L = M + 1; else R = M;
    }
    Pivot p{0, 0, 0};
    if (!dead && L < R) {
        const int64_t M = L + (R - L) / 2;
        const auto e = sa[M];
        const uint64_t d = (uint64_t)e & mask, off = (uint64_t)e >> bits;
        const uint64_t b0 = doc_start[d] + off, sl = doc_start[d + 1] - b0;
        uint64_t w[2] = {0, 0};
        for (int k = 0; k < 16 && (uint64_t)k < sl; ++k) w[k >> 3] |= (uint64_t)text[b0 + k] << (56 - 8 * (k & 7));
        p = Pivot{w[0], w[1], sl};
    }
    piv[id] = p;
}

// keyword <= suffix ?  decided from 16-byte prefixes when possible; returns 0 / 1, or 2 = undecided
__device__ __forceinline__ int kw_le_pivot(uint64_t khi, uint64_t klo, uint64_t m, const Pivot& p) {
    const uint64_t common = m < p.sl ? m : p.sl;
    const uint64_t c16 = common < 16 ? common : 16;
    // compare the first c16 bytes: mask both sides
    uint64_t mh = ~0ull, ml = ~0ull;
    if (c16 < 8) { mh = c16 ? ~0ull << (64 - 8 * c16) : 0ull; ml = 0; }
    else if (c16 < 16) { ml = c16 > 8 ? ~0ull << (64 - 8 * (c16 - 8)) : 0ull; }
    const uint64_t ah = khi & mh, bh = p.hi & mh, al = klo & ml, bl = p.lo & ml;
    if (ah != bh) return ah < bh ? 1 : 0;
    if (al != bl) return al < bl ? 1 : 0;
    if (common <= 16) return m <= p.sl ? 1 : 0;  // equal on the whole common part
    return 2;
}

// G = lanes per keyword.  G = 1: plain bisection below the pivot levels.  G = 8: the lanes of a group probe G slots per
// round, which split the range G + 1 ways — log9 instead of log2 rounds of DEPENDENT loads (6 instead of 19 below the
// pivot levels at 2^30 suffixes) for G times the loads: the better trade while a batch is too small to fill the GPU
// with one thread per keyword.  The array is sorted, so every search strategy finds the same bounds.
template <typename V, int G>
__global__ __launch_bounds__(256) void q_search_fast_kernel(typename SaOf<V>::ptr sa, uint64_t n,
                                                            const uint8_t* __restrict__ text,
                                                            const uint64_t* __restrict__ doc_start, int bits,
                                                            uint64_t mask, const uint8_t* __restrict__ blob,
                                                            const uint64_t* __restrict__ offs, uint64_t npat,
                                                            const Pivot* __restrict__ piv, int levels,
                                                            const uint64_t* __restrict__ keys64,
                                                            const uint32_t* __restrict__ keys32,
                                                            const void* __restrict__ keylow, int low_bits, int low_bytes,
                                                            const uint16_t* __restrict__ symmap, int nsym, uint32_t kbase,
                                                            bool refseq, int64_t* __restrict__ left_out,
                                                            uint64_t* __restrict__ hits_out) {
    __shared__ Pivot s_piv[PIVOT_NODES + 1];
    __shared__ uint16_t s_code[256];
    const bool keys = keys64 != nullptr || keys32 != nullptr;  // sorted initial keys available
    // (refseq: a reference-compat ordering — not globally sorted — is searched with the reference's own two
    //  bisections, levels = 0; the kept keys, reordered with the array, still decide most probes from one load)
    for (int i = threadIdx.x; levels > 0 && i < (1 << levels); i += 256) s_piv[i] = piv[i];
    if (keys) s_code[threadIdx.x] = symmap[threadIdx.x];
    __syncthreads();
    const uint64_t j = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (j >= npat) return;
    const int sub = (int)(threadIdx.x & (G - 1));             // this lane's place in its group
    const int gshift = (int)((threadIdx.x & 63) & ~(G - 1));  // first lane of the group inside the wavefront
    const uint8_t* k = blob + offs[j];
    const uint64_t m = offs[j + 1] - offs[j];
    if (m == 0 || offs[j + 1] < offs[j]) {  // empty pattern (device batches are not pre-validated): no rows
        if (sub == 0) {
            left_out[j] = 0;
            hits_out[j] = 0;
        }
        return;
    }
    uint64_t kw[2] = {0, 0};
    for (int q = 0; q < 16 && (uint64_t)q < m; ++q) kw[q >> 3] |= (uint64_t)k[q] << (56 - 8 * (q & 7));
    // keyword in the key alphabet: its first min(m, nsym) symbol codes, packed like keys[]; a byte that
    // does not occur in the text (code 0) means the keyword occurs nowhere
    const int kc = keys ? (int)(m < (uint64_t)nsym ? m : (uint64_t)nsym) : 0;
    uint64_t kwc = 0;
    bool absent = false;
    for (int q = 0; q < kc; ++q) {
        const uint64_t c = s_code[k[q]];
        absent |= c == 0;
        kwc = kwc * kbase + c;
    }
    if (keys)
        for (uint64_t q = kc; q < m; ++q) absent |= s_code[k[q]] == 0;
    if (absent) {
        if (sub == 0) {
            left_out[j] = 0;
            hits_out[j] = 0;
        }
        return;
    }
    // keys are numbers in base kbase: the suffixes starting with the keyword's first kc symbols are exactly
    // those with key in [klo, klo + kpw)
    uint64_t kpw = 1;
    for (int q = kc; q < nsym; ++q) kpw *= kbase;
    const uint64_t klo = kwc * kpw;
    // three-way answer from the key of slot M alone: -1 suffix < keyword, +1 keyword < suffix,
    // 0 = the suffix starts with the keyword's first kc symbols (decisive iff m <= nsym)
    // (split keys: key = (keys32 << low_bits) | keylow; the 32-bit part alone decides unless it equals the
    //  truncated range end it is compared with)
    const uint64_t khi = klo + (kpw - 1);  // last key of the range
    auto key_cmp = [&](int64_t M) -> int {
        uint64_t sk;
        if (keys64) {
            sk = keys64[M];
        } else {
            const uint64_t h = keys32[M];
            if (keylow) {
                const uint64_t a = klo >> low_bits, b = khi >> low_bits;
                if (h < a) return -1;
                if (h > b) return 1;
                if (h > a && h < b) return 0;
                sk = (h << low_bits) | (low_bytes == 1 ? (uint64_t)static_cast<const uint8_t*>(keylow)[M]
                                                        : (uint64_t)static_cast<const uint16_t*>(keylow)[M]);
            } else {
                sk = h;
            }
        }
        return sk < klo ? -1 : (sk > khi ? 1 : 0);
    };
    auto suffix_of = [&](int64_t M, const uint8_t*& sp, uint64_t& sl) {
        const auto e = sa[M];
        const uint64_t d = (uint64_t)e & mask, off = (uint64_t)e >> bits;
        const uint64_t b = doc_start[d] + off;
        sp = text + b;
        sl = doc_start[d + 1] - b;
    };
    // ---- lower bound (index.cpp:260-274), first `levels` levels from LDS
    int64_t L = 0, R = (int64_t)n - 1;
    uint32_t id = 1;
    auto le_at = [&](int64_t M, int le) -> bool {  // keyword <= suffix(M)?  (le = 2: not decided by a pivot)
        if (le == 2 && keys) {
            const int c = key_cmp(M);
            if (c != 0) le = c > 0 ? 1 : 0;
            else if (m <= (uint64_t)nsym) le = 1;  // keyword is a prefix of the suffix: keyword <= suffix
        }
        if (le == 2) {
            const uint8_t* sp;
            uint64_t sl;
            suffix_of(M, sp, sl);
            const int c = cmp_common(k, m, sp, sl);
            le = (c < 0 || (c == 0 && m <= sl)) ? 1 : 0;
        }
        return le != 0;
    };
    while (L < R && (G == 1 || (levels > 0 && id < (1u << levels)))) {  // (G > 1: the pivot levels only; every lane alike)
        const int64_t M = L + (R - L) / 2;
        int le = 2;
        if (levels > 0 && id < (1u << levels)) le = kw_le_pivot(kw[0], kw[1], m, s_piv[id]);
        if (le_at(M, le)) { R = M; id = 2 * id; } else { L = M + 1; id = 2 * id + 1; }
    }
    if constexpr (G > 1) {
        for (;;) {
            const bool open = L < R;
            if (!__any(open)) break;  // (wave-uniform: groups that are done idle through the remaining rounds)
            const int64_t span = R - L;
            const bool narrow = span <= (int64_t)G;  // the last round probes consecutive slots; slot R is the saturated answer
            bool le = false;
            if (open) {
                const int64_t M = narrow ? L + sub : L + (span * (sub + 1)) / (G + 1);
                le = M >= R ? true : le_at(M, 2);
            }
            const uint32_t gb = (uint32_t)((__ballot(le) >> gshift) & ((1ull << G) - 1ull));
            if (open) {
                const int f = gb ? __ffs((int)gb) - 1 : G;
                if (narrow) {
                    L = R = L + f;  // (f <= span: the lane on slot R always answers "le")
                } else if (f == G) {
                    L = L + (span * G) / (G + 1) + 1;
                } else {
                    const int64_t r2 = L + (span * (f + 1)) / (G + 1);
                    if (f > 0) L = L + (span * f) / (G + 1) + 1;
                    R = r2;
                }
            }
        }
    }
    const int64_t left = L;
    // ---- prefix upper bound (index.cpp:275-287) by galloping from `left`
    auto is_prefix = [&](int64_t M) -> bool {
        if (keys) {
            if (key_cmp(M) != 0) return false;
            if (m <= (uint64_t)nsym) return true;
        }
        const uint8_t* sp;
        uint64_t sl;
        suffix_of(M, sp, sl);
        return sl >= m && cmp_common(k, m, sp, sl) == 0;
    };
    int64_t right = left;  // first index >= left whose suffix does not start with the keyword
    if (refseq) {  // index.cpp:275-287 literally
        int64_t A = left - 1, B = (int64_t)n - 1;
        while (A < B) {
            const int64_t M = A + (B - A + 1) / 2;
            if (is_prefix(M)) A = M; else B = M - 1;
        }
        right = A + 1;
    } else if (G > 1) {
        // the group probes the G slots from `left` on at once; only a run of matches longer than that gallops on
        // (every lane of the group alike: the same addresses, no divergence inside the group)
        const int64_t M = left + sub;
        const bool pf = M < (int64_t)n && is_prefix(M);
        const uint32_t gb = (uint32_t)((__ballot(!pf) >> gshift) & ((1ull << G) - 1ull));
        if (gb) {
            right = left + (__ffs((int)gb) - 1);
        } else {
            int64_t good = left + G - 1, step = 1, bad = (int64_t)n;
            while (good + step < (int64_t)n) {
                if (is_prefix(good + step)) { good += step; step <<= 1; }
                else { bad = good + step; break; }
            }
            while (good + 1 < bad) {
                const int64_t mid = good + (bad - good) / 2;
                if (is_prefix(mid)) good = mid; else bad = mid;
            }
            right = good + 1;
        }
    } else if (n > 0 && is_prefix(left)) {
        int64_t good = left, step = 1, bad = (int64_t)n;
        while (good + step < (int64_t)n) {
            if (is_prefix(good + step)) { good += step; step <<= 1; }
            else { bad = good + step; break; }
        }
        while (good + 1 < bad) {  // invariant: good is a prefix match, bad is not (or n)
            const int64_t mid = good + (bad - good) / 2;
            if (is_prefix(mid)) good = mid; else bad = mid;
        }
        right = good + 1;
    }
    if (sub == 0) {
        left_out[j] = left;
        hits_out[j] = right > left ? (uint64_t)(right - left) : 0ull;
    }
}

struct OpSumMax {  // (a, b) = (sum, max)
    __device__ __forceinline__ U2 operator()(const U2& x, const U2& y) const { return U2{x.a + y.a, x.b > y.b ? x.b : y.b}; }
};
struct HitsIn2 {
    const uint64_t* hits;
    __device__ __forceinline__ U2 operator()(uint64_t j) const { return U2{hits[j], hits[j]}; }
};
struct HitsOut2 {
    uint64_t* hoff;
    uint64_t npat;
    __device__ __forceinline__ void operator()(uint64_t j, const U2& ex, const U2& in) const {
        hoff[j] = ex.a;
        if (j + 1 == npat) hoff[npat] = in.a;
    }
};

// ---- rows of patterns with at most 64 hits: one wavefront per pattern ---------------------------------
// The common case (keywords of a few characters or more) has short hit lists, for which a device-wide
// radix sort of (pattern, doc) keys plus two scans is mostly launch and synchronisation latency.  Here a
// wavefront loads its pattern's hits straight from the suffix array (one entry per lane), sorts the 64
// document indices with a bitonic network of lane shuffles, and run-length encodes them with a ballot —
// replaces index.cpp:288-322 for that pattern.  Phase 1 leaves compacted (doc, count) pairs at the
// pattern's hit offset and its row count; after a scan of the row counts phase 2 moves the rows to
// their CSR position and maps documents to object ids.
template <typename V>
__global__ __launch_bounds__(256) void q_wave_rows_kernel(typename SaOf<V>::ptr sa, uint64_t mask,
                                                          const int64_t* __restrict__ left,
                                                          const uint64_t* __restrict__ hits,
                                                          const uint64_t* __restrict__ hoff, uint64_t npat,
                                                          uint32_t* __restrict__ row_doc, uint32_t* __restrict__ row_cnt,
                                                          uint64_t* __restrict__ nrows, uint64_t cap,
                                                          unsigned long long* __restrict__ spill) {
    const int lane = threadIdx.x & 63;
    const uint64_t j = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= npat) return;
    // (speculative launches — buffers sized from the previous batch, hit totals not yet known to the host — flag
    //  what does not fit one wavefront or the buffers; the host then redoes the batch the ordinary way)
    if (hits[j] > 64 || hoff[j] + hits[j] > cap) {
        if (lane == 0) {
            nrows[j] = 0;
            if (spill) atomicOr(spill, 1ull);
        }
        return;
    }
    const uint32_t h = (uint32_t)hits[j];
    uint32_t v = 0xFFFFFFFFu;  // sentinel behind every document index
    if ((uint32_t)lane < h) v = (uint32_t)((uint64_t)sa[(uint64_t)left[j] + lane] & mask);
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int q = k >> 1; q > 0; q >>= 1) {
            const uint32_t o = __shfl_xor(v, q);
            const bool up = (lane & k) == 0;          // ascending block
            const bool lower = (lane & q) == 0;       // this lane keeps the smaller element of the pair
            const uint32_t mn = v < o ? v : o, mx = v < o ? o : v;
            v = (lower == up) ? mn : mx;
        }
    }
    const uint32_t prev = __shfl_up(v, 1);
    const bool head = (uint32_t)lane < h && (lane == 0 || v != prev);
    const uint64_t heads = __ballot(head);
    if (head) {
        const uint32_t r = __popcll(heads & ((1ull << lane) - 1ull));
        const uint64_t later = heads & ~((2ull << lane) - 1ull);       // heads behind this lane
        const uint32_t next = later ? (uint32_t)(__ffsll((unsigned long long)later) - 1) : h;
        row_doc[hoff[j] + r] = v;
        row_cnt[hoff[j] + r] = next - (uint32_t)lane;
    }
    if (lane == 0) nrows[j] = (uint64_t)__popcll(heads);
}

__global__ __launch_bounds__(256) void q_wave_emit_kernel(const uint32_t* __restrict__ row_doc,
                                                          const uint32_t* __restrict__ row_cnt,
                                                          const uint64_t* __restrict__ hoff,
                                                          const uint64_t* __restrict__ row_ptr, uint64_t npat,
                                                          const int64_t* __restrict__ ids, int64_t* __restrict__ out_ids,
                                                          int64_t* __restrict__ out_counts) {
    const int lane = threadIdx.x & 63;
    const uint64_t j = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= npat) return;
    const uint64_t a = row_ptr[j], nr = row_ptr[j + 1] - a;
    if ((uint64_t)lane < nr) {
        out_ids[a + lane] = ids[row_doc[hoff[j] + lane]];
        out_counts[a + lane] = (int64_t)row_cnt[hoff[j] + lane];
    }
}
struct NrowsIn {
    const uint64_t* nrows;
    __device__ __forceinline__ uint64_t operator()(uint64_t j) const { return nrows[j]; }
};

struct HitsIn {
    const uint64_t* hits;
    __device__ __forceinline__ uint64_t operator()(uint64_t j) const { return hits[j]; }
};
struct HitsOut {
    uint64_t* hoff;
    uint64_t npat;
    __device__ __forceinline__ void operator()(uint64_t j, uint64_t ex, uint64_t in) const {
        hoff[j] = ex;
        if (j + 1 == npat) hoff[npat] = in;
    }
};

// one thread per hit slot of the chunk [j0, j1) of patterns: find the owning pattern (binary search
// over the hit offsets), emit ((pattern - j0) << dbits) | doc
template <typename V>
__global__ __launch_bounds__(256) void q_expand_kernel(typename SaOf<V>::ptr sa, uint64_t mask, int dbits, int bits,
                                                       int obits, const int64_t* __restrict__ left,
                                                       const uint64_t* __restrict__ hoff, uint64_t j0, uint64_t j1,
                                                       uint64_t H, uint64_t* __restrict__ keys) {
    // (round 6: the workgroup's first and last slot are located in the whole chunk — ~20 dependent loads each —, every other thread
    //  only between those two patterns: a handful of patterns for 256 consecutive slots)
    __shared__ uint64_t s_rng[2];
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t h0 = hoff[j0];
    auto owner = [&](uint64_t slot, uint64_t lo, uint64_t hi) {  // largest j in [lo, hi] with hoff[j] <= slot
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo + 1) / 2;
            if (hoff[mid] <= slot) lo = mid; else hi = mid - 1;
        }
        return lo;
    };
    if (threadIdx.x == 0 || threadIdx.x == 255) {
        const uint64_t tb = (uint64_t)blockIdx.x * 256 + (threadIdx.x ? 255u : 0u);
        s_rng[threadIdx.x ? 1 : 0] = owner(h0 + (tb < H ? tb : H - 1), j0, j1 - 1);
    }
    __syncthreads();
    if (t >= H) return;
    const uint64_t slot = h0 + t;
    const uint64_t j = owner(slot, s_rng[0], s_rng[1]);
    const uint64_t i = (uint64_t)left[j] + (slot - hoff[j]);
    // obits > 0: the occurrence offset rides along as the least significant field (offset emission)
    const auto e = sa[i];
    const uint64_t pd = ((j - j0) << dbits) | ((uint64_t)e & mask);
    keys[t] = obits ? (pd << obits) | ((uint64_t)e >> bits) : pd;
}

// Rows of a sorted hit list: a row = a run of hits with one (pattern, doc), index.cpp:316-322.  Round 6: two kernels of their own
// instead of the generic scan with functors.  The generic scan gives every thread 16 CONSECUTIVE items, so one load instruction of a
// wave touched 64 different 128-byte lines and the occurrence offsets went out as 64 partial lines per store instruction: 1.55 +
// 6.75 ms for the 1.05 x 10^8 hits of BASELINE config 2's batch (0.25 TB/s) — a third of the whole query.  Here a wave's lanes
// take consecutive hits (16 rows of 256 per tile of SC_TILE), heads are ranked by wave ballots, and all traffic is coalesced.
__global__ __launch_bounds__(256) void q_run_count_kernel(const uint64_t* __restrict__ keys, uint64_t H, int obits,
                                                          uint64_t* __restrict__ partials) {
    static_assert(SC_TILE == 4096, "run kernels assume 4096-hit tiles (16 rows of 256)");
    __shared__ uint32_t s_w[4];
    const uint64_t tile0 = (uint64_t)blockIdx.x * SC_TILE;
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint64_t i = tile0 + (uint64_t)j * 256 + threadIdx.x;
        if (i < H) cnt += (i == 0 || (keys[i] >> obits) != (keys[i - 1] >> obits)) ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (uint64_t)s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
// partials: exclusive scan of the tile counts.  row r starts at hit slot row_first[r] (row_first[nrows] = H), row_key[r] =
// (pattern ∘ doc) of the row; hit_off (optional): occurrence offset of every hit slot, in sorted order
__global__ __launch_bounds__(256) void q_run_apply_kernel(const uint64_t* __restrict__ keys, uint64_t H, int obits,
                                                          const uint64_t* __restrict__ partials, uint64_t* __restrict__ row_first,
                                                          uint64_t* __restrict__ row_key, uint64_t* __restrict__ hit_off) {
    __shared__ uint32_t s_cnt[16][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint64_t tile0 = (uint64_t)blockIdx.x * SC_TILE;
    const uint64_t omask = (1ull << obits) - 1ull;
    uint64_t k[16];
    uint32_t below[16], heads = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint64_t i = tile0 + (uint64_t)j * 256 + threadIdx.x;
        const bool valid = i < H;
        k[j] = valid ? keys[i] : 0ull;
        const bool head = valid && (i == 0 || (k[j] >> obits) != (keys[i - 1] >> obits));
        const uint64_t bal = __builtin_amdgcn_ballot_w64(head);
        below[j] = (uint32_t)__popcll(bal & lt_mask);
        heads |= (head ? 1u : 0u) << j;
        if (lane == 0) s_cnt[j][wave] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    uint64_t run = partials[blockIdx.x];  // rows in front of this tile
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t c0 = s_cnt[j][0], c1 = s_cnt[j][1], c2 = s_cnt[j][2], c3 = s_cnt[j][3];
        const uint32_t pre = (wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2 : 0u);
        const uint64_t i = tile0 + (uint64_t)j * 256 + threadIdx.x;
        if (i < H) {
            const uint32_t head = (heads >> j) & 1u;
            const uint64_t ex = run + pre + below[j];
            if (head) {
                row_first[ex] = i;
                row_key[ex] = k[j] >> obits;
            }
            if (hit_off) hit_off[i] = k[j] & omask;
            if (i + 1 == H) row_first[ex + head] = H;
        }
        run += (uint64_t)c0 + c1 + c2 + c3;
    }
}

__global__ __launch_bounds__(256) void q_rows_kernel(const uint64_t* __restrict__ row_first,
                                                     const uint64_t* __restrict__ row_key, uint64_t nrows, int dbits,
                                                     const int64_t* __restrict__ ids, int64_t* __restrict__ out_ids,
                                                     int64_t* __restrict__ out_counts, uint64_t hits_base,
                                                     uint64_t* __restrict__ hit_ptr) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const uint64_t doc = row_key[r] & ((1ull << dbits) - 1ull);
    out_ids[r] = ids[doc];
    out_counts[r] = (int64_t)(row_first[r + 1] - row_first[r]);
    if (hit_ptr) hit_ptr[r] = hits_base + row_first[r];  // the row's occurrences are hit slots [hit_ptr[r], hit_ptr[r+1])
}

// row_ptr[j0 + k] = rows_base + first row of the chunk whose local pattern id >= k, k = 0 .. count-1
__global__ __launch_bounds__(256) void q_rowptr_kernel(const uint64_t* __restrict__ row_key, uint64_t nrows, int dbits,
                                                       uint64_t j0, uint64_t count, uint64_t rows_base,
                                                       uint64_t* __restrict__ row_ptr) {
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    uint64_t lo = 0, hi = nrows;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if ((row_key[mid] >> dbits) < k) lo = mid + 1; else hi = mid;
    }
    row_ptr[j0 + k] = rows_base + lo;
}

// grows a device buffer to `need` bytes, keeping its first `used` bytes
void grow_keep(DevBuf& b, size_t need, size_t used, hipStream_t s) {
    if (need <= b.bytes) return;
    DevBuf nb;
    nb.alloc(need + need / 2);
    if (used) CDB_HIP(hipMemcpyAsync(nb.p, b.p, used, hipMemcpyDeviceToDevice, s));
    CDB_HIP(hipStreamSynchronize(s));  // the old block goes back to the shared cache
    b = std::move(nb);
}

// runs the keyword search (fast path on sorted arrays, the reference's probe sequence otherwise)
template <typename V>
void launch_search(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat) {
    hipStream_t s = ix.stream;
    const auto sa = ix.sa_view<V>();
    const uint64_t* doc_start = ix.d_doc_start.as<uint64_t>();
    int t = ix.prof.begin(s);
    if (ix.sa_sorted && ix.use_fast_search && ix.size >= 4096) {
        if (ix.pivot_levels == 0) {
            int levels = PIVOT_LEVELS;
            while (levels > 1 && (1ull << levels) > ix.size / 2) --levels;
            ix.d_pivots.alloc(((size_t)1 << levels) * sizeof(Pivot));
            hipLaunchKernelGGL((q_pivots_kernel<V>), dim3((unsigned)ceil_div((1u << levels), 256)), dim3(256), 0, s, sa, ix.size,
                               ix.d_text, doc_start, (int)ix.bits, ix.mask, levels, ix.d_pivots.as<Pivot>());
            ix.pivot_levels = levels;
        }
        // a group of 8 lanes per keyword for SMALL batches (coalesced single queries, a few thousand keywords), where the
        // chain of dependent probes is the cost: 1000 keywords 0.090 -> 0.055 ms; from ~10^4 keywords on the random loads
        // themselves are (10^5 keywords: 0.17 ms with one lane, 0.34 ms with eight).  search_lanes: 0 = by batch size
        const bool wide = ix.search_lanes == 8 || (ix.search_lanes == 0 && npat <= 4096);
        auto kern = wide ? q_search_fast_kernel<V, 8> : q_search_fast_kernel<V, 1>;
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(npat * (wide ? 8 : 1), 256)), dim3(256), 0, s, sa, ix.size, ix.d_text,
                           doc_start, (int)ix.bits, ix.mask, d_blob, d_offs, npat, (const Pivot*)ix.d_pivots.as<Pivot>(),
                           ix.pivot_levels,
                           ix.key_nsym && ix.d_keys.p ? (const uint64_t*)ix.d_keys.as<uint64_t>() : (const uint64_t*)nullptr,
                           ix.key_nsym && ix.d_keys32.p ? (const uint32_t*)ix.d_keys32.as<uint32_t>() : (const uint32_t*)nullptr,
                           ix.key_nsym && ix.d_keylow.p ? (const void*)ix.d_keylow.p : (const void*)nullptr, ix.key_low_bits,
                           ix.key_low_bytes, (const uint16_t*)ix.d_symmap_q.as<uint16_t>(), ix.key_nsym, ix.key_base, false,
                           ix.q_left.as<int64_t>(), ix.q_right.as<uint64_t>());
    } else if (!ix.sa_sorted && ix.use_fast_search && ix.key_nsym) {
        hipLaunchKernelGGL((q_search_fast_kernel<V, 1>), dim3((unsigned)ceil_div(npat, 256)), dim3(256), 0, s, sa, ix.size, ix.d_text,
                           doc_start, (int)ix.bits, ix.mask, d_blob, d_offs, npat, (const Pivot*)nullptr, 0,
                           ix.d_keys.p ? (const uint64_t*)ix.d_keys.as<uint64_t>() : (const uint64_t*)nullptr,
                           ix.d_keys32.p ? (const uint32_t*)ix.d_keys32.as<uint32_t>() : (const uint32_t*)nullptr,
                           ix.d_keylow.p ? (const void*)ix.d_keylow.p : (const void*)nullptr, ix.key_low_bits, ix.key_low_bytes,
                           (const uint16_t*)ix.d_symmap_q.as<uint16_t>(), ix.key_nsym, ix.key_base, true, ix.q_left.as<int64_t>(),
                           ix.q_right.as<uint64_t>());
    } else {
        hipLaunchKernelGGL((q_search_kernel<V>), dim3((unsigned)ceil_div(npat, 256)), dim3(256), 0, s, sa, ix.size, ix.d_text,
                           doc_start, (int)ix.bits, ix.mask, d_blob, d_offs, npat, ix.q_left.as<int64_t>(),
                           ix.q_right.as<uint64_t>());
    }
    ix.prof.end(t, "q_search", npat * 2 * (uint64_t)bit_width64(ix.size) * 128, s);
}

// ---- OR over the keywords of one key (interface.cpp:78-113): per-document totals by atomics ------
template <typename V>
__global__ __launch_bounds__(256) void q_count_docs_kernel(typename SaOf<V>::ptr sa, uint64_t mask,
                                                           const int64_t* __restrict__ left,
                                                           const uint64_t* __restrict__ hoff, uint64_t npat, uint64_t H,
                                                           unsigned long long* __restrict__ doc_count) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < H; t += stride) {
        uint64_t lo = 0, hi = npat - 1;  // largest j with hoff[j] <= t
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo + 1) / 2;
            if (hoff[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const uint64_t i = (uint64_t)left[lo] + (t - hoff[lo]);
        atomicAdd(&doc_count[(uint64_t)sa[i] & mask], 1ull);
    }
}

struct NonZeroIn {
    const unsigned long long* c;
    __device__ __forceinline__ uint64_t operator()(uint64_t d) const { return c[d] ? 1ull : 0ull; }
};
struct IdRowsOut {  // row r of the union: key = id with the sign bit flipped (unsigned order == signed order)
    const unsigned long long* c;
    const int64_t* ids;
    uint64_t* key;
    uint64_t* val;
    __device__ __forceinline__ void operator()(uint64_t d, uint64_t ex, uint64_t in) const {
        if (in != ex) {
            key[ex] = (uint64_t)ids[d] ^ (1ull << 63);
            val[ex] = (uint64_t)c[d];
        }
    }
};
__global__ __launch_bounds__(256) void q_unflip_kernel(const uint64_t* __restrict__ key, const uint64_t* __restrict__ val,
                                                       uint64_t n, int64_t* __restrict__ ids, int64_t* __restrict__ counts) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    ids[r] = (int64_t)(key[r] ^ (1ull << 63));
    counts[r] = (int64_t)val[r];
}

// ---- highlight spans (reference database.cpp:58-76) -----------------------------------
// ac_automaton::render walks a document once and, for every position where a keyword ends, merges the
// occurrence [begin, end] into a span list: earlier spans starting at or after `begin` are dropped, an
// occurrence that begins inside the previous span extends it, anything else starts a new span.  The
// outcome is the union of all keyword occurrences of the document with OVERLAPPING occurrences fused
// and merely adjacent ones kept apart.  The suffix array already knows every occurrence (offset =
// entry >> bits), so no document has to be re-scanned: occurrences are sorted by (doc, begin) and a
// segmented running maximum of the ends decides where a new span starts.
template <typename V>
__global__ __launch_bounds__(256) void q_expand_occ_kernel(typename SaOf<V>::ptr sa, uint64_t mask, int bits, int obits,
                                                           const int64_t* __restrict__ left,
                                                           const uint64_t* __restrict__ hoff,
                                                           const uint64_t* __restrict__ offs, uint64_t npat, uint64_t H,
                                                           uint64_t* __restrict__ keys, uint64_t* __restrict__ ends) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < H; t += stride) {
        uint64_t lo = 0, hi = npat - 1;  // largest j with hoff[j] <= t
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo + 1) / 2;
            if (hoff[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const auto e = sa[(uint64_t)left[lo] + (t - hoff[lo])];
        const uint64_t doc = (uint64_t)e & mask, off = (uint64_t)e >> bits;
        keys[t] = (doc << obits) | off;
        ends[t] = off + (offs[lo + 1] - offs[lo]);  // one past the last byte of the occurrence
    }
}

// Occurrences by scanning the text itself — used for highlight spans when the suffix array is a
// reference-compat ordering of text with bytes >= 0x80: the reference's QUERY inherits the wrong ranges of
// its binary search there (and so does ours), but its highlighter re-scans the document text with an
// Aho-Corasick automaton (database.cpp:58-76) and therefore reports the true occurrences.  One thread
// per text position, keywords in LDS, a 256-bit first-byte filter rejects almost every position at once.
constexpr int SCAN_MAX_KW = 256;
constexpr int SCAN_MAX_BYTES = 8192;
template <bool FILL>
__global__ __launch_bounds__(256) void q_scan_occ_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                         const uint64_t* __restrict__ doc_start, uint64_t ndocs, int obits,
                                                         const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs,
                                                         uint32_t npat, unsigned long long* __restrict__ counter,
                                                         uint64_t* __restrict__ keys, uint64_t* __restrict__ ends) {
    __shared__ uint8_t s_blob[SCAN_MAX_BYTES];
    __shared__ uint32_t s_off[SCAN_MAX_KW + 1];
    __shared__ uint32_t s_first[8];
    if (threadIdx.x < 8) s_first[threadIdx.x] = 0;
    for (uint32_t i = threadIdx.x; i <= npat; i += 256) s_off[i] = (uint32_t)offs[i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < s_off[npat]; i += 256) s_blob[i] = blob[i];
    if (threadIdx.x < npat) atomicOr(&s_first[blob[s_off[threadIdx.x]] >> 5], 1u << (blob[s_off[threadIdx.x]] & 31));
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += stride) {
        const uint8_t c = text[p];
        if (!((s_first[c >> 5] >> (c & 31)) & 1u)) continue;
        uint64_t d = ~0ull, dend = 0, ds = 0;
        for (uint32_t k = 0; k < npat; ++k) {
            const uint32_t a = s_off[k], m = s_off[k + 1] - a;
            if (s_blob[a] != c || p + m > n) continue;
            uint32_t q = 1;
            while (q < m && text[p + q] == s_blob[a + q]) ++q;
            if (q < m) continue;
            if (d == ~0ull) {  // document of p (looked up once per matching position)
                uint64_t lo = 0, hi = ndocs - 1;
                while (lo < hi) {
                    const uint64_t mid = lo + (hi - lo + 1) / 2;
                    if (doc_start[mid] <= p) lo = mid; else hi = mid - 1;
                }
                d = lo;
                ds = doc_start[d];
                dend = doc_start[d + 1];
            }
            if (p + m > dend) continue;  // would run into the next document
            const unsigned long long slot = atomicAdd(counter, 1ull);
            if (FILL) {
                keys[slot] = (d << obits) | (p - ds);
                ends[slot] = (p - ds) + m;
            }
        }
    }
}

struct DocMax {  // (document, running maximum of occurrence ends) — segmented max
    uint64_t doc, mx;
};
struct OpDocMax {
    __device__ __forceinline__ DocMax operator()(const DocMax& a, const DocMax& b) const {
        return DocMax{b.doc, a.doc == b.doc ? (a.mx > b.mx ? a.mx : b.mx) : b.mx};
    }
};
struct OccIn {
    const uint64_t* keys;
    const uint64_t* ends;
    int obits;
    __device__ __forceinline__ DocMax operator()(uint64_t i) const { return DocMax{keys[i] >> obits, ends[i]}; }
};
struct OccOut {  // head[i] = 1 where a new span starts; incl[i] = running max end inside the document
    const uint64_t* keys;
    int obits;
    uint8_t* head;
    uint64_t* incl;
    __device__ __forceinline__ void operator()(uint64_t i, const DocMax& ex, const DocMax& in) const {
        const uint64_t k = keys[i];
        const uint64_t doc = k >> obits, begin = k & ((1ull << obits) - 1ull);
        head[i] = (uint8_t)((i == 0 || ex.doc != doc || begin >= ex.mx) ? 1 : 0);  // begin >= max end+... : no overlap
        incl[i] = in.mx;
    }
};
struct HeadIn {
    const uint8_t* head;
    __device__ __forceinline__ uint64_t operator()(uint64_t i) const { return head[i]; }
};
struct SpanOut {  // span k starts at occurrence i
    const uint8_t* head;
    const uint64_t* keys;
    uint64_t* span_first;
    uint64_t* span_key;
    uint64_t m;
    __device__ __forceinline__ void operator()(uint64_t i, uint64_t ex, uint64_t in) const {
        if (in != ex) {
            span_first[ex] = i;
            span_key[ex] = keys[i];
        }
        if (i + 1 == m) span_first[in] = m;
    }
};
__global__ __launch_bounds__(256) void q_span_rows_kernel(const uint64_t* __restrict__ span_first,
                                                          const uint64_t* __restrict__ span_key,
                                                          const uint64_t* __restrict__ incl, uint64_t nspans, int obits,
                                                          uint64_t* __restrict__ begin, uint64_t* __restrict__ end,
                                                          uint8_t* __restrict__ dochead) {
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nspans) return;
    begin[k] = span_key[k] & ((1ull << obits) - 1ull);
    end[k] = incl[span_first[k + 1] - 1] - 1;  // inclusive last byte, as ac_automaton's spans
    dochead[k] = (uint8_t)((k == 0 || (span_key[k] >> obits) != (span_key[k - 1] >> obits)) ? 1 : 0);
}
struct DocRowOut {  // row r (document) starts at span k
    const uint64_t* span_key;
    const int64_t* ids;
    int obits;
    uint64_t nspans;
    int64_t* out_ids;
    uint64_t* span_ptr;
    __device__ __forceinline__ void operator()(uint64_t k, uint64_t ex, uint64_t in) const {
        if (in != ex) {
            out_ids[ex] = ids[span_key[k] >> obits];
            span_ptr[ex] = k;
        }
        if (k + 1 == nspans) span_ptr[in] = nspans;
    }
};

template <typename V>
DeviceCsr query_typed(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat, bool with_offsets) {
    hipStream_t s = ix.stream;
    DeviceCsr out;
    out.npat = npat;
    ix.q_rowptr.ensure((npat + 1) * 8);
    if (npat == 0 || ix.size == 0 || ix.width == 0) {
        CDB_HIP(hipMemsetAsync(ix.q_rowptr.p, 0, (npat + 1) * 8, s));
        ix.q_ids.ensure(16);
        ix.q_counts.ensure(16);
        CDB_HIP(hipStreamSynchronize(s));
        return out;
    }
    const auto sa = ix.sa_view<V>();
    ix.q_left.ensure(npat * 8);
    ix.q_right.ensure(npat * 8);  // hit counts
    ix.q_hoff.ensure((npat + 1) * 8);
    int t = 0;
    launch_search<V>(ix, d_blob, d_offs, npat);

    HitsIn2 hin{ix.q_right.as<uint64_t>()};
    // Speculative wavefront rows: when the previous batch of this index went down the wavefront path, buffers
    // sized from it (x 1.5) let this batch run search -> scan -> rows -> scan -> emit without the host learning
    // the hit totals in between; they come back together with the row count in ONE round trip.  Anything
    // that does not fit (a hit list over 64, more hits than the buffers hold) raises a flag and the batch is
    // redone below with the totals known.
    if (!with_offsets && ix.use_wave_rows && ix.ndocs < 0xFFFFFFFFull && ix.q_spec_cap > 0) {
        const uint64_t cap = ix.q_spec_cap;
        scan_totals_device<U2>(s, ix.scan_partials, hin, npat, OpSumMax{}, U2{0, 0});
        const uint64_t nb1 = ceil_div(npat, SC_TILE);
        ix.q_spec.ensure(4 * sizeof(uint64_t));  // {H, maxh, spill flag, nrows}
        CDB_HIP(hipMemcpyAsync(ix.q_spec.p, ix.scan_partials.as<U2>() + nb1, sizeof(U2), hipMemcpyDeviceToDevice, s));
        CDB_HIP(hipMemsetAsync(ix.q_spec.as<uint64_t>() + 2, 0, sizeof(uint64_t), s));
        scan_apply<U2>(s, ix.scan_partials, hin, npat, OpSumMax{}, U2{0, 0}, HitsOut2{ix.q_hoff.as<uint64_t>(), npat});
        ix.q_keys0.ensure(cap * 4);
        ix.q_keys1.ensure(cap * 4);
        ix.q_flags.ensure(npat * 8);
        ix.q_ids.ensure(std::max<uint64_t>(cap, 2) * 8);
        ix.q_counts.ensure(std::max<uint64_t>(cap, 2) * 8);
        t = ix.prof.begin(s);
        hipLaunchKernelGGL((q_wave_rows_kernel<V>), dim3((unsigned)ceil_div(npat, 4)), dim3(256), 0, s, sa, ix.mask,
                           (const int64_t*)ix.q_left.as<int64_t>(), (const uint64_t*)ix.q_right.as<uint64_t>(),
                           (const uint64_t*)ix.q_hoff.as<uint64_t>(), npat, ix.q_keys0.as<uint32_t>(), ix.q_keys1.as<uint32_t>(),
                           ix.q_flags.as<uint64_t>(), cap, ix.q_spec.as<unsigned long long>() + 2);
        ix.prof.end(t, "q_wave_rows", cap * (sizeof(typename SaOf<V>::val) + 8), s);
        NrowsIn nin{ix.q_flags.as<uint64_t>()};
        scan_totals_device<uint64_t>(s, ix.scan_partials, nin, npat, OpAdd{}, (uint64_t)0);
        scan_apply<uint64_t>(s, ix.scan_partials, nin, npat, OpAdd{}, (uint64_t)0, HitsOut{ix.q_rowptr.as<uint64_t>(), npat});
        hipLaunchKernelGGL(q_wave_emit_kernel, dim3((unsigned)ceil_div(npat, 4)), dim3(256), 0, s,
                           (const uint32_t*)ix.q_keys0.as<uint32_t>(), (const uint32_t*)ix.q_keys1.as<uint32_t>(),
                           (const uint64_t*)ix.q_hoff.as<uint64_t>(), (const uint64_t*)ix.q_rowptr.as<uint64_t>(), npat,
                           (const int64_t*)ix.d_ids.as<int64_t>(), ix.q_ids.as<int64_t>(), ix.q_counts.as<int64_t>());
        CDB_HIP(hipMemcpyAsync(ix.q_spec.as<uint64_t>() + 3, ix.q_rowptr.as<uint64_t>() + npat, 8, hipMemcpyDeviceToDevice, s));
        uint64_t h4[4] = {0, 0, 1, 0};
        CDB_HIP(hipMemcpyAsync(h4, ix.q_spec.p, sizeof(h4), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipGetLastError());
        CDB_HIP(hipStreamSynchronize(s));
        if (h4[2] == 0) {
            out.nhits = h4[0];
            out.nrows = h4[3];
            ix.q_spec_cap = std::max<uint64_t>(h4[0] + h4[0] / 2, 4096);
            return out;
        }
        ix.q_spec_cap = 0;  // this batch is not of that kind: the ordinary path below decides again
    }
    const U2 tot = scan_totals<U2>(s, ix.scan_partials, hin, npat, OpSumMax{}, U2{0, 0});
    scan_apply<U2>(s, ix.scan_partials, hin, npat, OpSumMax{}, U2{0, 0}, HitsOut2{ix.q_hoff.as<uint64_t>(), npat});
    const uint64_t H = tot.a, maxh = tot.b;
    out.nhits = H;
    if (H == 0) {
        CDB_HIP(hipMemsetAsync(ix.q_rowptr.p, 0, (npat + 1) * 8, s));
        ix.q_ids.ensure(16);
        ix.q_counts.ensure(16);
        CDB_HIP(hipStreamSynchronize(s));
        return out;
    }
    if (!with_offsets && maxh <= 64 && ix.use_wave_rows && ix.ndocs < 0xFFFFFFFFull && H <= (1ull << 28)) {
        // every pattern's hit list fits one wavefront: sort + run-length encode per pattern in registers
        ix.q_keys0.ensure(H * 4);   // row_doc
        ix.q_keys1.ensure(H * 4);   // row_cnt
        ix.q_flags.ensure(npat * 8);  // rows per pattern
        t = ix.prof.begin(s);
        hipLaunchKernelGGL((q_wave_rows_kernel<V>), dim3((unsigned)ceil_div(npat, 4)), dim3(256), 0, s, sa, ix.mask,
                           (const int64_t*)ix.q_left.as<int64_t>(), (const uint64_t*)ix.q_right.as<uint64_t>(),
                           (const uint64_t*)ix.q_hoff.as<uint64_t>(), npat, ix.q_keys0.as<uint32_t>(), ix.q_keys1.as<uint32_t>(),
                           ix.q_flags.as<uint64_t>(), H, (unsigned long long*)nullptr);
        ix.prof.end(t, "q_wave_rows", H * (sizeof(typename SaOf<V>::val) + 8), s);
        NrowsIn nin{ix.q_flags.as<uint64_t>()};
        // rows <= hits, so the result arrays are sized by H and the number of rows is fetched together with
        // the final synchronisation instead of costing a round trip of its own
        ix.q_ids.ensure(std::max<uint64_t>(H, 2) * 8);
        ix.q_counts.ensure(std::max<uint64_t>(H, 2) * 8);
        scan_totals_device<uint64_t>(s, ix.scan_partials, nin, npat, OpAdd{}, (uint64_t)0);
        scan_apply<uint64_t>(s, ix.scan_partials, nin, npat, OpAdd{}, (uint64_t)0, HitsOut{ix.q_rowptr.as<uint64_t>(), npat});
        hipLaunchKernelGGL(q_wave_emit_kernel, dim3((unsigned)ceil_div(npat, 4)), dim3(256), 0, s,
                           (const uint32_t*)ix.q_keys0.as<uint32_t>(), (const uint32_t*)ix.q_keys1.as<uint32_t>(),
                           (const uint64_t*)ix.q_hoff.as<uint64_t>(), (const uint64_t*)ix.q_rowptr.as<uint64_t>(), npat,
                           (const int64_t*)ix.d_ids.as<int64_t>(), ix.q_ids.as<int64_t>(), ix.q_counts.as<int64_t>());
        uint64_t nrows = 0;
        CDB_HIP(hipMemcpyAsync(&nrows, ix.q_rowptr.as<uint64_t>() + npat, 8, hipMemcpyDeviceToHost, s));
        CDB_HIP(hipGetLastError());
        CDB_HIP(hipStreamSynchronize(s));
        out.nrows = nrows;
        ix.q_spec_cap = std::max<uint64_t>(H + H / 2, 4096);  // the next batch may run speculatively
        return out;
    }
    ix.q_spec_cap = 0;
    const int dbits = (int)ix.bits;
    const int obits = with_offsets ? ix.off_bits : 0;  // offset field of the sort key (offset emission)
    if (with_offsets) ix.q_hitoff.ensure(H * 8);
    // Chunks of patterns whose hit lists fit the scratch budget (16 B of sort scratch per hit); almost
    // always one chunk.  A short pattern over a big corpus can match a large share of the text, and a
    // whole batch of them can exceed any buffer — they are then resolved chunk by chunk.
    std::vector<uint64_t> cut{0, npat};
    if (H > ix.query_hit_budget) {
        std::vector<uint64_t> hoff(npat + 1);
        CDB_HIP(hipMemcpyAsync(hoff.data(), ix.q_hoff.p, (npat + 1) * 8, hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        cut.assign(1, 0);
        uint64_t start = 0;
        for (uint64_t j = 1; j <= npat; ++j) {
            if (hoff[j] - hoff[start] > ix.query_hit_budget && j - 1 > start) {
                cut.push_back(j - 1);
                start = j - 1;
            }
        }
        cut.push_back(npat);
    }
    uint64_t rows_total = 0, hits_done = 0;
    for (size_t c = 0; c + 1 < cut.size(); ++c) {
        const uint64_t j0 = cut[c], j1 = cut[c + 1];
        uint64_t Hc = H;
        if (cut.size() > 2) {
            uint64_t e[2];
            CDB_HIP(hipMemcpyAsync(&e[0], ix.q_hoff.as<uint64_t>() + j0, 8, hipMemcpyDeviceToHost, s));
            CDB_HIP(hipMemcpyAsync(&e[1], ix.q_hoff.as<uint64_t>() + j1, 8, hipMemcpyDeviceToHost, s));
            CDB_HIP(hipStreamSynchronize(s));
            Hc = e[1] - e[0];
        }
        if (Hc == 0) {
            hipLaunchKernelGGL(q_rowptr_kernel, dim3((unsigned)ceil_div(j1 - j0, 256)), dim3(256), 0, s,
                               (const uint64_t*)nullptr, (uint64_t)0, dbits, j0, j1 - j0, rows_total, ix.q_rowptr.as<uint64_t>());
            continue;
        }
        const int jbits = bit_width64(j1 - j0 - 1);
        if (dbits + jbits + obits > 64) throw Error("pattern batch too large for one call");
        ix.q_keys0.ensure(Hc * 8);
        ix.q_keys1.ensure(Hc * 8);
        t = ix.prof.begin(s);
        hipLaunchKernelGGL((q_expand_kernel<V>), dim3((unsigned)ceil_div(Hc, 256)), dim3(256), 0, s, sa, ix.mask, dbits,
                           (int)ix.bits, obits, (const int64_t*)ix.q_left.as<int64_t>(), (const uint64_t*)ix.q_hoff.as<uint64_t>(), j0, j1, Hc,
                           ix.q_keys0.as<uint64_t>());
        ix.prof.end(t, "q_expand", Hc * (sizeof(typename SaOf<V>::val) + 8), s);
        // stable sort of the chunk by (pattern ∘ doc); passes over constant digits are skipped
        const int sel = radix_sort<uint64_t, NoVal>(s, ix.rws, ix.prof, ix.q_keys0.as<uint64_t>(), ix.q_keys1.as<uint64_t>(),
                                                    (NoVal*)nullptr, (NoVal*)nullptr, Hc, 0, dbits + jbits + obits, nullptr);
        const uint64_t* keys = sel == 0 ? ix.q_keys0.as<uint64_t>() : ix.q_keys1.as<uint64_t>();
        uint64_t* spare = sel == 0 ? ix.q_keys1.as<uint64_t>() : ix.q_keys0.as<uint64_t>();

        const uint64_t nbr = ceil_div(Hc, (uint64_t)SC_TILE);
        ix.scan_partials.ensure(scan_partials_slots(nbr) * sizeof(uint64_t));
        t = ix.prof.begin(s);
        hipLaunchKernelGGL(q_run_count_kernel, dim3((unsigned)nbr), dim3(256), 0, s, keys, Hc, obits, ix.scan_partials.as<uint64_t>());
        const uint64_t nrows = scan_totals_from_partials<uint64_t>(s, ix.scan_partials, Hc, OpAdd{}, (uint64_t)0);
        ix.q_flags.ensure((nrows + 1) * 8);  // row_first
        // row keys go to the spare key buffer (nrows <= Hc)
        hipLaunchKernelGGL(q_run_apply_kernel, dim3((unsigned)nbr), dim3(256), 0, s, keys, Hc, obits,
                           (const uint64_t*)ix.scan_partials.as<uint64_t>(), ix.q_flags.as<uint64_t>(), spare,
                           with_offsets ? ix.q_hitoff.as<uint64_t>() + hits_done : (uint64_t*)nullptr);
        ix.prof.end(t, "q_runs", Hc * (16 + (with_offsets ? 8 : 0)) + nrows * 16, s);
        grow_keep(ix.q_ids, (rows_total + nrows) * 8, rows_total * 8, s);
        grow_keep(ix.q_counts, (rows_total + nrows) * 8, rows_total * 8, s);
        if (with_offsets) grow_keep(ix.q_hitptr, (rows_total + nrows + 1) * 8, rows_total * 8, s);
        hipLaunchKernelGGL(q_rows_kernel, dim3((unsigned)ceil_div(nrows, 256)), dim3(256), 0, s,
                           (const uint64_t*)ix.q_flags.as<uint64_t>(), (const uint64_t*)spare, nrows, dbits,
                           (const int64_t*)ix.d_ids.as<int64_t>(), ix.q_ids.as<int64_t>() + rows_total,
                           ix.q_counts.as<int64_t>() + rows_total, hits_done,
                           with_offsets ? ix.q_hitptr.as<uint64_t>() + rows_total : (uint64_t*)nullptr);
        hipLaunchKernelGGL(q_rowptr_kernel, dim3((unsigned)ceil_div(j1 - j0, 256)), dim3(256), 0, s, (const uint64_t*)spare,
                           nrows, dbits, j0, j1 - j0, rows_total, ix.q_rowptr.as<uint64_t>());
        rows_total += nrows;
        hits_done += Hc;
    }
    out.nrows = rows_total;
    if (with_offsets) {
        ix.q_hitptr.ensure((rows_total + 1) * 8);
        CDB_HIP(hipMemcpyAsync(ix.q_hitptr.as<uint64_t>() + rows_total, &hits_done, 8, hipMemcpyHostToDevice, s));
    }
    CDB_HIP(hipMemcpyAsync(ix.q_rowptr.as<uint64_t>() + npat, &rows_total, 8, hipMemcpyHostToDevice, s));
    CDB_HIP(hipGetLastError());
    radix_check_error(s, ix.rws);
    CDB_HIP(hipStreamSynchronize(s));
    return out;
}

template <typename V>
DeviceCsr query_or_typed(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat) {
    hipStream_t s = ix.stream;
    DeviceCsr out;
    out.npat = npat;
    ix.q_ids.ensure(16);
    ix.q_counts.ensure(16);
    if (npat == 0 || ix.size == 0 || ix.width == 0) return out;
    const auto sa = ix.sa_view<V>();
    ix.q_left.ensure(npat * 8);
    ix.q_right.ensure(npat * 8);
    ix.q_hoff.ensure((npat + 1) * 8);
    launch_search<V>(ix, d_blob, d_offs, npat);
    HitsIn hin{ix.q_right.as<uint64_t>()};
    const uint64_t H = scan_totals<uint64_t>(s, ix.scan_partials, hin, npat, OpAdd{}, (uint64_t)0);
    scan_apply<uint64_t>(s, ix.scan_partials, hin, npat, OpAdd{}, (uint64_t)0, HitsOut{ix.q_hoff.as<uint64_t>(), npat});
    out.nhits = H;
    if (H == 0) {
        CDB_HIP(hipStreamSynchronize(s));
        return out;
    }
    DevBuf doc_count;
    doc_count.alloc(ix.ndocs * 8);
    CDB_HIP(hipMemsetAsync(doc_count.p, 0, ix.ndocs * 8, s));
    const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(H, 256), 1u << 20);
    hipLaunchKernelGGL((q_count_docs_kernel<V>), dim3(grid), dim3(256), 0, s, sa, ix.mask,
                       (const int64_t*)ix.q_left.as<int64_t>(), (const uint64_t*)ix.q_hoff.as<uint64_t>(), npat, H,
                       doc_count.as<unsigned long long>());
    NonZeroIn nz{doc_count.as<unsigned long long>()};
    const uint64_t nrows = scan_totals<uint64_t>(s, ix.scan_partials, nz, ix.ndocs, OpAdd{}, (uint64_t)0);
    out.nrows = nrows;
    DevBuf k0, k1, v0, v1;
    k0.alloc(nrows * 8); k1.alloc(nrows * 8); v0.alloc(nrows * 8); v1.alloc(nrows * 8);
    scan_apply<uint64_t>(s, ix.scan_partials, nz, ix.ndocs, OpAdd{}, (uint64_t)0,
                         IdRowsOut{doc_count.as<unsigned long long>(), ix.d_ids.as<int64_t>(), k0.as<uint64_t>(), v0.as<uint64_t>()});
    const int sel = radix_sort<uint64_t, uint64_t>(s, ix.rws, ix.prof, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint64_t>(),
                                                   v1.as<uint64_t>(), nrows, 0, 64, nullptr);
    ix.q_ids.ensure(nrows * 8);
    ix.q_counts.ensure(nrows * 8);
    hipLaunchKernelGGL(q_unflip_kernel, dim3((unsigned)ceil_div(nrows, 256)), dim3(256), 0, s,
                       (const uint64_t*)(sel ? k1 : k0).as<uint64_t>(), (const uint64_t*)(sel ? v1 : v0).as<uint64_t>(), nrows,
                       ix.q_ids.as<int64_t>(), ix.q_counts.as<int64_t>());
    CDB_HIP(hipGetLastError());
    radix_check_error(s, ix.rws);
    CDB_HIP(hipStreamSynchronize(s));
    return out;
}

template <typename V>
SpanResult query_spans_typed(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat,
                             uint64_t total_pattern_bytes) {
    hipStream_t s = ix.stream;
    SpanResult out;
    if (npat == 0 || ix.size == 0 || ix.width == 0) return out;
    const auto sa = ix.sa_view<V>();
    const int obits = ix.width * 8 - (int)ix.bits;  // offset bits of an entry
    const bool by_scan = !ix.sa_sorted;  // reference-compat ordering: occurrences come from the text itself
    uint64_t H = 0;
    DevBuf k0, k1, e0, e1, head, incl, d_cnt;
    if (by_scan) {
        if (npat > SCAN_MAX_KW || total_pattern_bytes > SCAN_MAX_BYTES)
            throw Error("too many highlight keywords for one request on a reference-compatible index");
        d_cnt.alloc(8);
        CDB_HIP(hipMemsetAsync(d_cnt.p, 0, 8, s));
        const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(ix.size, 256), 1u << 16);
        hipLaunchKernelGGL((q_scan_occ_kernel<false>), dim3(grid), dim3(256), 0, s, ix.d_text, ix.size,
                           (const uint64_t*)ix.d_doc_start.as<uint64_t>(), ix.ndocs, obits, d_blob, d_offs, (uint32_t)npat,
                           d_cnt.as<unsigned long long>(), (uint64_t*)nullptr, (uint64_t*)nullptr);
        CDB_HIP(hipMemcpyAsync(&H, d_cnt.p, 8, hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
    } else {
        ix.q_left.ensure(npat * 8);
        ix.q_right.ensure(npat * 8);
        ix.q_hoff.ensure((npat + 1) * 8);
        launch_search<V>(ix, d_blob, d_offs, npat);
        HitsIn hin{ix.q_right.as<uint64_t>()};
        H = scan_totals<uint64_t>(s, ix.scan_partials, hin, npat, OpAdd{}, (uint64_t)0);
        scan_apply<uint64_t>(s, ix.scan_partials, hin, npat, OpAdd{}, (uint64_t)0, HitsOut{ix.q_hoff.as<uint64_t>(), npat});
    }
    out.nhits = H;
    if (H == 0) {
        CDB_HIP(hipStreamSynchronize(s));
        return out;
    }
    if (H > (1ull << 31)) throw Error("too many occurrences for one highlight request");
    k0.alloc(H * 8); k1.alloc(H * 8); e0.alloc(H * 8); e1.alloc(H * 8); head.alloc(H); incl.alloc(H * 8);
    if (by_scan) {
        CDB_HIP(hipMemsetAsync(d_cnt.p, 0, 8, s));
        const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(ix.size, 256), 1u << 16);
        hipLaunchKernelGGL((q_scan_occ_kernel<true>), dim3(grid), dim3(256), 0, s, ix.d_text, ix.size,
                           (const uint64_t*)ix.d_doc_start.as<uint64_t>(), ix.ndocs, obits, d_blob, d_offs, (uint32_t)npat,
                           d_cnt.as<unsigned long long>(), k0.as<uint64_t>(), e0.as<uint64_t>());
    } else {
        const unsigned grid = (unsigned)std::min<uint64_t>(ceil_div(H, 256), 1u << 20);
        hipLaunchKernelGGL((q_expand_occ_kernel<V>), dim3(grid), dim3(256), 0, s, sa, ix.mask, (int)ix.bits, obits,
                           (const int64_t*)ix.q_left.as<int64_t>(), (const uint64_t*)ix.q_hoff.as<uint64_t>(), d_offs, npat, H,
                           k0.as<uint64_t>(), e0.as<uint64_t>());
    }
    const int sel = radix_sort<uint64_t, uint64_t>(s, ix.rws, ix.prof, k0.as<uint64_t>(), k1.as<uint64_t>(), e0.as<uint64_t>(),
                                                   e1.as<uint64_t>(), H, 0, std::min(64, obits + (int)ix.bits), nullptr);
    const uint64_t* keys = (sel ? k1 : k0).as<uint64_t>();
    const uint64_t* ends = (sel ? e1 : e0).as<uint64_t>();
    uint64_t* spare_k = (sel ? k0 : k1).as<uint64_t>();
    uint64_t* spare_e = (sel ? e0 : e1).as<uint64_t>();
    // equal begins of one document: the stable sort keeps them in keyword order; the running maximum
    // makes the order irrelevant
    OccIn oin{keys, ends, obits};
    const DocMax ident{~0ull, 0};
    (void)scan_totals<DocMax>(s, ix.scan_partials, oin, H, OpDocMax{}, ident);
    scan_apply<DocMax>(s, ix.scan_partials, oin, H, OpDocMax{}, ident, OccOut{keys, obits, head.as<uint8_t>(), incl.as<uint64_t>()});
    HeadIn hd{head.as<uint8_t>()};
    const uint64_t nspans = scan_totals<uint64_t>(s, ix.scan_partials, hd, H, OpAdd{}, (uint64_t)0);
    out.nspans = nspans;
    // span_first -> spare_e (nspans + 1 <= H + 1 entries: allocate separately), span_key -> spare_k
    DevBuf span_first, dochead;
    span_first.alloc((nspans + 1) * 8);
    dochead.alloc(nspans);
    scan_apply<uint64_t>(s, ix.scan_partials, hd, H, OpAdd{}, (uint64_t)0,
                         SpanOut{head.as<uint8_t>(), keys, span_first.as<uint64_t>(), spare_k, H});
    ix.q_keys0.ensure(nspans * 8);  // begin
    ix.q_keys1.ensure(nspans * 8);  // end
    hipLaunchKernelGGL(q_span_rows_kernel, dim3((unsigned)ceil_div(nspans, 256)), dim3(256), 0, s,
                       (const uint64_t*)span_first.as<uint64_t>(), (const uint64_t*)spare_k, (const uint64_t*)incl.as<uint64_t>(),
                       nspans, obits, ix.q_keys0.as<uint64_t>(), ix.q_keys1.as<uint64_t>(), dochead.as<uint8_t>());
    HeadIn dh{dochead.as<uint8_t>()};
    const uint64_t ndocs = scan_totals<uint64_t>(s, ix.scan_partials, dh, nspans, OpAdd{}, (uint64_t)0);
    out.ndocs = ndocs;
    ix.q_ids.ensure(ndocs * 8);
    ix.q_rowptr.ensure((ndocs + 1) * 8);
    scan_apply<uint64_t>(s, ix.scan_partials, dh, nspans, OpAdd{}, (uint64_t)0,
                         DocRowOut{spare_k, ix.d_ids.as<int64_t>(), obits, nspans, ix.q_ids.as<int64_t>(), ix.q_rowptr.as<uint64_t>()});
    (void)spare_e;
    CDB_HIP(hipGetLastError());
    radix_check_error(s, ix.rws);
    CDB_HIP(hipStreamSynchronize(s));
    return out;
}

}  // namespace

// ---- one keyword, one wavefront, one launch ---------------------------------------------------------
// string_index::query() is called with ONE keyword (database.cpp:392).  The batched pipeline costs ~12 launches
// and four host round trips for it (~105 us); here a single wavefront does everything: both bounds by a
// 64-ary search (every lane probes one position per round: log65 n = 5-6 rounds of dependent loads instead
// of 2 log2 n = 60), the <= 64 hits are sorted and run-length encoded in registers exactly like
// q_wave_rows_kernel, and the rows go straight into host-mapped memory.  The keyword travels in the kernel
// arguments.  Hit lists of up to 4096 entries are sorted in LDS by the whole workgroup instead; longer ones,
// keywords of more than 120 bytes take the batched path.  On not globally sorted (reference-compat) arrays one
// lane walks the reference's own two bisections instead of the 64-ary search.
struct SingleKw {
    uint32_t len;
    uint8_t bytes[124];
};
constexpr uint32_t SINGLE_MAX_HITS = 4096;
struct SingleOut {
    uint64_t nrows;  // written LAST by the kernel; ~0 = still pending, ~0 - 1 = not answered here (more than SINGLE_MAX_HITS hits)
    uint64_t hits;
    int64_t ids[SINGLE_MAX_HITS];
    int64_t counts[SINGLE_MAX_HITS];
};

// The answer to one keyword (k[0 .. m) in LDS) by one workgroup of 256 threads; every thread enters and leaves together
// (the resident kernel below calls it once per request).  `done` = the word that announces the rows (written last).
template <typename V>
__device__ __forceinline__ void q_single_answer(typename SaOf<V>::ptr sa, uint64_t n, const uint8_t* __restrict__ text,
                                                const uint64_t* __restrict__ doc_start, int bits, uint64_t mask,
                                                const int64_t* __restrict__ ids, const uint8_t* k, uint64_t m,
                                                SingleOut* __restrict__ out, bool sorted, const SingleKeys& sk) {
    __shared__ int64_t s_left;
    __shared__ uint64_t s_hits;
    __shared__ int s_inwin;          // the entries of the hits are already in s_win (fetched with the last probe window)
    __shared__ uint64_t s_win[64];
    __shared__ uint32_t s_doc[SINGLE_MAX_HITS];
    __shared__ int64_t s_rid[SINGLE_MAX_HITS];    // rows are assembled in LDS and leave for the host-mapped block
    __shared__ uint32_t s_rcnt[SINGLE_MAX_HITS];  // with consecutive lanes on consecutive slots
    __shared__ uint32_t s_wcnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) {  // ---- the search is the first wavefront's business
    // three-way compare of the keyword with the suffix at slot M: <0 keyword smaller, 0 keyword is a prefix of
    // the suffix or equal on the common part
    auto probe = [&](int64_t M, bool& le, bool& pref) {
        if (sk.nsym) {  // the kept sort key of slot M decides most probes with ONE load (see q_search_fast_kernel)
            uint64_t key;
            bool known = true;
            if (sk.keys64) {
                key = sk.keys64[M];
            } else {
                const uint64_t hpart = sk.keys32[M];
                // (the low digits are fetched beside the high part, not after it: one round trip instead of two for the
                //  lanes that need them — the wavefront waits for its slowest lane)
                uint64_t lowv = 0;
                if (sk.keylow)
                    lowv = sk.low_bytes == 1 ? (uint64_t)static_cast<const uint8_t*>(sk.keylow)[M]
                                             : (uint64_t)static_cast<const uint16_t*>(sk.keylow)[M];
                if (sk.keylow) {
                    const uint64_t a = sk.klo >> sk.low_bits, b2 = sk.khi >> sk.low_bits;
                    if (hpart < a || hpart > b2 || (hpart > a && hpart < b2)) {
                        key = hpart < a ? 0 : (hpart > b2 ? ~0ull : sk.klo);  // below / above / inside the range
                        known = false;
                        if (hpart < a) { le = false; pref = false; return; }
                        if (hpart > b2) { le = true; pref = false; return; }
                        if (sk.decisive) { le = true; pref = true; return; }
                    } else {
                        key = (hpart << sk.low_bits) | lowv;
                    }
                } else {
                    key = hpart;
                }
            }
            if (known) {
                if (key < sk.klo) { le = false; pref = false; return; }
                if (key > sk.khi) { le = true; pref = false; return; }
                if (sk.decisive) { le = true; pref = true; return; }
            }
            // the suffix starts with the keyword's first symbols and the keyword is longer than the key: the text decides
        }
        const auto e = sa[M];
        const uint64_t d = (uint64_t)e & mask, off = (uint64_t)e >> bits;
        const uint64_t b = doc_start[d] + off, sl = doc_start[d + 1] - b;
        const int c = cmp_common(k, m, text + b, sl);
        le = c < 0 || (c == 0 && m <= sl);
        pref = c == 0 && sl >= m;
    };
    auto shfl64 = [&](int64_t v, int src) -> int64_t {
        const uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)((uint64_t)v >> 32), src);
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    if (!sorted) {
        // reference-compat ordering of text with bytes >= 0x80: the array is not globally sorted, so the answer
        // is whatever the reference's two bisections visit (index.cpp:260-287) — one lane walks exactly them
        if (lane == 0) {
            int64_t L = 0, R = (int64_t)n - 1;
            while (L < R) {
                const int64_t M = L + (R - L) / 2;
                bool le, pf;
                probe(M, le, pf);
                if (le) R = M; else L = M + 1;
            }
            const int64_t lft = L;
            L = lft - 1;
            R = (int64_t)n - 1;
            while (L < R) {
                const int64_t M = L + (R - L + 1) / 2;
                bool le, pf;
                probe(M, le, pf);
                if (pf) L = M; else R = M - 1;
            }
            const int64_t rgt = L + 1;
            s_left = lft;
            s_hits = rgt > lft ? (uint64_t)(rgt - lft) : 0ull;
            s_inwin = 0;
        }
    } else {
    // ---- lower bound (index.cpp:260-274): smallest M in [0, n-1] with keyword <= suffix(M), else n-1
    // (with a key directory: every slot in front of lo0 holds a smaller key, slot hi0 a larger one — same answer)
    int64_t L = 0, R = (int64_t)n - 1;
    if (sk.ranged) {
        L = (int64_t)sk.lo0;
        R = (int64_t)sk.hi0 < R ? (int64_t)sk.hi0 : R;
    }
    while (R - L >= 64) {
        const int64_t M = L + ((R - L) / 65) * (lane + 1) + (((R - L) % 65) * (lane + 1)) / 65;
        bool le, pf;
        probe(M, le, pf);
        const uint64_t b = __ballot(le);
        if (b == 0) {
            L = shfl64(M, 63) + 1;
        } else {
            const int f = __ffsll((unsigned long long)b) - 1;
            R = shfl64(M, f);
            if (f > 0) L = shfl64(M, f - 1) + 1;
        }
    }
    bool in_window = false;  // the whole hit range lies inside the last probe window: no further round trips
    int64_t right_w = 0;
    {
        const int64_t M = L + lane;
        bool le = true, pf = false;
        const bool probed = M < R;
        typename SaOf<V>::val ew = 0;
        if (probed) {
            ew = sa[M];       // (fetched beside the probe: if the hits end inside this window they are already here)
            probe(M, le, pf);  // (slot R itself is the saturated answer)
        }
        const uint64_t b = __ballot(le);
        const int f = __ffsll((unsigned long long)b) - 1;
        // first probed slot at or behind the lower bound that is no match
        const uint64_t nomatch = __ballot(probed && !pf) & ~((1ull << f) - 1ull);
        if (nomatch != 0) {
            in_window = true;
            right_w = L + (__ffsll((unsigned long long)nomatch) - 1);
            const uint32_t lo = __shfl((uint32_t)ew, (lane + f) & 63);
            const uint32_t hi = sizeof(typename SaOf<V>::val) == 8 ? __shfl((uint32_t)((uint64_t)ew >> 32), (lane + f) & 63) : 0u;
            s_win[lane] = ((uint64_t)hi << 32) | lo;  // entries of slots left, left + 1, ... (lanes behind the window: junk)
        }
        L += f;
    }
    const int64_t left = L;
    // ---- prefix upper bound (index.cpp:275-287): the hits are [left, right)
    int64_t right = left;
    if (in_window) {
        right = right_w;
    } else {
        bool le, pf = false;
        const int64_t M = left + lane;
        if (M < (int64_t)n) probe(M, le, pf);
        const uint64_t np = ~__ballot(pf);  // lanes whose slot is no match (or beyond the array)
        if (np != 0) {
            right = left + (__ffsll((unsigned long long)np) - 1);
        } else {  // more than 64 hits: 64-ary search for the first non-match in (left + 63, n]
            int64_t A = left + 64, B = (int64_t)n;  // answer in [A, B], slot B counts as a non-match
            while (B - A >= 64) {
                const int64_t P = A + ((B - A) / 65) * (lane + 1) + (((B - A) % 65) * (lane + 1)) / 65;
                bool l2, p2;
                probe(P, l2, p2);
                const uint64_t nb = __ballot(!p2);
                if (nb == 0) {
                    A = shfl64(P, 63) + 1;
                } else {
                    const int f = __ffsll((unsigned long long)nb) - 1;
                    B = shfl64(P, f);
                    if (f > 0) A = shfl64(P, f - 1) + 1;
                }
            }
            const int64_t P = A + lane;
            bool l2, p2 = false;
            if (P < B) probe(P, l2, p2);
            const uint64_t nb = ~__ballot(p2);
            right = A + (__ffsll((unsigned long long)nb) - 1);
        }
    }
    if (lane == 0) {
        s_left = left;
        s_hits = (uint64_t)(right - left);
        s_inwin = in_window ? 1 : 0;
    }
    }  // sorted
    }  // wave 0
    __syncthreads();
    const int64_t left = s_left;
    const uint64_t hits = s_hits;
    const uint32_t h = (uint32_t)hits;
    if (hits > SINGLE_MAX_HITS) {
        if (tid == 0) {
            out->hits = hits;
            __hip_atomic_store(&out->nrows, ~0ull - 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // "not answered here"
        }
    } else if (h <= 64) {
        // ---- rows (index.cpp:288-322) of a short hit list: sorted and run-length encoded in registers
        if (wave == 0) {
        uint32_t v = 0xFFFFFFFFu;
        if ((uint32_t)lane < h)  // (doc indices fit 32 bits: index.cpp:199)
            v = s_inwin ? (uint32_t)(s_win[lane] & mask) : (uint32_t)((uint64_t)sa[(uint64_t)left + lane] & mask);
#pragma unroll
        for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
            for (int q = kk >> 1; q > 0; q >>= 1) {
                const uint32_t o = __shfl_xor(v, q);
                const bool up = (lane & kk) == 0;
                const bool lower = (lane & q) == 0;
                const uint32_t mn = v < o ? v : o, mx = v < o ? o : v;
                v = (lower == up) ? mn : mx;
            }
        }
        const uint32_t prev = __shfl_up(v, 1);
        const bool head = (uint32_t)lane < h && (lane == 0 || v != prev);
        const uint64_t heads = __ballot(head);
        if (head) {
            const uint32_t r = __popcll(heads & ((1ull << lane) - 1ull));
            const uint64_t later = heads & ~((2ull << lane) - 1ull);
            const uint32_t next = later ? (uint32_t)(__ffsll((unsigned long long)later) - 1) : h;
            out->ids[r] = ids[v];
            out->counts[r] = (int64_t)(next - (uint32_t)lane);
        }
        __threadfence_system();  // the rows are visible to the host before the row count that announces them
        if (lane == 0) {
            out->hits = hits;
            __hip_atomic_store(&out->nrows, (uint64_t)__popcll(heads), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        }  // wave 0
    } else {
    // ---- up to SINGLE_MAX_HITS hits: bitonic sort of the document indices in LDS by the whole workgroup, heads
    // counted per thread chunk, rows written in order
    uint32_t cap = 128;
    while (cap < h) cap <<= 1;
    for (uint32_t i = tid; i < cap; i += 256) s_doc[i] = i < h ? (uint32_t)((uint64_t)sa[(uint64_t)left + i] & mask) : 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t kk = 2; kk <= cap; kk <<= 1) {
        for (uint32_t q = kk >> 1; q > 0; q >>= 1) {
            for (uint32_t i = tid; i < cap; i += 256) {
                const uint32_t partner = i ^ q;
                if (partner > i) {
                    const uint32_t a = s_doc[i], b = s_doc[partner];
                    const bool up = (i & kk) == 0;
                    if ((a > b) == up) {
                        s_doc[i] = b;
                        s_doc[partner] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    // thread t owns the consecutive slots [t * per, (t + 1) * per)
    const uint32_t per = cap / 256 ? cap / 256 : 1;
    const uint32_t b0 = (uint32_t)tid * per;
    uint32_t nheads = 0;
    for (uint32_t i = b0; i < b0 + per && i < h; ++i) nheads += (i == 0 || s_doc[i] != s_doc[i - 1]) ? 1u : 0u;
    // exclusive prefix of nheads over the 256 threads
    uint32_t incl = nheads;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t x = __shfl_up(incl, off);
        if (lane >= off) incl += x;
    }
    if (lane == 63) s_wcnt[wave] = incl;
    __syncthreads();
    uint32_t base_r = incl - nheads;
    for (int w = 0; w < wave; ++w) base_r += s_wcnt[w];
    const uint32_t total_rows = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    for (uint32_t i = b0; i < b0 + per && i < h; ++i) {
        if (i == 0 || s_doc[i] != s_doc[i - 1]) {
            const uint32_t dv = s_doc[i];
            uint32_t e = i + 1;
            while (e < h && s_doc[e] == dv) ++e;  // run length (runs crossing into the next chunk are walked here)
            s_rid[base_r] = ids[dv];
            s_rcnt[base_r] = e - i;
            ++base_r;
        }
    }
    __syncthreads();
    for (uint32_t r = tid; r < total_rows; r += 256) {
        out->ids[r] = s_rid[r];
        out->counts[r] = (int64_t)s_rcnt[r];
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        out->hits = hits;
        __hip_atomic_store(&out->nrows, (uint64_t)total_rows, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    }
    __syncthreads();  // (the LDS state is free for the next request)
}

template <typename V>
__global__ __launch_bounds__(256) void q_single_kernel(typename SaOf<V>::ptr sa, uint64_t n,
                                                      const uint8_t* __restrict__ text,
                                                      const uint64_t* __restrict__ doc_start, int bits, uint64_t mask,
                                                      const int64_t* __restrict__ ids, SingleKw kw, SingleOut* __restrict__ out,
                                                      bool sorted, SingleKeys sk) {
    __shared__ uint8_t s_kw[128];
    const int tid = threadIdx.x;
    if (tid < 128) s_kw[tid] = tid < 124 ? kw.bytes[tid] : (uint8_t)0;
    __syncthreads();
    q_single_answer<V>(sa, n, text, doc_start, bits, mask, ids, s_kw, (uint64_t)kw.len, out, sorted, sk);
}

// ---- the same answer from a RESIDENT workgroup (option resident_query) ----------------------------------------------
// A launch costs ~5 us before the first instruction runs, and a kernel that runs once starts cold (instruction fetch,
// wave start-up): more than the search itself.  With resident_query = 1 one workgroup stays on the device and polls a
// host-mapped mailbox; the host posts the keyword there and polls the answer block like before.  The request is a
// handful of 8-byte words that each carry 7 payload bytes and a 1-byte sequence tag: every word is read atomically, so
// a snapshot whose tags all equal the expected sequence number is consistent — one PCIe round trip per poll, no flag
// word to read first.  The workgroup leaves by itself after ~3 ms without a request (so device-wide synchronisations —
// hipFree, hipDeviceSynchronize of anybody in the process — are held up by at most that) and announces it in `exited`,
// written LAST; the host relaunches it with the next request.  Builds, loads and destroy stop it first (query_resident_stop).
constexpr int RES_WORDS = 22;       // 154 payload bytes: len u32, decisive u8, ranged u8, pad, klo u64, khi u64, lo0 u32, hi0 u32,
                                    // keyword <= 120 bytes
constexpr uint32_t RES_IDLE_POLLS = 2500;
constexpr int64_t RESIDENT_AUTO_GAP_US = 1000;   // resident_query = 2 (automatic): lone keywords closer together than this ...
constexpr uint32_t RESIDENT_AUTO_STREAK = 6;     // ... this many times in a row go to the resident workgroup
constexpr uint32_t RES_MAX_SERVED = 4096;  // it also leaves after this many answers (~30 ms of back-to-back queries): a
                                           // device-wide synchronisation elsewhere in the process is never starved by traffic
struct ResidentBox {
    uint64_t w[RES_WORDS];          // host -> device: request words, (payload 56 bits | tag << 56)
    uint64_t stop;                  // host -> device: leave now
    uint64_t pad0[64 - RES_WORDS - 1];
    uint64_t exited;                // device -> host: nonzero = the workgroup has left (nothing is written after it)
};

template <typename V>
__global__ __launch_bounds__(256) void q_resident_kernel(typename SaOf<V>::ptr sa, uint64_t n, const uint8_t* __restrict__ text,
                                                        const uint64_t* __restrict__ doc_start, int bits, uint64_t mask,
                                                        const int64_t* __restrict__ ids, ResidentBox* __restrict__ box,
                                                        SingleOut* __restrict__ out, bool sorted, SingleKeys sk, uint32_t seq0) {
    __shared__ __attribute__((aligned(8))) uint8_t s_req[RES_WORDS * 8];
    __shared__ int s_state;  // 0 = request in s_req, 1 = leave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t seq = seq0;  // last request answered
    for (;;) {
        if (wave == 0) {  // ---- the first wavefront polls the mailbox: one 8-byte load per lane and poll
            uint32_t idle = 0;
            for (;;) {
                const uint32_t want = (seq + 1u) & 0xFFu;
                uint64_t v = (uint64_t)want << 56;
                if (lane < RES_WORDS) v = __hip_atomic_load(&box->w[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else if (lane == RES_WORDS) v = __hip_atomic_load(&box->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? ~0ull : v;
                const bool stop = __any(lane == RES_WORDS && v == ~0ull);
                const bool fresh = __all(lane >= RES_WORDS || (uint32_t)(v >> 56) == want);
                if (fresh) {
                    if (lane < RES_WORDS) {
#pragma unroll
                        for (int b = 0; b < 7; ++b) s_req[lane * 7 + b] = (uint8_t)(v >> (8 * b));
                    }
                    if (lane == 0) s_state = 0;
                    break;
                }
                if (stop || ++idle > RES_IDLE_POLLS) {
                    if (lane == 0) s_state = 1;
                    break;
                }
            }
        }
        __syncthreads();
        if (s_state == 1) break;
        seq += 1;
        const uint32_t len = *reinterpret_cast<const uint32_t*>(s_req);
        const uint64_t klo = *reinterpret_cast<const uint64_t*>(s_req + 8), khi = *reinterpret_cast<const uint64_t*>(s_req + 16);
        SingleKeys k2 = sk;
        k2.decisive = s_req[4] != 0;
        k2.klo = klo;
        k2.khi = khi;
        k2.ranged = s_req[5] != 0;
        k2.lo0 = *reinterpret_cast<const uint32_t*>(s_req + 24);
        k2.hi0 = *reinterpret_cast<const uint32_t*>(s_req + 28);
        q_single_answer<V>(sa, n, text, doc_start, bits, mask, ids, s_req + 32, (uint64_t)len, out, sorted, k2);
        if (seq - seq0 >= RES_MAX_SERVED) break;  // (uniform; the host starts the next one with its next request)
    }
    if (tid == 0) {
        __threadfence_system();
        __hip_atomic_store(&box->exited, (uint64_t)seq + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// true = answered (rows in freshly malloc'd *ids_out / *counts_out); false = take the batched path
namespace {
void single_empty_rows(Index& ix, int64_t** ids_out, int64_t** counts_out, size_t* nrows) {
    *nrows = 0;
    *ids_out = (int64_t*)std::malloc(8);
    *counts_out = (int64_t*)std::malloc(8);
    if (!*ids_out || !*counts_out) {
        std::free(*ids_out);
        std::free(*counts_out);
        *ids_out = *counts_out = nullptr;
        throw std::bad_alloc();
    }
    ix.qstats.nhits = ix.qstats.nrows = 0;
}
}  // namespace

// starts the resident workgroup unless it is running (ix.mu held); seq0 = the last request already answered
void query_resident_ensure(Index& ix, const SingleKeys& sk) {
    ResidentBox* box = static_cast<ResidentBox*>(ix.h_res);
    if (ix.res_running && __atomic_load_n(&box->exited, __ATOMIC_ACQUIRE) == 0) return;
    if (ix.res_running) CDB_HIP(hipStreamSynchronize(ix.res_stream));  // (it has announced its exit: this returns at once)
    __atomic_store_n(&box->exited, 0ull, __ATOMIC_RELAXED);
    __atomic_store_n(&box->stop, 0ull, __ATOMIC_RELEASE);
    SingleKeys base = sk;
    base.decisive = false;
    base.klo = base.khi = 0;
    ix.res_keys = base;
    const uint32_t seq0 = ix.res_seq - 1u;
    SingleOut* out = static_cast<SingleOut*>(ix.d_single);
    sa_dispatch(ix, [&](auto tag) {
        using T = decltype(tag);
        hipLaunchKernelGGL((q_resident_kernel<T>), dim3(1), dim3(256), 0, ix.res_stream, ix.sa_view<T>(), ix.size, ix.d_text,
                           (const uint64_t*)ix.d_doc_start.as<uint64_t>(), (int)ix.bits, ix.mask, (const int64_t*)ix.d_ids.as<int64_t>(),
                           static_cast<ResidentBox*>(ix.d_res), out, ix.sa_sorted, base, seq0);
    });
    CDB_HIP(hipGetLastError());
    ix.res_running = true;
}

// the resident workgroup reads the index arrays: it leaves before they are replaced or freed (builds, loads, destroy)
void query_resident_stop(Index& ix) {
    if (!ix.res_running || !ix.h_res) return;
    ResidentBox* box = static_cast<ResidentBox*>(ix.h_res);
    __atomic_store_n(&box->stop, 1ull, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(ix.res_stream);
    ix.res_running = false;
}

// Host-side key directory (index_impl.h: h_keydir): one bisection per cell over the kept keys, once per index.
__global__ __launch_bounds__(256) void q_keydir_kernel(SingleKeys sk, uint64_t n, int shift, uint32_t cells, uint32_t* __restrict__ dir) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > cells) return;
    if (i == cells) {
        dir[i] = (uint32_t)n;
        return;
    }
    const uint64_t T = i << shift;
    uint64_t lo = 0, hi = n;  // first slot in [0, n] whose key is >= T
    while (lo < hi) {
        const uint64_t M = lo + (hi - lo) / 2;
        uint64_t key;
        if (sk.keys64) {
            key = sk.keys64[M];
        } else {
            key = sk.keys32[M];
            if (sk.keylow)
                key = (key << sk.low_bits) | (sk.low_bytes == 1 ? (uint64_t)static_cast<const uint8_t*>(sk.keylow)[M]
                                                                : (uint64_t)static_cast<const uint16_t*>(sk.keylow)[M]);
        }
        if (key >= T) hi = M; else lo = M + 1;
    }
    dir[i] = (uint32_t)lo;
}

// builds the directory at the index's first lone keyword (ix.mu held; sk = the kept keys of the index)
void query_keydir_ensure(Index& ix, const SingleKeys& sk) {
    if (ix.keydir_tried || !ix.key_directory) return;
    ix.keydir_tried = true;
    const uint64_t n = ix.size;
    if (!ix.sa_sorted || n < (1ull << 16) || n >= 0xFFFFFFFFull || !sk.nsym) return;
    unsigned __int128 mx = 1;  // keys are numbers below key_base^key_nsym
    for (int q = 0; q < ix.key_nsym; ++q) {
        mx *= (unsigned)ix.key_base;
        if (mx > ((unsigned __int128)1 << 63)) return;
    }
    const int kb = bit_width64((uint64_t)(mx - 1));
    int lg = 0;
    while ((2ull << lg) <= n) ++lg;
    const int bits = std::min(std::min(24, kb), std::max(8, lg - 6));  // ~64 slots per cell, at most 2^24 cells (64 MiB)
    const uint32_t cells = 1u << bits;
    DevBuf d_dir;
    d_dir.alloc(((size_t)cells + 1) * sizeof(uint32_t));
    hipStream_t s = ix.stream;
    hipLaunchKernelGGL(q_keydir_kernel, dim3((unsigned)ceil_div((uint64_t)cells + 1, 256)), dim3(256), 0, s, sk, n, kb - bits, cells,
                       d_dir.as<uint32_t>());
    std::vector<uint32_t> dir((size_t)cells + 1);
    CDB_HIP(hipMemcpyAsync(dir.data(), d_dir.p, dir.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
    ix.h_keydir = std::move(dir);
    ix.keydir_shift = kb - bits;
    ix.keydir_bits = bits;
}

// The lone-keyword path in two halves, so that a caller holding several indexes (shards.hip: one per GPU) can have all
// their kernels in flight before it waits for the first answer.
SingleLaunch query_single_launch(Index& ix, const char* kw, size_t len) {
    if (!ix.use_single_query || ix.width == 0 || ix.size == 0 || len == 0 || len > 120 ||
        ix.ndocs >= 0xFFFFFFFFull)
        return SingleLaunch::NotApplicable;
    hipStream_t s = ix.stream;
    if (!ix.h_single) {
        CDB_HIP(hipHostMalloc(&ix.h_single, sizeof(SingleOut), hipHostMallocMapped));
        CDB_HIP(hipHostGetDevicePointer(&ix.d_single, ix.h_single, 0));
    }
    SingleKw k{};
    k.len = (uint32_t)len;
    std::memcpy(k.bytes, kw, len);
    SingleOut* out = static_cast<SingleOut*>(ix.h_single);
    SingleKeys sk;
    if (ix.sa_sorted && ix.use_fast_search && ix.key_nsym && (ix.d_keys.p || ix.d_keys32.p)) {
        const int kc = (int)std::min<size_t>(len, (size_t)ix.key_nsym);
        uint64_t kwc = 0, kpw = 1;
        bool absent = false;
        for (size_t q = 0; q < len; ++q) {
            const uint64_t c = ix.h_symmap_q[(uint8_t)kw[q]];
            absent |= c == 0;
            if ((int)q < kc) kwc = kwc * ix.key_base + c;
        }
        if (absent) return SingleLaunch::Absent;  // a byte the text never holds: the keyword occurs nowhere
        for (int q = kc; q < ix.key_nsym; ++q) kpw *= ix.key_base;
        sk.keys64 = ix.d_keys.p ? ix.d_keys.as<uint64_t>() : nullptr;
        sk.keys32 = ix.d_keys32.p ? ix.d_keys32.as<uint32_t>() : nullptr;
        sk.keylow = ix.d_keylow.p;
        sk.low_bits = ix.key_low_bits;
        sk.low_bytes = ix.key_low_bytes;
        sk.nsym = ix.key_nsym;
        sk.decisive = len <= (size_t)ix.key_nsym;
        sk.klo = kwc * kpw;
        sk.khi = sk.klo + (kpw - 1);
        query_keydir_ensure(ix, sk);
        if (!ix.h_keydir.empty()) {
            const uint64_t cells = 1ull << ix.keydir_bits;
            const uint64_t c_lo = sk.klo >> ix.keydir_shift, c_hi = sk.khi >> ix.keydir_shift;
            sk.ranged = true;
            sk.lo0 = ix.h_keydir[std::min<uint64_t>(c_lo, cells)];
            sk.hi0 = ix.h_keydir[std::min<uint64_t>(c_hi + 1, cells)];
        }
    }
    out->nrows = ~0ull;
    if (ix.resident_mode == 2) {
        // automatic: lone keywords arriving back to back (the reference's caller resolves a query keyword by keyword,
        // interface.cpp:79-113) switch to the resident workgroup; a pause sends the next keyword through a launch again while the
        // workgroup idles out on its own
        const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        const bool close = now - ix.last_single_ns < RESIDENT_AUTO_GAP_US * 1000;
        ix.last_single_ns = now;
        ix.single_streak = close ? ix.single_streak + 1u : 0u;
        ix.resident_query = ix.single_streak >= RESIDENT_AUTO_STREAK;
    } else {
        ix.resident_query = ix.resident_mode == 1;
    }
    if (ix.resident_query) {
        // ---- post the request to the resident workgroup (started now if it is not there)
        if (!ix.res_stream) CDB_HIP(hipStreamCreateWithFlags(&ix.res_stream, hipStreamNonBlocking));
        if (!ix.h_res) {
            CDB_HIP(hipHostMalloc(&ix.h_res, sizeof(ResidentBox), hipHostMallocMapped));
            std::memset(ix.h_res, 0, sizeof(ResidentBox));
            CDB_HIP(hipHostGetDevicePointer(&ix.d_res, ix.h_res, 0));
        }
        ResidentBox* box = static_cast<ResidentBox*>(ix.h_res);
        uint8_t pay[RES_WORDS * 7] = {0};
        const uint32_t l32 = (uint32_t)len;
        std::memcpy(pay, &l32, 4);
        pay[4] = sk.decisive ? 1 : 0;
        pay[5] = sk.ranged ? 1 : 0;
        std::memcpy(pay + 8, &sk.klo, 8);
        std::memcpy(pay + 16, &sk.khi, 8);
        const uint32_t lo32 = (uint32_t)sk.lo0, hi32 = (uint32_t)sk.hi0;
        std::memcpy(pay + 24, &lo32, 4);
        std::memcpy(pay + 28, &hi32, 4);
        std::memcpy(pay + 32, kw, len);
        const uint32_t seq = ++ix.res_seq;
        for (int i = 0; i < RES_WORDS; ++i) {
            uint64_t wv = (uint64_t)(seq & 0xFFu) << 56;
            for (int b = 0; b < 7; ++b) wv |= (uint64_t)pay[i * 7 + b] << (8 * b);
            __atomic_store_n(&box->w[i], wv, __ATOMIC_RELAXED);
        }
        __atomic_thread_fence(__ATOMIC_RELEASE);
        query_resident_ensure(ix, sk);
        return SingleLaunch::Launched;
    }
    sa_dispatch(ix, [&](auto tag) {
        using T = decltype(tag);
        hipLaunchKernelGGL((q_single_kernel<T>), dim3(1), dim3(256), 0, s, ix.sa_view<T>(), ix.size, ix.d_text,
                           (const uint64_t*)ix.d_doc_start.as<uint64_t>(), (int)ix.bits, ix.mask, (const int64_t*)ix.d_ids.as<int64_t>(), k,
                           static_cast<SingleOut*>(ix.d_single), ix.sa_sorted, sk);
    });
    CDB_HIP(hipGetLastError());
    return SingleLaunch::Launched;
}

// false = the kernel handed the keyword over (too many hits for it): use the batched path
bool query_single_collect(Index& ix, int64_t** ids_out, int64_t** counts_out, size_t* nrows) {
    hipStream_t s = ix.stream;
    SingleOut* out = static_cast<SingleOut*>(ix.h_single);
    // The kernel publishes its row count last (system-scope release) into host-mapped memory: the host polls that word
    // instead of paying for hipStreamSynchronize's wake-up (~10 us of the ~20 us a call used to cost).  The stream
    // keeps its order for whatever is launched next; a kernel that never answers (device error) is left to the
    // ordinary synchronisation after ~2 ms.
    if (ix.resident_query && ix.res_running) {
        ResidentBox* box = static_cast<ResidentBox*>(ix.h_res);
        volatile uint64_t* flag = &out->nrows;
        uint64_t v = ~0ull;
        for (uint64_t spin = 0; spin < (1ull << 28); ++spin) {
            v = __atomic_load_n(flag, __ATOMIC_ACQUIRE);
            if (v != ~0ull) break;
            if ((spin & 63) == 63 && __atomic_load_n(&box->exited, __ATOMIC_ACQUIRE) != 0) {
                // the workgroup left (idle timeout) without seeing this request: start a new one, which finds it posted
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != ~0ull) continue;  // (answered just before it left)
                ix.res_running = false;
                query_resident_ensure(ix, ix.res_keys);
            }
            __builtin_ia32_pause();
        }
        if (v == ~0ull) throw Error("HIP error: the resident query kernel did not answer");
        ix.res_answers++;
    } else {
        ix.launched_answers++;
        volatile uint64_t* flag = &out->nrows;
        uint64_t v = ~0ull;
        for (uint32_t spin = 0; spin < (1u << 22); ++spin) {
            v = __atomic_load_n(flag, __ATOMIC_ACQUIRE);
            if (v != ~0ull) break;
            __builtin_ia32_pause();
        }
        if (v == ~0ull) CDB_HIP(hipStreamSynchronize(s));
    }
    if (out->nrows >= ~0ull - 1) {
        if (out->nrows == ~0ull) throw Error("HIP error: the single-keyword kernel did not answer");
        if (getenv("CDB_DEBUG_SINGLE")) std::fprintf(stderr, "[single] handed over: hits=%llu\n", (unsigned long long)out->hits);
        if (!ix.resident_query) CDB_HIP(hipStreamSynchronize(s));
        return false;
    }
    *nrows = (size_t)out->nrows;
    *ids_out = (int64_t*)std::malloc(std::max<uint64_t>(out->nrows, 1) * 8);  // (released by the caller through cdb_free)
    *counts_out = (int64_t*)std::malloc(std::max<uint64_t>(out->nrows, 1) * 8);
    if (!*ids_out || !*counts_out) {
        std::free(*ids_out);
        std::free(*counts_out);
        *ids_out = *counts_out = nullptr;
        throw std::bad_alloc();
    }
    std::memcpy(*ids_out, out->ids, out->nrows * 8);
    std::memcpy(*counts_out, out->counts, out->nrows * 8);
    ix.qstats.nhits = out->hits;
    ix.qstats.nrows = out->nrows;
    return true;
}

bool query_single_on_device(Index& ix, const char* kw, size_t len, int64_t** ids_out, int64_t** counts_out, size_t* nrows) {
    switch (query_single_launch(ix, kw, len)) {
        case SingleLaunch::NotApplicable: return false;
        case SingleLaunch::Absent: single_empty_rows(ix, ids_out, counts_out, nrows); return true;
        default: return query_single_collect(ix, ids_out, counts_out, nrows);
    }
}

void query_single_empty(Index& ix, int64_t** ids_out, int64_t** counts_out, size_t* nrows) { single_empty_rows(ix, ids_out, counts_out, nrows); }

DeviceCsr query_batch_on_device(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat, bool with_offsets) {
    DeviceCsr r = sa_dispatch(ix, [&](auto tag) { return query_typed<decltype(tag)>(ix, d_blob, d_offs, npat, with_offsets); });
    ix.prof.resolve();
    return r;
}

// ---- $correlation filter + ranking of one key's union on the device (interface.cpp:137-146) ----------
// Rows of query_or (ascending id) whose summed count lies in [lo, hi) are compacted into (~count, id) pairs
// and sorted by a STABLE radix sort: descending $correlation, ties ascending by object id.  (The reference
// ranks with an unstable std::sort, so its order among equal counts is an artefact of introsort — the same
// situation as the suffix array's equal suffixes; ascending id is the canonical form here.)
struct CorrIn {
    const int64_t* counts;
    int64_t lo, hi;
    __device__ __forceinline__ uint64_t operator()(uint64_t r) const { return counts[r] >= lo && counts[r] < hi ? 1ull : 0ull; }
};
struct CorrOut {
    const int64_t* ids;
    const int64_t* counts;
    uint64_t* key;
    uint64_t* val;
    __device__ __forceinline__ void operator()(uint64_t r, uint64_t ex, uint64_t in) const {
        if (in != ex) {
            key[ex] = ~(uint64_t)counts[r];
            val[ex] = (uint64_t)ids[r];
        }
    }
};
__global__ __launch_bounds__(256) void q_ranked_out_kernel(const uint64_t* __restrict__ key, const uint64_t* __restrict__ val,
                                                           uint64_t n, int64_t* __restrict__ ids, int64_t* __restrict__ counts) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    ids[r] = (int64_t)val[r];
    counts[r] = (int64_t)~key[r];
}

// rows (ids ascending) in ix.q_ids / q_counts -> filtered to lo <= count < hi and ranked, in place
static DeviceCsr rank_rows_on_device(Index& ix, DeviceCsr r, int64_t lo, int64_t hi, uint64_t limit) {
    hipStream_t s = ix.stream;
    if (r.nrows == 0) {
        ix.prof.resolve();
        return r;
    }
    CorrIn cin{ix.q_counts.as<int64_t>(), lo, hi};
    const uint64_t m = scan_totals<uint64_t>(s, ix.scan_partials, cin, r.nrows, OpAdd{}, (uint64_t)0);
    DevBuf k0, k1, v0, v1;
    k0.alloc(std::max<uint64_t>(m, 1) * 8); k1.alloc(std::max<uint64_t>(m, 1) * 8);
    v0.alloc(std::max<uint64_t>(m, 1) * 8); v1.alloc(std::max<uint64_t>(m, 1) * 8);
    if (m) {
        scan_apply<uint64_t>(s, ix.scan_partials, cin, r.nrows, OpAdd{}, (uint64_t)0,
                             CorrOut{ix.q_ids.as<int64_t>(), ix.q_counts.as<int64_t>(), k0.as<uint64_t>(), v0.as<uint64_t>()});
        const int sel = radix_sort<uint64_t, uint64_t>(s, ix.rws, ix.prof, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint64_t>(),
                                                       v1.as<uint64_t>(), m, 0, 64, nullptr);
        const uint64_t keep = limit ? std::min<uint64_t>(limit, m) : m;
        hipLaunchKernelGGL(q_ranked_out_kernel, dim3((unsigned)ceil_div(keep, 256)), dim3(256), 0, s,
                           (const uint64_t*)(sel ? k1 : k0).as<uint64_t>(), (const uint64_t*)(sel ? v1 : v0).as<uint64_t>(), keep,
                           ix.q_ids.as<int64_t>(), ix.q_counts.as<int64_t>());
        r.nrows = keep;
    } else {
        r.nrows = 0;
    }
    CDB_HIP(hipGetLastError());
    radix_check_error(s, ix.rws);
    CDB_HIP(hipStreamSynchronize(s));
    ix.prof.resolve();
    return r;
}

DeviceCsr query_ranked_on_device(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat, int64_t lo, int64_t hi,
                                 uint64_t limit) {
    DeviceCsr r = sa_dispatch(ix, [&](auto tag) { return query_or_typed<decltype(tag)>(ix, d_blob, d_offs, npat); });
    return rank_rows_on_device(ix, r, lo, hi, limit);
}

// ---- AND across keys (interface.cpp:114-134) ---------------------------------------------------------------------
// filter() intersects the per-key row lists by object id and adds the counts up.  Every list holds an id at most once
// and ascends by id, so after ONE stable sort of the concatenated lists an id survives iff it heads a run of exactly
// `nlists` equal keys; the run's counts are summed on the way out (compaction by scan).  The $correlation filter and
// the ranking (interface.cpp:137-146) follow on the device as for a single key.
__global__ __launch_bounds__(256) void q_and_flip_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ counts, uint64_t n,
                                                         uint64_t* __restrict__ key, uint64_t* __restrict__ val) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    key[r] = (uint64_t)ids[r] ^ (1ull << 63);  // unsigned order == signed order
    val[r] = (uint64_t)counts[r];
}
struct AndIn {  // 1 where slot i closes a run of nl equal ids
    const uint64_t* key;
    uint64_t nl;
    __device__ __forceinline__ uint64_t operator()(uint64_t i) const { return i + 1 >= nl && key[i + 1 - nl] == key[i] ? 1ull : 0ull; }
};
struct AndOut {
    const uint64_t* key;
    const uint64_t* val;
    uint64_t nl;
    int64_t* ids;
    int64_t* counts;
    __device__ __forceinline__ void operator()(uint64_t i, uint64_t ex, uint64_t in) const {
        if (in == ex) return;
        uint64_t sum = 0;
        for (uint64_t q = 0; q < nl; ++q) sum += val[i - q];
        ids[ex] = (int64_t)(key[i] ^ (1ull << 63));
        counts[ex] = (int64_t)sum;
    }
};

DeviceCsr and_merge_on_device(Index& ix, const std::vector<DeviceRows>& lists, bool ranked, int64_t lo, int64_t hi, uint64_t limit) {
    hipStream_t s = ix.stream;
    DeviceCsr out;
    ix.q_ids.ensure(16);
    ix.q_counts.ensure(16);
    uint64_t total = 0;
    for (const DeviceRows& l : lists) {
        if (l.n == 0) return out;  // an empty list empties the intersection
        total += l.n;
    }
    if (lists.empty()) return out;
    DevBuf k0, k1, v0, v1;
    k0.alloc(total * 8); k1.alloc(total * 8); v0.alloc(total * 8); v1.alloc(total * 8);
    uint64_t at = 0;
    for (const DeviceRows& l : lists) {
        hipLaunchKernelGGL(q_and_flip_kernel, dim3((unsigned)ceil_div(l.n, 256)), dim3(256), 0, s, l.d_ids, l.d_counts, l.n,
                           k0.as<uint64_t>() + at, v0.as<uint64_t>() + at);
        at += l.n;
    }
    const int sel = radix_sort<uint64_t, uint64_t>(s, ix.rws, ix.prof, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint64_t>(), v1.as<uint64_t>(),
                                                   total, 0, 64, nullptr);
    const uint64_t* key = (sel ? k1 : k0).as<uint64_t>();
    const uint64_t* val = (sel ? v1 : v0).as<uint64_t>();
    AndIn ain{key, (uint64_t)lists.size()};
    const uint64_t nrows = scan_totals<uint64_t>(s, ix.scan_partials, ain, total, OpAdd{}, (uint64_t)0);
    ix.q_ids.ensure(std::max<uint64_t>(nrows, 2) * 8);
    ix.q_counts.ensure(std::max<uint64_t>(nrows, 2) * 8);
    scan_apply<uint64_t>(s, ix.scan_partials, ain, total, OpAdd{}, (uint64_t)0,
                         AndOut{key, val, (uint64_t)lists.size(), ix.q_ids.as<int64_t>(), ix.q_counts.as<int64_t>()});
    out.nrows = nrows;
    CDB_HIP(hipGetLastError());
    radix_check_error(s, ix.rws);
    CDB_HIP(hipStreamSynchronize(s));
    if (ranked) return rank_rows_on_device(ix, out, lo, hi, limit);
    ix.prof.resolve();
    return out;
}

DeviceCsr query_or_on_device(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat) {
    DeviceCsr r = sa_dispatch(ix, [&](auto tag) { return query_or_typed<decltype(tag)>(ix, d_blob, d_offs, npat); });
    ix.prof.resolve();
    return r;
}

SpanResult query_spans_on_device(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat,
                                 uint64_t total_pattern_bytes) {
    SpanResult r = sa_dispatch(ix, [&](auto tag) { return query_spans_typed<decltype(tag)>(ix, d_blob, d_offs, npat, total_pattern_bytes); });
    ix.prof.resolve();
    return r;
}

}  // namespace cdb
