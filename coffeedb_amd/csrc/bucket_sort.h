// bucket_sort.h — second half of the HYBRID initial sort of the suffix-array build: after a few global radix
// passes have partitioned the records into buckets that fit a workgroup's LDS, every bucket is finished where it
// lies — loaded once, LSD-sorted in LDS on the rest of its key, and written once as final suffix-array entries,
// group flags and (optionally) the kept search keys.
//
// Why: the LSD sort of radix_sort.h moves every record through HBM once per 8 key bits (C1: 40-bit keys = 5 passes
// x 18 B per suffix); the bucket sort replaces the passes over the low `rbits` bits — and the separate flag kernel —
// by ONE read and ONE write of the records (the "LDS-staged key buckets" of the north star).  Reference
// counterpart: the comparison-sorted leaves of the reference's MSD radix (index.cpp:86-95), which also finish small
// buckets locally.
//
// Key coding (radix_sort.h: rs_hyb_key): the dense key K of a suffix is split as b = K / w (bucket, < NB <= 2^(8 G))
// and r = K % w (< 2^rbits).  G global passes sort by b (its lowest `lead` digits travel in the auxiliary byte(s), the
// rest sits above r in the 32-bit key); the last of them also records where every bucket starts.
//
// Work assignment: the array is cut into nominal windows of STEP records; a workgroup owns the buckets that START in
// its window and works through them in rounds of as many consecutive buckets as fit its capacity (one big bucket,
// or hundreds of small ones — then the bucket number relative to the round's first bucket becomes the top digits of
// the local sort key).  A bucket larger than the capacity is reported; the build then falls back to the plain LSD sort.
#pragma once
#include "radix_sort.h"
#include "scan.h"

namespace cdb {

struct HybridPlan {
    bool ok = false;
    int G = 0;          // global passes (digits of b)
    int lead = 0;       // ... of which on auxiliary digits (1: u8, 2: u16)
    int rbits = 0;      // bits of r
    uint64_t w = 0;     // bucket width in key space
    uint64_t magic = 0; // floor(2^64 / w)
    uint64_t nb = 0;    // number of buckets
    int cap = 0;        // records a workgroup can finish at once
};

struct BucketSortParams {
    uint64_t n;
    uint64_t nb;        // buckets; bstart has nb + 1 entries (bstart[nb] = n)
    uint64_t w;         // K = b * w + r
    int rbits, lead_bits;
    uint32_t kbase;     // key base (alphabet + 1): K % kbase == 0 <=> the suffix ends inside the key
    uint64_t kmagic;    // floor(2^64 / kbase) + 1, 0 for a power of two
    int out_low_bits;   // kept keys: k32 = K >> out_low_bits, low = K & (2^out_low_bits - 1)
    uint64_t step;      // nominal window
};

struct OpMinU64 {
    __device__ __forceinline__ uint64_t operator()(const uint64_t& a, const uint64_t& b) const { return a < b ? a : b; }
};
// reverse min-scan over the bucket-start table: a bucket nobody wrote to (empty) starts where the next one does
template <typename P>
struct BStartRevIn {
    const P* t;
    uint64_t nb, n;
    __device__ __forceinline__ uint64_t operator()(uint64_t i) const {  // i = 0 .. nb  <->  bucket nb - i
        if (i == 0) return n;
        const P v = t[nb - i];
        return v == (P)~(P)0 ? ~0ull : (uint64_t)v;
    }
};
template <typename P>
struct BStartRevOut {
    P* t;
    uint64_t nb;
    __device__ __forceinline__ void operator()(uint64_t i, uint64_t, uint64_t incl) const { t[nb - i] = (P)incl; }
};

// first bucket whose start is >= pos (bstart is non-decreasing, bstart[nb] = n)
template <typename P>
__device__ __forceinline__ uint64_t bs_lower_bound(const P* __restrict__ bstart, uint64_t nb, uint64_t pos) {
    uint64_t lo = 0, hi = nb;  // answer in [0, nb]
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if ((uint64_t)bstart[mid] < pos) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// V = entry type, W = auxiliary digit type (u8 / u16), P = position type of the bucket table, KW = type of the kept
// low digits (u8 / u16).  NT x IPT = capacity.
template <typename V, typename W, typename P, typename KW, int NT, int IPT>
__global__ __launch_bounds__(NT) void bs_local_sort_kernel(uint32_t* __restrict__ k32, V* __restrict__ ent, W* __restrict__ aux,
                                                           const P* __restrict__ bstart, BucketSortParams pr,
                                                           uint8_t* __restrict__ flags, KW* __restrict__ keylow_out,
                                                           unsigned long long* __restrict__ oversize /*[0] count, [1] largest*/) {
    constexpr int CAP = NT * IPT;
    constexpr int NW = NT / 64;
    constexpr int WCHUNK = 64 * IPT;
    constexpr size_t STAGE = (sizeof(V) > 4 ? sizeof(V) : 4) * (size_t)CAP;
    __shared__ __attribute__((aligned(16))) unsigned char s_stage[STAGE];
    __shared__ uint32_t s_whist[NW][256];
    __shared__ uint32_t s_tstart[256];
    __shared__ uint32_t s_wsum[4];
    uint32_t* s_keys = reinterpret_cast<uint32_t*>(s_stage);
    V* s_vals = reinterpret_cast<V*>(s_stage);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wbase = wave * WCHUNK + lane;
    const uint32_t rmask = pr.rbits >= 32 ? 0xFFFFFFFFu : ((1u << pr.rbits) - 1u);
    const uint32_t lmask = (1u << pr.lead_bits) - 1u;
    const uint64_t max_rel = pr.rbits >= 32 ? 1ull : (1ull << (32 - pr.rbits));  // buckets one round can tell apart

    const uint64_t win_lo = (uint64_t)blockIdx.x * pr.step;
    const uint64_t win_hi = win_lo + pr.step < pr.n ? win_lo + pr.step : pr.n;
    uint64_t b = bs_lower_bound(bstart, pr.nb, win_lo);
    const uint64_t b_end = bs_lower_bound(bstart, pr.nb, win_hi);  // buckets [b, b_end) start in this window
    while (b < b_end) {
        // ---- one round: buckets [b, b1) with at most CAP records and at most max_rel buckets
        const uint64_t lo = (uint64_t)bstart[b];
        uint64_t b1;
        {
            // largest b1 in (b, b_end] with bstart[b1] - lo <= CAP (bisect), capped by max_rel
            uint64_t l = b, h = b_end;  // invariant: bstart[l] - lo <= CAP
            if (h - b > max_rel) h = b + max_rel;
            while (l < h) {
                const uint64_t mid = l + (h - l + 1) / 2;
                if ((uint64_t)bstart[mid] - lo <= (uint64_t)CAP) l = mid; else h = mid - 1;
            }
            b1 = l;
        }
        if (b1 == b) {  // bucket b alone exceeds the capacity: report it (the host falls back to the plain sort)
            if (tid == 0) {
                atomicAdd(&oversize[0], 1ull);
                atomicMax(&oversize[1], (unsigned long long)((uint64_t)bstart[b + 1] - lo));
            }
            b += 1;
            continue;
        }
        const uint32_t m = (uint32_t)((uint64_t)bstart[b1] - lo);
        const uint32_t nrel = (uint32_t)(b1 - b);
        if (m == 0) {  // (only empty buckets)
            b = b1;
            continue;
        }
        int lbits = pr.rbits;
        if (nrel > 1) lbits += 32 - __clz(nrel - 1);
        const int npass = lbits > 8 ? (lbits + 7) / 8 : 1;
        // ---- load (wave-striped: every wave owns a contiguous chunk, which keeps the local sort stable)
        uint32_t lk[IPT];
        V val[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t li = wbase + j * 64;
            lk[j] = 0xFFFFFFFFu;
            val[j] = V(0);
            if (li < m) {
                const uint32_t k = k32[lo + li];
                // bucket of the record relative to the round's first one (bucket numbers fit 32 bits: nb <= 2^24)
                const uint32_t brel = (((pr.rbits >= 32 ? 0u : (k >> pr.rbits)) << pr.lead_bits) | ((uint32_t)aux[lo + li] & lmask)) - (uint32_t)b;
                lk[j] = (pr.rbits >= 32 ? 0u : (brel << pr.rbits)) | (k & rmask);
                val[j] = ent[lo + li];
            }
        }
        // ---- LSD passes in LDS (positions < CAP < 2^16: two per register)
        static_assert(IPT % 2 == 0 && NT * IPT < 65536, "packed positions");
        uint32_t pp[IPT / 2];
        auto get_pos = [&](int j) -> uint32_t { return (j & 1) ? (pp[j >> 1] >> 16) : (pp[j >> 1] & 0xFFFFu); };
        auto set_pos = [&](int j, uint32_t v) { pp[j >> 1] = (j & 1) ? ((pp[j >> 1] & 0xFFFFu) | (v << 16)) : ((pp[j >> 1] & 0xFFFF0000u) | v); };
        for (int p = 0; p < npass; ++p) {
            for (int i = tid; i < NW * 256; i += NT) (&s_whist[0][0])[i] = 0;
            __syncthreads();  // (also: the staging buffer of the previous pass has been read back)
            const int sh = 8 * p;
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                // slots behind the round's records hold all-ones keys, i.e. digit 255 in every pass, and the largest
                // indices of their wave's chunk: they stay behind every real record
                const uint32_t d = (lk[j] >> sh) & 255u;
                set_pos(j, atomicAdd(&s_whist[wave][d], 1u));  // rank inside the wave (lane order: radix_sort.h ATOMRANK)
            }
            __syncthreads();
            uint32_t cnt = 0, incl = 0;
            if (tid < 256) {
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const uint32_t t = s_whist[w][tid];
                    s_whist[w][tid] = cnt;
                    cnt += t;
                }
                incl = cnt;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t v = __shfl_up(incl, off);
                    if (lane >= off) incl += v;
                }
                if (lane == 63) s_wsum[wave] = incl;
            }
            __syncthreads();
            if (tid < 256) {
                uint32_t wpre = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (w < wave) wpre += s_wsum[w];
                s_tstart[tid] = wpre + incl - cnt;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const uint32_t d = (lk[j] >> sh) & 255u;
                const uint32_t at = s_tstart[d] + s_whist[wave][d] + get_pos(j);
                set_pos(j, at);
                s_keys[at] = lk[j];
            }
            __syncthreads();
            if (p + 1 < npass) {
#pragma unroll
                for (int j = 0; j < IPT; ++j) lk[j] = s_keys[wbase + j * 64];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < IPT; ++j) s_vals[get_pos(j)] = val[j];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < IPT; ++j) val[j] = s_vals[wbase + j * 64];
                // (the next pass starts with a barrier before the staging buffer is written again)
            }
        }
        // ---- the round's records are sorted (keys in LDS): flags and kept keys, then the entries
        // (after pass 0 the padding slots sit at positions >= m and their keys are all ones)
#pragma unroll 2
        for (int j = 0; j < IPT; ++j) {
            const uint32_t i = j * NT + tid;
            if (i < m) {
                const uint32_t x = s_keys[i];
                const bool head = i == 0 || s_keys[i - 1] != x;
                const bool tail = i + 1 == m || s_keys[i + 1] != x;
                const uint64_t K = (b + (uint64_t)(pr.rbits >= 32 ? 0u : (x >> pr.rbits))) * pr.w + (uint64_t)(x & rmask);
                const bool exhausted = pr.kmagic ? (K - __umul64hi(K, pr.kmagic) * pr.kbase) == 0 : (K & (uint64_t)(pr.kbase - 1u)) == 0;
                flags[lo + i] = (uint8_t)((head ? 1 : 0) | ((!(head && tail) && !exhausted) ? 2 : 0));
                if (keylow_out) {  // kept search keys in the layout of the plain split sort
                    k32[lo + i] = (uint32_t)(K >> pr.out_low_bits);
                    keylow_out[lo + i] = (KW)(K & ((1ull << pr.out_low_bits) - 1ull));
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < IPT; ++j) s_vals[get_pos(j)] = val[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t i = j * NT + tid;
            if (i < m) ent[lo + i] = s_vals[i];
        }
        __syncthreads();  // the next round reuses the staging buffer
        b = b1;
    }
}

// Finishes a hybrid sort in place: k32 / ent / aux hold the records sorted by bucket, bstart the (raw) table of the
// last global pass.  Afterwards ent = suffix-array entries in key order, flags = group flags; with keep_keys k32 /
// keylow = the sorted keys in split layout (keylow may alias aux when the types agree).  Returns false when a
// bucket did not fit (nothing usable was produced).
template <typename V, typename W, typename P, typename KW>
bool bucket_sort_finish(hipStream_t s, Profiler& prof, DevBuf& scan_partials, uint32_t* k32, V* ent, W* aux, P* bstart,
                        const HybridPlan& plan, uint64_t n, uint32_t kbase, uint64_t kmagic, int out_low_bits, uint8_t* flags,
                        KW* keylow_out, uint64_t* largest_bucket) {
    // empty buckets inherit the start of the next non-empty one
    BStartRevIn<P> rin{bstart, plan.nb, n};
    scan_totals_device<uint64_t>(s, scan_partials, rin, plan.nb + 1, OpMinU64{}, ~0ull);
    scan_apply<uint64_t>(s, scan_partials, rin, plan.nb + 1, OpMinU64{}, ~0ull, BStartRevOut<P>{bstart, plan.nb});
    DevBuf d_over;
    d_over.alloc(2 * sizeof(uint64_t));
    CDB_HIP(hipMemsetAsync(d_over.p, 0, 2 * sizeof(uint64_t), s));
    BucketSortParams pr{n, plan.nb, plan.w, plan.rbits, 8 * plan.lead, kbase, kmagic, out_low_bits, 0};
    int t = prof.begin(s);
    const bool big = plan.cap > 4096;
    if (big) {
        constexpr int IPT = sizeof(V) == 8 ? 12 : 18;
        pr.step = (uint64_t)1024 * 16;
        hipLaunchKernelGGL((bs_local_sort_kernel<V, W, P, KW, 1024, IPT>), dim3((unsigned)ceil_div(n, pr.step)), dim3(1024), 0, s, k32, ent,
                           aux, (const P*)bstart, pr, flags, keylow_out, d_over.as<unsigned long long>());
    } else {
        pr.step = 2048;
        hipLaunchKernelGGL((bs_local_sort_kernel<V, W, P, KW, 256, 16>), dim3((unsigned)ceil_div(n, pr.step)), dim3(256), 0, s, k32, ent, aux,
                           (const P*)bstart, pr, flags, keylow_out, d_over.as<unsigned long long>());
    }
    prof.end(t, "sa_bucket_sort", n * (2 * (4 + sizeof(V)) + sizeof(W) + 1 + (keylow_out ? sizeof(KW) : 0)), s);
    uint64_t over[2] = {0, 0};
    CDB_HIP(hipMemcpyAsync(over, d_over.p, sizeof(over), hipMemcpyDeviceToHost, s));
    CDB_HIP(hipGetLastError());
    CDB_HIP(hipStreamSynchronize(s));
    if (largest_bucket) *largest_bucket = over[1];
    return over[0] == 0;
}

constexpr int BS_CAP_BIG32 = 1024 * 18, BS_CAP_BIG64 = 1024 * 12, BS_CAP_SMALL = 256 * 16;

}  // namespace cdb
