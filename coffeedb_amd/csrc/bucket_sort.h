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

// Fully unrolled per-slot loops let the scheduler hoist every load of a phase to its front, which multiplies the
// registers in flight by the slots per thread; a scheduling fence after every four slots keeps a phase at "four slots
// in flight" (enough to cover LDS latency, a quarter of the registers).
#define BS_FENCE(j)                                         \
    do {                                                    \
        if (((j) & 3) == 3) __builtin_amdgcn_sched_barrier(0); \
    } while (0)

constexpr int BS_IPT_BIG = 26;                     // 1024 x 26 = 26 Ki records per round: 104 KB of staging
constexpr int BS_CAP_BIG = 1024 * BS_IPT_BIG, BS_CAP_SMALL = 256 * 16;

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
// low digits (u8 / u16).  NT x IPT = capacity (< 2^16 records).
//
// A record is held as TWO registers — the local key and (position << 16 | index of the record in the round's input) —
// and the entries themselves never pass through registers: they are read once, coalesced, into the staging buffer
// after the keys are done, and leave through an LDS gather by the sorted indices.  That is what lets a workgroup
// finish 32 Ki records (128 KB of staging + 16 KB of counters) without spilling.
template <typename V, typename W, typename P, typename KW, int NT, int IPT>
__global__ __launch_bounds__(NT) void bs_local_sort_kernel(uint32_t* __restrict__ k32, V* __restrict__ ent, W* __restrict__ aux,
                                                           const P* __restrict__ bstart, BucketSortParams pr,
                                                           uint8_t* __restrict__ flags, KW* __restrict__ keylow_out,
                                                           unsigned long long* __restrict__ oversize /*[0] count, [1] largest*/) {
    constexpr int CAP = NT * IPT;
    static_assert(CAP <= 65536, "positions and indices are 16-bit");
    constexpr int NW = NT / 64;
    constexpr int WCHUNK = 64 * IPT;
    constexpr int VCAP = sizeof(V) > 4 ? CAP / 2 : CAP;  // entries staged per sweep of the final gather
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[CAP];  // keys, then 16-bit indices, then entries
    __shared__ uint32_t s_whist[NW][256];
    __shared__ uint32_t s_tstart[256];
    __shared__ uint32_t s_wsum[4];
    __shared__ unsigned long long s_round[3];  // {first bucket, one past the last bucket, records} of the round
    uint16_t* s_idx = reinterpret_cast<uint16_t*>(s_keys);
    V* s_vals = reinterpret_cast<V*>(s_keys);
    const int tid0 = threadIdx.x, lane = tid0 & 63, wave = tid0 >> 6;
    const uint32_t wbase0 = wave * WCHUNK + lane;
    const uint32_t rmask = pr.rbits >= 32 ? 0xFFFFFFFFu : ((1u << pr.rbits) - 1u);
    const uint32_t lmask = (1u << pr.lead_bits) - 1u;
    const uint64_t max_rel = pr.rbits >= 32 ? 1ull : (1ull << (32 - pr.rbits));  // buckets one round can tell apart

    const uint64_t win_lo = (uint64_t)blockIdx.x * pr.step;
    const uint64_t win_hi = win_lo + pr.step < pr.n ? win_lo + pr.step : pr.n;
    uint64_t b = 0, b_end = 0;
    uint32_t tid = tid0, wbase = wbase0;
    if (tid == 0) {
        s_round[0] = bs_lower_bound(bstart, pr.nb, win_lo);
        s_round[1] = bs_lower_bound(bstart, pr.nb, win_hi);  // buckets [b, b_end) start in this window
    }
    __syncthreads();
    b = s_round[0];
    b_end = s_round[1];
    while (b < b_end) {
        // ---- one round: buckets [b, b1) with at most CAP records and at most max_rel buckets (found by one thread)
        // (opaque copies: otherwise the compiler hoists the 2 x IPT per-slot index and address computations out of
        //  the round loop and keeps them alive — in scratch memory — across it)
        tid = tid0;
        wbase = wbase0;
        asm volatile("" : "+v"(tid), "+v"(wbase));
        __syncthreads();
        if (tid == 0) {
            const uint64_t lo0 = (uint64_t)bstart[b];
            uint64_t l = b, h = b_end;  // invariant: bstart[l] - lo0 <= CAP
            if (h - b > max_rel) h = b + max_rel;
            while (l < h) {
                const uint64_t mid = l + (h - l + 1) / 2;
                if ((uint64_t)bstart[mid] - lo0 <= (uint64_t)CAP) l = mid; else h = mid - 1;
            }
            s_round[1] = l;
            s_round[2] = (uint64_t)bstart[l] - lo0;
            if (l == b) {  // bucket b alone exceeds the capacity: report it (the host falls back to the plain sort)
                atomicAdd(&oversize[0], 1ull);
                atomicMax(&oversize[1], (unsigned long long)((uint64_t)bstart[b + 1] - lo0));
            }
        }
        __syncthreads();
        const uint64_t b1 = s_round[1];
        const uint32_t m = (uint32_t)s_round[2];
        if (b1 == b) {
            b += 1;
            continue;
        }
        if (m == 0) {  // (only empty buckets)
            b = b1;
            continue;
        }
        const uint64_t lo = (uint64_t)bstart[b];
        const uint32_t nrel = (uint32_t)(b1 - b);
        int lbits = pr.rbits;
        if (nrel > 1) lbits += 32 - __clz(nrel - 1);
        const int npass = lbits > 8 ? (lbits + 7) / 8 : 1;
        // ---- load (wave-striped: every wave owns a contiguous chunk, which keeps the local sort stable)
        uint32_t lk[IPT], pi[IPT];  // local key; (position << 16) | index of the record in the round's input
        {
            const uint32_t* kp = k32 + lo;
            const W* ap = aux + lo;
            const uint32_t b32 = (uint32_t)b;
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const uint32_t li = wbase + j * 64;
                lk[j] = 0xFFFFFFFFu;
                if (li < m) {
                    const uint32_t k = kp[li];
                    // bucket of the record relative to the round's first one (bucket numbers fit 32 bits: nb <= 2^24)
                    const uint32_t brel = (((pr.rbits >= 32 ? 0u : (k >> pr.rbits)) << pr.lead_bits) | ((uint32_t)ap[li] & lmask)) - b32;
                    lk[j] = (pr.rbits >= 32 ? 0u : (brel << pr.rbits)) | (k & rmask);
                }
                BS_FENCE(j);
            }
            // (the input indices are derived from an opaque copy of the lane's base: as the same values as the load
            //  addresses above they would be kept zero-extended to 64 bits, two registers per slot, across the passes)
            uint32_t wb2 = wbase;
            asm volatile("" : "+v"(wb2));
#pragma unroll
            for (int j = 0; j < IPT; ++j) pi[j] = wb2 + j * 64;
        }
        // ---- LSD passes in LDS
        for (int p = 0; p < npass; ++p) {
            for (int i = tid; i < NW * 256; i += NT) (&s_whist[0][0])[i] = 0;
            __syncthreads();  // (also: the staging buffer of the previous pass has been read back)
            const int sh = 8 * p;
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                // slots behind the round's records hold all-ones keys, i.e. digit 255 in every pass, and the largest
                // indices of their wave's chunk: they stay behind every real record
                const uint32_t d = (lk[j] >> sh) & 255u;
                // rank inside the wave (same-address LDS atomics of one instruction complete in lane order: radix_sort.h)
                pi[j] = (pi[j] & 0xFFFFu) | (atomicAdd(&s_whist[wave][d], 1u) << 16);
                BS_FENCE(j);
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
                // (the digit and the counter address are recomputed from an opaque copy of the key: shared with the
                //  ranking loop above they would stay in registers across the barriers, two more per slot)
                uint32_t kx = lk[j];
                asm volatile("" : "+v"(kx));
                const uint32_t d = (kx >> sh) & 255u;
                const uint32_t at = s_tstart[d] + s_whist[wave][d] + (pi[j] >> 16);
                pi[j] = (pi[j] & 0xFFFFu) | (at << 16);
                s_keys[at] = lk[j];
                BS_FENCE(j);
            }
            __syncthreads();
            if (p + 1 < npass) {
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    lk[j] = s_keys[wbase + j * 64];
                    BS_FENCE(j);
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    s_idx[pi[j] >> 16] = (uint16_t)pi[j];
                    BS_FENCE(j);
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    pi[j] = s_idx[wbase + j * 64];
                    BS_FENCE(j);
                }
                // (the next pass starts with a barrier before the staging buffer is written again)
            }
        }
        // ---- the round's records are sorted (keys in LDS): flags and kept keys
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
        // ---- the entries: sorted position -> input index, then the input window is staged (coalesced) and gathered
        // in LDS.  8-byte entries are staged in two halves of the window.
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
                    lk[j] = pi[j] & 0xFFFFu;
                    BS_FENCE(j);
                }  // (keys are done: their registers carry the indices)
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
                    s_idx[pi[j] >> 16] = (uint16_t)lk[j];
                    BS_FENCE(j);
                }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
                    lk[j] = s_idx[j * NT + tid];
                    BS_FENCE(j);
                }  // input index of the record at sorted position j * NT + tid
        __syncthreads();
        V outv[IPT];
        for (uint32_t half = 0; half < (uint32_t)(CAP / VCAP); ++half) {
            const uint32_t h0 = half * VCAP;
            if (h0 >= m) break;
            for (uint32_t i = tid; i < (uint32_t)VCAP && h0 + i < m; i += NT) s_vals[i] = ent[lo + h0 + i];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const uint32_t src = lk[j] - h0;
                if ((uint32_t)(j * NT + tid) < m && src < (uint32_t)VCAP) outv[j] = s_vals[src];
                BS_FENCE(j);
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t i = j * NT + tid;
            if (i < m) ent[lo + i] = outv[j];
            BS_FENCE(j);
        }
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
        constexpr int IPT = BS_IPT_BIG;
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


}  // namespace cdb
