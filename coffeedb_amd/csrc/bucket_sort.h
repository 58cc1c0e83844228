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
    int list_mode;      // workgroup kernel: wfirst holds (first, one past last) bucket pairs instead of window starts
    int ablate;         // timing experiments only (CDB_BS_ABLATE; WRONG results): 1 = no wave sorts, 2 = no flags / kept keys,
                        // 4 = no entry gather, 8 = no counting / scatter
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

// ---- sorting inside a wavefront: a bitonic network over 64 * R values held R per lane (element r * 64 + lane) ------
// Exchanges between lanes are shuffles, exchanges across 64-element blocks are register pairs: no LDS traffic, no
// barriers.  Ascending order; callers pad with all-ones.
// value of lane ^ Q without the LDS crossbar: DPP controls inside a row of 16 lanes (quad permutations, row rotation,
// half-row mirror), v_permlane16/32_swap across rows — one or two full-rate vector instructions instead of a
// ds_bpermute (gfx950; tools/experiments/dpp/xor_partner_test.hip checks every pattern)
template <int Q>
__device__ __forceinline__ uint32_t bs_xor_partner(uint32_t x, int lane) {
    if constexpr (Q == 1) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    } else if constexpr (Q == 2) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    } else if constexpr (Q == 4) {
        const int t = __builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);     // row_half_mirror: lane ^ 7
        return (uint32_t)__builtin_amdgcn_mov_dpp(t, 0x1B, 0xF, 0xF, true);        // quad_perm [3,2,1,0]: lane ^ 3
    } else if constexpr (Q == 8) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, true);  // row_ror:8
    } else if constexpr (Q == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else {
        static_assert(Q == 32, "lane distance inside a wavefront");
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}
__device__ __forceinline__ uint32_t bs_xor_partner_q(uint32_t x, int q, int lane) {  // (q is a constant after unrolling)
    switch (q) {
        case 1: return bs_xor_partner<1>(x, lane);
        case 2: return bs_xor_partner<2>(x, lane);
        case 4: return bs_xor_partner<4>(x, lane);
        case 8: return bs_xor_partner<8>(x, lane);
        case 16: return bs_xor_partner<16>(x, lane);
        default: return bs_xor_partner<32>(x, lane);
    }
}

template <int R>
__device__ __forceinline__ void bs_wave_sort(uint32_t (&v)[R], int lane) {
#pragma unroll
    for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
        for (int q = k >> 1; q > 0; q >>= 1) {
            if (q >= 64) {
                const int rq = q >> 6;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if ((r & rq) == 0) {
                        const int r2 = r | rq;
                        const bool up = ((r * 64) & k) == 0;  // (k >= 128 here: bit k of the element index is a bit of r)
                        const uint32_t a = v[r], c = v[r2];
                        const uint32_t mn = a < c ? a : c, mx = a < c ? c : a;
                        v[r] = up ? mn : mx;
                        v[r2] = up ? mx : mn;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t o = bs_xor_partner_q(v[r], q, lane);
                    const bool up = (((r * 64) | lane) & k) == 0;
                    const bool lower = (lane & q) == 0;
                    const uint32_t mn = v[r] < o ? v[r] : o, mx = v[r] < o ? o : v[r];
                    v[r] = (lower == up) ? mn : mx;
                }
            }
        }
    }
}

// one wavefront sorts x[0, len) in LDS for any len (rare path: sub-buckets of more than 256 records): bitonic merge
// sort in its "flip" form, every comparator ascending, partners beyond len skipped (= padding with +infinity)
__device__ __forceinline__ void bs_wave_sort_lds(uint32_t* x, uint32_t len, int lane) {
    uint32_t np2 = 2;
    while (np2 < len) np2 <<= 1;
    for (uint32_t k = 2; k <= np2; k <<= 1) {
        for (uint32_t t = lane; t < np2 / 2; t += 64) {  // flip: i = block start + u, partner = block end - u
            const uint32_t blk = t / (k / 2), u = t % (k / 2);
            const uint32_t i = blk * k + u, p2 = blk * k + (k - 1 - u);
            if (p2 < len) {
                const uint32_t a = x[i], c = x[p2];
                if (a > c) { x[i] = c; x[p2] = a; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t q = k >> 2; q > 0; q >>= 1) {
            for (uint32_t t = lane; t < np2 / 2; t += 64) {
                const uint32_t i = ((t / q) * 2 * q) + (t % q), p2 = i + q;
                if (p2 < len) {
                    const uint32_t a = x[i], c = x[p2];
                    if (a > c) { x[i] = c; x[p2] = a; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// for every nominal window of `step` records: the first bucket that starts inside it (one parallel search per
// window instead of a chain of dependent loads at the head of every workgroup)
template <typename P>
__global__ __launch_bounds__(256) void bs_windows_kernel(const P* __restrict__ bstart, uint64_t nb, uint64_t n, uint64_t step,
                                                         uint64_t nwin, uint32_t* __restrict__ wfirst /*[nwin + 1]*/) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t > nwin) return;
    const uint64_t pos = t * step < n ? t * step : n;
    wfirst[t] = (uint32_t)bs_lower_bound(bstart, nb, pos);
}

// V = entry type, W = auxiliary digit type (u8 / u16), P = position type of the bucket table, KW = type of the kept
// low digits (u8 / u16).  NT x IPT = capacity (<= 2^15 records: input indices are 15-bit).
//
// One round = consecutive buckets [b, b1) of together m <= capacity records whose local key
//     lk = (bucket - b) << rbits | r            (lbits <= 25 bits)
// is sorted in two steps:
//   A. counting sort in LDS on the top hb = min(8, lbits) bits of lk — ONE pass, and it need not be stable, because
//      what it stores per record is the word  (low bits of lk) << 15 | (index of the record in the round's input):
//   B. every sub-bucket (one value of the top bits: ~ m / 256 records) is sorted by ONE wavefront in registers
//      (bs_wave_sort), as plain integers — low key bits first, input index as tie-break, i.e. stably.  The wavefront
//      then knows the final neighbours of its records and writes their group flags and kept keys itself.
// The entries never pass through registers before they are final: the input window is read once, coalesced, into
// the staging buffer after the keys are done, and leaves through an LDS gather by the sorted indices.
template <typename V, typename W, typename P, typename KW, int NT, int IPT>
__global__ __launch_bounds__(NT) void bs_local_sort_kernel(uint32_t* __restrict__ k32, V* __restrict__ ent, W* __restrict__ aux,
                                                           const P* __restrict__ bstart, const uint32_t* __restrict__ wfirst,
                                                           BucketSortParams pr, uint8_t* __restrict__ flags,
                                                           KW* __restrict__ keylow_out,
                                                           unsigned long long* __restrict__ oversize /*[0] count, [1] largest*/) {
    constexpr int CAP = NT * IPT;
    static_assert(CAP <= 32768, "input indices are 15-bit");
    constexpr int NW = NT / 64;
    constexpr int VCAP = sizeof(V) > 4 ? CAP / 2 : CAP;  // entries staged per sweep of the final gather
    constexpr int LBITS_MAX = 25;                        // 8 bits of counting sort + 17 bits beside the 15-bit index
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[CAP];  // packed records, then entries
    __shared__ uint32_t s_cnt[256], s_start[257];
    __shared__ uint32_t s_wsum[4];
    __shared__ unsigned long long s_round[3];  // {unused, one past the last bucket, records} of the round
    V* s_vals = reinterpret_cast<V*>(s_keys);
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t rmask = (1u << pr.rbits) - 1u;  // (rbits <= 24: bucket_sort_plan)
    const uint32_t lmask = (1u << pr.lead_bits) - 1u;
    const uint64_t max_rel = 1ull << (LBITS_MAX - pr.rbits);  // buckets one round can tell apart

    // buckets [b, b_end): those that start in this workgroup's window, or (list mode) one listed range
    uint64_t b = pr.list_mode ? wfirst[2 * blockIdx.x] : wfirst[blockIdx.x];
    const uint64_t b_end = pr.list_mode ? wfirst[2 * blockIdx.x + 1] : wfirst[blockIdx.x + 1];
    while (b < b_end) {
        // ---- the round: as many buckets from b on as fit (64-ary search by the first wavefront)
        __syncthreads();
        if (wave == 0) {
            const uint64_t lo0 = (uint64_t)bstart[b];
            uint64_t l = b, h = b_end;  // invariant: bstart[l] - lo0 <= CAP; answer = largest such index in [l, h]
            if (h - b > max_rel) h = b + max_rel;
            while (l < h) {
                const uint64_t span = h - l;
                const uint64_t c = l + (span * (uint64_t)(lane + 1) + 63) / 64;  // l < c <= h, ascending with the lane
                const bool ok = (uint64_t)bstart[c] - lo0 <= (uint64_t)CAP;
                const uint64_t okm = __ballot(ok);
                const int nok = __popcll(okm);  // (monotone: the first nok lanes)
                const uint64_t lnew = nok ? __shfl(c, nok - 1) : l;
                const uint64_t hnew = nok < 64 ? __shfl(c, nok) - 1 : h;
                l = lnew;
                h = hnew;
            }
            if (lane == 0) {
                s_round[1] = l;
                s_round[2] = (uint64_t)bstart[l] - lo0;
                if (l == b) {  // bucket b alone exceeds the capacity: report it (the host falls back)
                    atomicAdd(&oversize[0], 1ull);
                    atomicMax(&oversize[1], (unsigned long long)((uint64_t)bstart[b + 1] - lo0));
                }
            }
        }
        if (tid < 256) s_cnt[tid] = 0;
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
        // top digit: as many bits as give sub-buckets of ~48 records (one wavefront sorts up to 64 in a single
        // register), at least lbits - 17 (the rest must fit beside the index), at most 8
        int hb = 32 - __clz((m + 47) / 48);
        if (hb < lbits - 17) hb = lbits - 17;
        if (hb > 8) hb = 8;
        if (hb > lbits) hb = lbits;
        const int lowbits = lbits - hb;
        const uint32_t lowmask = (1u << lowbits) - 1u;
        // ---- A. load + count the top digit
        uint32_t lk[IPT];
        {
            const uint32_t* kp = k32 + lo;
            const W* ap = aux + lo;
            const uint32_t b32 = (uint32_t)b;
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const uint32_t li = j * NT + tid;
                lk[j] = 0;
                if (li < m) {
                    const uint32_t k = kp[li];
                    // bucket of the record relative to the round's first one (bucket numbers fit 32 bits: nb <= 2^24)
                    const uint32_t brel = (((k >> pr.rbits) << pr.lead_bits) | ((uint32_t)ap[li] & lmask)) - b32;
                    lk[j] = (brel << pr.rbits) | (k & rmask);
                    if (!(pr.ablate & 8)) atomicAdd(&s_cnt[lk[j] >> lowbits], 1u);
                }
                BS_FENCE(j);
            }
        }
        __syncthreads();
        // exclusive scan of the 256 counts
        {
            uint32_t cnt = 0, incl = 0;
            if (tid < 256) {
                cnt = s_cnt[tid];
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
                s_start[tid] = wpre + incl - cnt;
                s_cnt[tid] = wpre + incl - cnt;  // running fill position
                if (tid == 255) s_start[256] = wpre + incl;
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t li = j * NT + tid;
            if (li < m) {
                uint32_t kx = lk[j];
                asm volatile("" : "+v"(kx));  // (no sharing of the digit with the counting loop across the barriers)
                const uint32_t at = (pr.ablate & 8) ? li : atomicAdd(&s_cnt[kx >> lowbits], 1u);
                s_keys[at] = ((kx & lowmask) << 15) | li;
            }
            BS_FENCE(j);
        }
        __syncthreads();
        // ---- B. every sub-bucket sorted by one wavefront; flags and kept keys of its records
        for (uint32_t d = wave; d < (1u << hb); d += NW) {
            const uint32_t st = s_start[d], len = s_start[d + 1] - st;
            if (len == 0 || (pr.ablate & 8)) continue;
            uint32_t* x = s_keys + st;
            // emits the records of sorted slots [st + base, st + base + 64): v = packed word of this lane's slot,
            // pv / nv = the packed words before / behind it (all-ones = none inside the sub-bucket)
            auto emit = [&](uint32_t i, uint32_t v, uint32_t pv, uint32_t nv) {
                if (i >= len || (pr.ablate & 2)) return;
                const uint32_t low = v >> 15;
                const bool head = i == 0 || (pv >> 15) != low;          // (another sub-bucket = another key)
                const bool tail = i + 1 == len || (nv >> 15) != low;
                const uint32_t lkx = (d << lowbits) | low;
                const uint64_t K = (b + (uint64_t)(lkx >> pr.rbits)) * pr.w + (uint64_t)(lkx & rmask);
                const bool exhausted = pr.kmagic ? (K - __umul64hi(K, pr.kmagic) * pr.kbase) == 0 : (K & (uint64_t)(pr.kbase - 1u)) == 0;
                const uint64_t g = lo + st + i;
                flags[g] = (uint8_t)((head ? 1 : 0) | ((!(head && tail) && !exhausted) ? 2 : 0));
                if (keylow_out) {  // kept search keys in the layout of the plain split sort
                    k32[g] = (uint32_t)(K >> pr.out_low_bits);
                    keylow_out[g] = (KW)(K & ((1ull << pr.out_low_bits) - 1ull));
                }
            };
            if (len <= 64) {
                uint32_t v[1] = {(uint32_t)lane < len ? x[lane] : 0xFFFFFFFFu};
                if (!(pr.ablate & 1)) bs_wave_sort<1>(v, lane);
                if ((uint32_t)lane < len) x[lane] = v[0];
                const uint32_t pv = __shfl_up(v[0], 1), nv = __shfl_down(v[0], 1);
                emit(lane, v[0], pv, nv);
            } else if (len <= 128) {
                uint32_t v[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) v[r] = r * 64 + lane < len ? x[r * 64 + lane] : 0xFFFFFFFFu;
                if (!(pr.ablate & 1)) bs_wave_sort<2>(v, lane);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (r * 64 + lane < len) x[r * 64 + lane] = v[r];
                    uint32_t pv = __shfl_up(v[r], 1), nv = __shfl_down(v[r], 1);
                    if (r > 0) { const uint32_t e = __shfl(v[r - 1], 63); if (lane == 0) pv = e; }
                    if (r < 1) { const uint32_t e = __shfl(v[r + 1], 0); if (lane == 63) nv = e; }
                    emit(r * 64 + lane, v[r], pv, nv);
                }
            } else if (len <= 256) {
                uint32_t v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = r * 64 + lane < len ? x[r * 64 + lane] : 0xFFFFFFFFu;
                if (!(pr.ablate & 1)) bs_wave_sort<4>(v, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r * 64 + lane < len) x[r * 64 + lane] = v[r];
                    uint32_t pv = __shfl_up(v[r], 1), nv = __shfl_down(v[r], 1);
                    if (r > 0) { const uint32_t e = __shfl(v[r - 1], 63); if (lane == 0) pv = e; }
                    if (r < 3) { const uint32_t e = __shfl(v[r + 1], 0); if (lane == 63) nv = e; }
                    emit(r * 64 + lane, v[r], pv, nv);
                }
            } else {
                bs_wave_sort_lds(x, len, lane);
                for (uint32_t i = lane; i < len; i += 64)
                    emit(i, x[i], i ? x[i - 1] : 0xFFFFFFFFu, i + 1 < len ? x[i + 1] : 0xFFFFFFFFu);
            }
        }
        __syncthreads();
        // ---- the entries: sorted position -> input index, then the input window is staged (coalesced) and gathered
        // in LDS.  8-byte entries are staged in two halves of the window.
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t i = j * NT + tid;
            lk[j] = i < m ? (s_keys[i] & 0x7FFFu) : 0u;  // (the keys are done: their registers carry the indices)
            BS_FENCE(j);
        }
        __syncthreads();
        V outv[IPT];
        for (uint32_t half = 0; half < (uint32_t)(CAP / VCAP); ++half) {
            const uint32_t h0 = half * VCAP;
            if (h0 >= m || (pr.ablate & 4)) break;
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

// ---- small buckets: one WAVEFRONT per round, straight from global memory --------------------------------------------
// After three global passes a bucket holds a few dozen records (C1: 64 on average).  A wavefront takes consecutive
// buckets of together <= 256 records, loads them into registers (4 per lane), sorts the packed words
//     ((bucket - first bucket) << rbits | r) << 8 | index in the round
// with the register network above — low key bits first, input index as tie-break, i.e. stably —, pulls the entries
// to their sorted places with lane shuffles, and writes entries, group flags and kept keys.  No LDS, no barriers,
// many wavefronts per SIMD to hide the loads: the kernel streams.  Buckets of more than 256 records are listed for
// the workgroup kernel above.
template <typename V>
__device__ __forceinline__ V bs_shfl_any(V v, int src) {
    if constexpr (sizeof(V) == 4) {
        return (V)__shfl((uint32_t)v, src);
    } else {
        const uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)((uint64_t)v >> 32), src);
        return (V)(((uint64_t)hi << 32) | lo);
    }
}

template <int R, typename V, typename W, typename KW>
__device__ __forceinline__ void bs_wave_round(uint32_t* __restrict__ k32, V* __restrict__ ent, W* __restrict__ aux, uint64_t lo, uint32_t m,
                                              uint32_t b32, const BucketSortParams& pr, uint8_t* __restrict__ flags,
                                              KW* __restrict__ keylow_out, int lane) {
    const uint32_t rmask = (1u << pr.rbits) - 1u;
    const uint32_t lmask = (1u << pr.lead_bits) - 1u;
    uint32_t v[R];
    V e[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t li = r * 64 + lane;
        v[r] = 0xFFFFFFFFu;
        e[r] = V(0);
        if (li < m) {
            const uint32_t k = k32[lo + li];
            const uint32_t brel = (((k >> pr.rbits) << pr.lead_bits) | ((uint32_t)aux[lo + li] & lmask)) - b32;
            v[r] = (((brel << pr.rbits) | (k & rmask)) << 8) | li;
            e[r] = ent[lo + li];
        }
    }
    if (!(pr.ablate & 1)) bs_wave_sort<R>(v, lane);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = r * 64 + lane;
        // the entry of the record that sorted to slot i sits in register (idx >> 6) of lane (idx & 63)
        const uint32_t idx = v[r] & 255u;
        V o = bs_shfl_any<V>(e[0], (int)(idx & 63u));
#pragma unroll
        for (int q = 1; q < R; ++q) {
            const V t = bs_shfl_any<V>(e[q], (int)(idx & 63u));
            o = (idx >> 6) == (uint32_t)q ? t : o;
        }
        // neighbours in sorted order: whole-wave shifts by one lane (DPP wave_shr:1 / wave_shl:1), the lanes at the ends
        // take the edge values of the neighbouring register
        uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp((int)v[r], (int)v[r], 0x138, 0xF, 0xF, false);
        uint32_t nv = (uint32_t)__builtin_amdgcn_update_dpp((int)v[r], (int)v[r], 0x130, 0xF, 0xF, false);
        if (r > 0) { const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v[r - 1], 63); if (lane == 0) pv = x; }
        if (r + 1 < R) { const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v[r + 1], 0); if (lane == 63) nv = x; }
        if (i < m) {
            const uint32_t lkx = v[r] >> 8;
            const bool head = i == 0 || (pv >> 8) != lkx;
            const bool tail = i + 1 == m || (nv >> 8) != lkx;
            const uint64_t K = ((uint64_t)b32 + (uint64_t)(lkx >> pr.rbits)) * pr.w + (uint64_t)(lkx & rmask);
            const bool exhausted = pr.kmagic ? (K - __umul64hi(K, pr.kmagic) * pr.kbase) == 0 : (K & (uint64_t)(pr.kbase - 1u)) == 0;
            const uint64_t g = lo + i;
            ent[g] = o;
            flags[g] = (uint8_t)((head ? 1 : 0) | ((!(head && tail) && !exhausted) ? 2 : 0));
            if (keylow_out) {
                k32[g] = (uint32_t)(K >> pr.out_low_bits);
                keylow_out[g] = (KW)(K & ((1ull << pr.out_low_bits) - 1ull));
            }
        }
    }
}

template <typename V, typename W, typename P, typename KW>
__global__ __launch_bounds__(256) void bs_wave_sort_kernel(uint32_t* __restrict__ k32, V* __restrict__ ent, W* __restrict__ aux,
                                                          const P* __restrict__ bstart, BucketSortParams pr, uint32_t per_wave,
                                                          uint8_t* __restrict__ flags, KW* __restrict__ keylow_out,
                                                          uint32_t* __restrict__ big_list, uint32_t big_cap,
                                                          unsigned long long* __restrict__ big_count) {
    const int lane = threadIdx.x & 63;
    const uint64_t wv = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint64_t b = wv * per_wave;
    if (b >= pr.nb) return;
    const uint64_t b_end = b + per_wave < pr.nb ? b + per_wave : pr.nb;  // this wavefront's buckets
    // buckets one round can tell apart: the packed word holds (bucket - first) << rbits | r above the 8-bit index
    const uint64_t max_rel = 1ull << (24 - pr.rbits);
    while (b < b_end) {
        const uint64_t lo = (uint64_t)bstart[b];
        const uint64_t b_lim = b_end - b > max_rel ? b + max_rel : b_end;
        const uint64_t c = b + 1 + (uint64_t)lane;
        const uint64_t endc = c <= b_lim ? (uint64_t)bstart[c] : ~0ull;
        const bool ok = c <= b_lim && endc - lo <= 256;
        const int nok = __popcll(__ballot(ok));  // (bucket starts ascend: the lanes that fit are the first nok)
        if (nok == 0) {  // bucket b alone has more than 256 records: the workgroup kernel takes it
            if (lane == 0) {
                const unsigned long long at = atomicAdd(big_count, 1ull);
                if (at < big_cap) {
                    big_list[2 * at] = (uint32_t)b;
                    big_list[2 * at + 1] = (uint32_t)b + 1;
                }
            }
            b += 1;
            continue;
        }
        const uint32_t m = (uint32_t)(bs_shfl_any<uint64_t>(endc, nok - 1) - lo);
        if (m > 128) bs_wave_round<4, V, W, KW>(k32, ent, aux, lo, m, (uint32_t)b, pr, flags, keylow_out, lane);
        else if (m > 64) bs_wave_round<2, V, W, KW>(k32, ent, aux, lo, m, (uint32_t)b, pr, flags, keylow_out, lane);
        else if (m > 0) bs_wave_round<1, V, W, KW>(k32, ent, aux, lo, m, (uint32_t)b, pr, flags, keylow_out, lane);
        b += (uint64_t)nok;
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
    BucketSortParams pr{n, plan.nb, plan.w, plan.rbits, 8 * plan.lead, kbase, kmagic, out_low_bits, 0, 0, 0};
    if (const char* e = std::getenv("CDB_BS_ABLATE")) pr.ablate = std::atoi(e);
    const bool big = plan.cap > 4096;
    if (plan.rbits <= 22 && (double)n / (double)plan.nb <= 160.0) {
        // small buckets: one wavefront per round of <= 256 records (bs_wave_sort_kernel); what does not fit goes to
        // the workgroup kernel through a list
        // buckets per wavefront: ~16 rounds of <= 256 records each (short-lived wavefronts start and finish in step
        // with each other and leave the memory pipes idle while they all sort; CDB_BS_PER_WAVE overrides)
        uint32_t per_wave = 1;
        while (per_wave < 4096 && (double)n / (double)plan.nb * (2.0 * per_wave) <= 16.0 * 256.0) per_wave *= 2;
        if (const char* e = std::getenv("CDB_BS_PER_WAVE")) per_wave = (uint32_t)std::max(1, std::atoi(e));
        const uint64_t waves = ceil_div(plan.nb, (uint64_t)per_wave);
        constexpr uint32_t BIG_CAP = 1u << 20;
        DevBuf d_list;
        d_list.alloc((size_t)BIG_CAP * 2 * sizeof(uint32_t));
        int t = prof.begin(s);
        hipLaunchKernelGGL((bs_wave_sort_kernel<V, W, P, KW>), dim3((unsigned)ceil_div(waves, 4)), dim3(256), 0, s, k32, ent, aux,
                           (const P*)bstart, pr, per_wave, flags, keylow_out, d_list.as<uint32_t>(), BIG_CAP,
                           d_over.as<unsigned long long>());
        prof.end(t, "sa_bucket_wave_sort", n * (2 * (4 + sizeof(V)) + sizeof(W) + 1 + (keylow_out ? sizeof(KW) : 0)), s);
        uint64_t nbig = 0;
        CDB_HIP(hipMemcpyAsync(&nbig, d_over.p, sizeof(nbig), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipGetLastError());
        CDB_HIP(hipStreamSynchronize(s));
        if (nbig > BIG_CAP) {
            if (largest_bucket) *largest_bucket = 0;
            return false;  // (too many large buckets for the list: a skewed corpus — the caller falls back)
        }
        if (nbig == 0) {
            if (largest_bucket) *largest_bucket = 0;
            return true;
        }
        CDB_HIP(hipMemsetAsync(d_over.p, 0, 2 * sizeof(uint64_t), s));
        pr.list_mode = 1;
        int t2 = prof.begin(s);
        if (big)
            hipLaunchKernelGGL((bs_local_sort_kernel<V, W, P, KW, 1024, BS_IPT_BIG>), dim3((unsigned)nbig), dim3(1024), 0, s, k32, ent, aux,
                               (const P*)bstart, (const uint32_t*)d_list.as<uint32_t>(), pr, flags, keylow_out,
                               d_over.as<unsigned long long>());
        else
            hipLaunchKernelGGL((bs_local_sort_kernel<V, W, P, KW, 256, 16>), dim3((unsigned)nbig), dim3(256), 0, s, k32, ent, aux,
                               (const P*)bstart, (const uint32_t*)d_list.as<uint32_t>(), pr, flags, keylow_out,
                               d_over.as<unsigned long long>());
        prof.end(t2, "sa_bucket_sort", 0, s);
        uint64_t over2[2] = {0, 0};
        CDB_HIP(hipMemcpyAsync(over2, d_over.p, sizeof(over2), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipGetLastError());
        CDB_HIP(hipStreamSynchronize(s));
        if (largest_bucket) *largest_bucket = over2[1];
        return over2[0] == 0;
    }
    pr.step = big ? (uint64_t)1024 * 16 : 2048;
    const uint64_t nwin = ceil_div(n, pr.step);
    DevBuf d_win;
    d_win.alloc((nwin + 1) * sizeof(uint32_t));
    hipLaunchKernelGGL((bs_windows_kernel<P>), dim3((unsigned)ceil_div(nwin + 1, 256)), dim3(256), 0, s, (const P*)bstart, plan.nb, n,
                       pr.step, nwin, d_win.as<uint32_t>());
    int t = prof.begin(s);
    if (big) {
        constexpr int IPT = BS_IPT_BIG;
        hipLaunchKernelGGL((bs_local_sort_kernel<V, W, P, KW, 1024, IPT>), dim3((unsigned)nwin), dim3(1024), 0, s, k32, ent, aux,
                           (const P*)bstart, (const uint32_t*)d_win.as<uint32_t>(), pr, flags, keylow_out,
                           d_over.as<unsigned long long>());
    } else {
        hipLaunchKernelGGL((bs_local_sort_kernel<V, W, P, KW, 256, 16>), dim3((unsigned)nwin), dim3(256), 0, s, k32, ent, aux,
                           (const P*)bstart, (const uint32_t*)d_win.as<uint32_t>(), pr, flags, keylow_out,
                           d_over.as<unsigned long long>());
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
