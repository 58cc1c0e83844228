// scan.h — generic device-wide scan (reduce → scan of tile partials → apply) used for stream compaction,
// group-start propagation (max-scan) and CSR offsets.  Tile = 256 threads × 16 consecutive items.
//
// The input is a functor In: (uint64_t i) -> T, the consumer a functor Out: (i, exclusive, inclusive);
// both are evaluated on the device, so producers/consumers (gathers, key packing, scatters) are fused
// into the scan instead of materialising intermediate arrays in HBM.
#pragma once
#include "common.h"

namespace cdb {

constexpr int SC_NT = 256;
constexpr int SC_IPT = 16;
constexpr int SC_TILE = SC_NT * SC_IPT;

struct OpAdd {
    template <typename T> __device__ __forceinline__ T operator()(const T& a, const T& b) const { return a + b; }
};
struct OpMax {
    template <typename T> __device__ __forceinline__ T operator()(const T& a, const T& b) const { return a > b ? a : b; }
};
struct U2 {
    uint64_t a, b;
    __host__ __device__ U2 operator+(const U2& o) const { return U2{a + o.a, b + o.b}; }
};

// 64-bit-safe lane shuffle for trivially copyable T (built from 32-bit DPP/bpermute moves)
template <typename T>
__device__ __forceinline__ T shfl_up_any(const T& v, int delta) {
    static_assert(sizeof(T) % 4 == 0, "shuffle payload must be a multiple of 4 bytes");
    T out;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&v);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) dst[i] = __shfl_up(src[i], delta);
    return out;
}

// inclusive scan of one value per thread across the 256-thread workgroup (wave-level shuffle scan,
// then one LDS exchange of the 4 wave totals); returns the inclusive value and the workgroup total.
template <typename T, typename Op>
__device__ __forceinline__ T block_scan_incl(T v, Op op, T* s_buf /*[>= 4]*/, T& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T x = shfl_up_any(v, off);
        if (lane >= off) v = op(x, v);
    }
    __syncthreads();  // s_buf may still be read by a previous call
    if (lane == 63) s_buf[wave] = v;
    __syncthreads();
    T pre = s_buf[0];
    total = op(op(s_buf[0], s_buf[1]), op(s_buf[2], s_buf[3]));
    if (wave >= 2) pre = op(pre, s_buf[1]);
    if (wave >= 3) pre = op(pre, s_buf[2]);
    if (wave >= 1) v = op(pre, v);
    return v;
}

// Input functors may offer `load8(base, n, identity, v[SC_IPT])` to fetch a thread's consecutive items at once
// (e.g. one 16-byte load for byte flags); the default falls back to operator().
template <typename In, typename T>
__device__ __forceinline__ auto scan_load8(const In& in, uint64_t base, uint64_t n, const T& identity, T (&v)[SC_IPT], int)
    -> decltype(in.load8(base, n, identity, v), void()) {
    in.load8(base, n, identity, v);
}
template <typename In, typename T>
__device__ __forceinline__ void scan_load8(const In& in, uint64_t base, uint64_t n, const T& identity, T (&v)[SC_IPT], long) {
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k) v[k] = base + k < n ? in(base + k) : identity;
}

template <typename T, typename In, typename Op>
__global__ __launch_bounds__(SC_NT) void scan_reduce_kernel(In in, uint64_t n, Op op, T identity, T* partials) {
    __shared__ T s_buf[8];
    const uint64_t base = (uint64_t)blockIdx.x * SC_TILE + (uint64_t)threadIdx.x * SC_IPT;
    T v[SC_IPT];
    scan_load8(in, base, n, identity, v, 0);
    T acc = identity;
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k) acc = op(acc, v[k]);
    T total;
    block_scan_incl(acc, op, s_buf, total);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// in-place exclusive scan of the tile partials by ONE workgroup of 1024 threads (16 Ki partials per
// sweep); partials[nb] receives the grand total
constexpr int SP_NT = 1024;
template <typename T, typename Op>
__global__ __launch_bounds__(SP_NT) void scan_partials_kernel(T* partials, uint64_t nb, Op op, T identity) {
    __shared__ T s_w[SP_NT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T carry = identity;
    for (uint64_t c = 0; c < nb; c += (uint64_t)SP_NT * SC_IPT) {
        const uint64_t base = c + (uint64_t)threadIdx.x * SC_IPT;
        T v[SC_IPT];
        T acc = identity;
#pragma unroll
        for (int k = 0; k < SC_IPT; ++k) {
            v[k] = base + k < nb ? partials[base + k] : identity;
            acc = op(acc, v[k]);
        }
        T incl = acc;  // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const T x = shfl_up_any(incl, off);
            if (lane >= off) incl = op(x, incl);
        }
        T prev = shfl_up_any(incl, 1);  // inclusive value of the previous lane
        __syncthreads();                // s_w of the previous sweep is consumed
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        T run = carry;                  // everything before this thread
        T total = identity;
#pragma unroll
        for (int w = 0; w < SP_NT / 64; ++w) {
            if (w < wave) run = op(run, s_w[w]);
            total = op(total, s_w[w]);
        }
        if (lane > 0) run = op(run, prev);
#pragma unroll
        for (int k = 0; k < SC_IPT; ++k) {
            if (base + k < nb) partials[base + k] = run;
            run = op(run, v[k]);
        }
        carry = op(carry, total);
    }
    if (threadIdx.x == 0) partials[nb] = carry;
}

template <typename T, typename In, typename Out, typename Op>
__global__ __launch_bounds__(SC_NT) void scan_apply_kernel(In in, uint64_t n, Op op, T identity, const T* partials,
                                                           Out out) {
    __shared__ T s_buf[8];
    const uint64_t base = (uint64_t)blockIdx.x * SC_TILE + (uint64_t)threadIdx.x * SC_IPT;
    T v[SC_IPT];
    scan_load8(in, base, n, identity, v, 0);
    T acc = identity;
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k) acc = op(acc, v[k]);
    T total;
    const T incl = block_scan_incl(acc, op, s_buf, total);
    // exclusive start of this thread = tile prefix ∘ inclusive value of the previous thread
    T prev = shfl_up_any(incl, 1);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) s_buf[4 + (threadIdx.x >> 6)] = incl;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x > 0) prev = s_buf[4 + (threadIdx.x >> 6) - 1];
    T run = partials[blockIdx.x];
    if (threadIdx.x > 0) run = op(run, prev);
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k) {
        const T nxt = op(run, v[k]);
        if (base + k < n) out(base + k, run, nxt);
        run = nxt;
    }
}

// functors that run the scan kernels over the tile partials themselves (second level)
template <typename T> struct PartialsIn {
    const T* p;
    __device__ __forceinline__ T operator()(uint64_t i) const { return p[i]; }
};
template <typename T> struct PartialsOut {  // exclusive prefix in place, grand total behind the last one
    T* p;
    uint64_t nb;
    __device__ __forceinline__ void operator()(uint64_t i, const T& ex, const T& in) const {
        p[i] = ex;
        if (i + 1 == nb) p[nb] = in;
    }
};

// exclusive scan of the nb tile partials in place (partials[nb] = grand total).  One workgroup sweeps up to
// 2^16 partials; beyond that (n > 2^28 items) the partials are scanned like the data: reduce per 4096, scan
// the few second-level partials, apply — three parallel launches instead of a 30 us sweep per 16 Ki partials.
template <typename T, typename Op>
void scan_partials_inplace(hipStream_t s, DevBuf& partials, uint64_t nb, Op op, T identity) {
    T* d_part = partials.as<T>();
    if (nb <= (1ull << 16)) {
        hipLaunchKernelGGL((scan_partials_kernel<T, Op>), dim3(1), dim3(SP_NT), 0, s, d_part, nb, op, identity);
        return;
    }
    const uint64_t nb2 = ceil_div(nb, SC_TILE);
    T* d_part2 = d_part + nb + 1;  // second-level partials live behind the first level (space reserved by callers)
    PartialsIn<T> pin{d_part};
    hipLaunchKernelGGL((scan_reduce_kernel<T, PartialsIn<T>, Op>), dim3((unsigned)nb2), dim3(SC_NT), 0, s, pin, nb, op, identity, d_part2);
    hipLaunchKernelGGL((scan_partials_kernel<T, Op>), dim3(1), dim3(SP_NT), 0, s, d_part2, nb2, op, identity);
    hipLaunchKernelGGL((scan_apply_kernel<T, PartialsIn<T>, PartialsOut<T>, Op>), dim3((unsigned)nb2), dim3(SC_NT), 0, s, pin, nb, op,
                       identity, (const T*)d_part2, PartialsOut<T>{d_part, nb});
}
inline uint64_t scan_partials_slots(uint64_t nb) { return nb + 1 + ceil_div(nb, SC_TILE) + 1; }

// Phase 1: reduce + scan of partials; returns the grand total (synchronises the stream).
template <typename T, typename In, typename Op>
T scan_totals(hipStream_t s, DevBuf& partials, In in, uint64_t n, Op op, T identity) {
    const uint64_t nb = ceil_div(n, SC_TILE);
    partials.ensure(scan_partials_slots(nb) * sizeof(T));
    T* d_part = partials.as<T>();
    if (nb) hipLaunchKernelGGL((scan_reduce_kernel<T, In, Op>), dim3((unsigned)nb), dim3(SC_NT), 0, s, in, n, op, identity, d_part);
    scan_partials_inplace<T, Op>(s, partials, nb, op, identity);
    T total;
    CDB_HIP(hipMemcpyAsync(&total, d_part + nb, sizeof(T), hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
    return total;
}

// Phase 1 when the producer of the data has already left the raw tile sums in `partials` (slots 0 .. nb - 1, same tile
// geometry): only the scan of the partials and the grand total remain.
template <typename T, typename Op>
T scan_totals_from_partials(hipStream_t s, DevBuf& partials, uint64_t n, Op op, T identity) {
    const uint64_t nb = ceil_div(n, SC_TILE);
    scan_partials_inplace<T, Op>(s, partials, nb, op, identity);
    T total;
    CDB_HIP(hipMemcpyAsync(&total, partials.as<T>() + nb, sizeof(T), hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
    return total;
}

// Phase 1 without the host round trip: partials[nb] holds the grand total on the device only.
template <typename T, typename In, typename Op>
void scan_totals_device(hipStream_t s, DevBuf& partials, In in, uint64_t n, Op op, T identity) {
    const uint64_t nb = ceil_div(n, SC_TILE);
    partials.ensure(scan_partials_slots(nb) * sizeof(T));
    T* d_part = partials.as<T>();
    if (nb) hipLaunchKernelGGL((scan_reduce_kernel<T, In, Op>), dim3((unsigned)nb), dim3(SC_NT), 0, s, in, n, op, identity, d_part);
    scan_partials_inplace<T, Op>(s, partials, nb, op, identity);
}

// Phase 2: apply (uses the partials left by scan_totals for the same `in`, n, op).
template <typename T, typename In, typename Out, typename Op>
void scan_apply(hipStream_t s, DevBuf& partials, In in, uint64_t n, Op op, T identity, Out out) {
    const uint64_t nb = ceil_div(n, SC_TILE);
    if (!nb) return;
    hipLaunchKernelGGL((scan_apply_kernel<T, In, Out, Op>), dim3((unsigned)nb), dim3(SC_NT), 0, s, in, n, op, identity,
                       (const T*)partials.as<T>(), out);
}

}  // namespace cdb
