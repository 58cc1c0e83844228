// sa_build.hip — generalised suffix-array construction on gfx950.
//
// Replaces string_index::build()'s sort (reference /root/reference/src/index.cpp:209-231: fill
// sa[k] = (off << bits) | doc, then multithreaded MSD radix with std::sort leaves,
// index.cpp:75-128).  Result contract: the same multiset of entries, ordered by suffix text in
// unsigned byte order with "end of document" smallest (index.h:61-73), and equal suffixes (identical
// tails of different documents) ordered by document index — the canonical form of the reference's
// arbitrary tie order.
//
// Algorithm (GPU-first; none of the reference's task queue survives):
//   1. alphabet        : byte histogram -> dense order-preserving symbol codes, `symbits` per symbol
//   2. key width       : `nsym` symbols per key, chosen from a sorted sample of the text so that few suffixes
//                        stay unresolved (code 0 = end of document: shorter suffixes first)
//   3. initial sort    : stable LSD radix sort of (key, entry) records, entry = (off << bits) | doc
//                        (radix_sort.h).  The key is the symbol codes as a number in base alphabet+1 when that
//                        saves a pass over bit-aligned symbols (digit histograms: sa_keyhist_kernel, rolling
//                        keys over the text), else bit-aligned (histograms from the byte counts).  The first
//                        pass GENERATES its records from the text (TextGen) and drops the digit(s) it sorts on
//                        into a byte (two bytes) per suffix, so later passes move (u32 key, entry, u8/u16)
//                        when the rest of the key fits 32 bits; sa_keygen_kernel + (u64 key, entry) otherwise.
//                        Corpora of 2^32 bytes and more are first partitioned by their first symbol (entries
//                        only) and sorted bucket by bucket with keys gathered per bucket (memory).
//   4. refinement      : only groups of still-equal keys are touched again.  While few suffixes are
//                        unresolved they are re-keyed straight from the text (next symbols);
//                        otherwise an inverse array (rank per text position) is built and classic
//                        prefix doubling runs on the unresolved groups (sort key = group id ∘ rank of the
//                        suffix h symbols further on).  A group whose members end inside the compared
//                        prefix is final: its members are identical suffixes, and because every sort is
//                        stable and the initial order is text order they already ascend by document.
//   5. reference order : for text with bytes >= 0x80 the reference's signed-bucket order is reproduced by
//                        block rotations (apply_reference_order) unless reference_compat is switched off.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "index_impl.h"
#include "records_sweep.h"
#include "vl_code.h"
#include "scan.h"

// Timing-only ablations of the record gather (WRONG records): compile-time only (-DRS_GATHER_ABL=<bits>, a library of its own loaded
// through CDB_LIB_PATH by tools/big_ablate.sh), never a run-time switch of the product build.
#ifndef RS_GATHER_ABL
#define RS_GATHER_ABL 0
#endif

namespace cdb {
namespace {

constexpr int KG_TILE = 2048;  // suffixes per workgroup in key generation
constexpr int KG_LOOK = 64;    // look-ahead bytes staged behind the tile

// ---------------------------------------------------------------------------------------------
// 1. alphabet: how often every byte value occurs
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sa_bytecount_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                           unsigned long long* __restrict__ counts) {
    __shared__ uint32_t s[256];
    s[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t words = n / 16;
    const uint4* t4 = reinterpret_cast<const uint4*>(text);
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += stride) {
        const uint4 v = t4[w];
        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            atomicAdd(&s[x[q] & 0xFF], 1u);
            atomicAdd(&s[(x[q] >> 8) & 0xFF], 1u);
            atomicAdd(&s[(x[q] >> 16) & 0xFF], 1u);
            atomicAdd(&s[x[q] >> 24], 1u);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 15)) atomicAdd(&s[text[words * 16 + threadIdx.x]], 1u);
    __syncthreads();
    if (s[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s[threadIdx.x]);
}

// The same per tile of RS_GEN8_TILE positions (bucket-wise builds with look-back-free generated passes, radix_sort.h:
// TextGen::tile_base): counts[tile][byte] — every position starts a suffix whose bucket is its first byte's, so these rows, with
// their columns mapped byte -> bucket slot, are the per-tile digit counts of the records / partition pass, and their column sums
// are the byte histogram.  Wave-autonomous like sa_tile_paircount_kernel below.
constexpr uint32_t TBC_TILES_PER_WAVE = 8;
// PAIR (round 5): the same sweep also counts, per byte value, the positions whose NEXT byte is >= 0x80 (sa_pairclass_kernel's
// pair[b][1], document ends ignored as there): the counter index carries that bit (byte | next_high << 8, 512 counters per copy
// instead of 256 — still one LDS atomic per position), a tile's row is the sum of both halves, the upper halves add up in four
// registers per lane and reach `pair_hi[b]` once per wavefront.  Saves the reference-order build its second sweep over the text.
template <bool PAIR>
__global__ __launch_bounds__(256) void sa_tile_bytecount_kernel(const uint8_t* __restrict__ text, uint64_t n, uint32_t tiles, uint32_t* __restrict__ counts,
                                                                unsigned long long* __restrict__ pair_hi /*[256], PAIR only*/) {
    // (four copies of a wave's counters, by lane — and 8 banks apart: text is skewed, a copy stride of 256 words would leave the
    //  four counters of a frequent byte on ONE bank, where their atomics serialise just as on one address)
    constexpr int NB = PAIR ? 512 : 256;
    constexpr int CS = NB + 8;
    __shared__ uint32_t s_cnt[4][4 * CS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t* cnt = &s_cnt[wave][(lane & 3) * CS];
    uint32_t* call = &s_cnt[wave][0];
    uint32_t pc[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < TBC_TILES_PER_WAVE; ++k) {
        const uint32_t tile = (blockIdx.x * 4u + (uint32_t)wave) * TBC_TILES_PER_WAVE + k;
        if (tile >= tiles) break;  // (uniform per wavefront)
#pragma unroll
        for (int q = 0; q < (4 * CS + 63) / 64; ++q)
            if (q * 64 + lane < 4 * CS) call[q * 64 + lane] = 0;
        const uint64_t b0 = (uint64_t)tile * RS_GEN8_TILE;
        constexpr int VEC = RS_GEN8_TILE / 16 / 64;
#pragma unroll 2
        for (int r = 0; r < VEC; ++r) {
            const uint64_t p = b0 + ((uint64_t)r * 64 + lane) * 16;
            uint32_t x[4] = {0, 0, 0, 0};
            const bool full = p + 16 <= n;
            if (full) {
                const uint4 v = *reinterpret_cast<const uint4*>(text + p);
                x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
            }
            uint32_t follow = 0;  // PAIR: the byte behind this lane's sixteen (the neighbour lane's first; lane 63 reads it)
            if constexpr (PAIR) {
                follow = __shfl_down(x[0] & 0xFFu, 1);
                if (lane == 63 || !full) follow = p + 16 < n ? (uint32_t)text[p + 16] : 0u;
                // (a neighbour lane without a full vector handed over 0: only the last vectors of the text, whose own lanes take the
                //  byte-wise path below — the lane in front of them reads its follower itself)
                const uint64_t pn = p + 16;
                if (full && lane != 63 && !(pn + 16 <= n)) follow = pn < n ? (uint32_t)text[pn] : 0u;
            }
            if (p >= n) continue;
            if (full) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if constexpr (PAIR) {
                        const uint32_t nx = q < 3 ? x[q + 1] & 0xFFu : follow;  // the byte behind this dword
                        atomicAdd(&cnt[(x[q] & 0xFF) | ((x[q] >> 7) & 0x100u)], 1u);
                        atomicAdd(&cnt[((x[q] >> 8) & 0xFF) | ((x[q] >> 15) & 0x100u)], 1u);
                        atomicAdd(&cnt[((x[q] >> 16) & 0xFF) | ((x[q] >> 23) & 0x100u)], 1u);
                        atomicAdd(&cnt[(x[q] >> 24) | ((nx << 1) & 0x100u)], 1u);
                    } else {
                        atomicAdd(&cnt[x[q] & 0xFF], 1u);
                        atomicAdd(&cnt[(x[q] >> 8) & 0xFF], 1u);
                        atomicAdd(&cnt[(x[q] >> 16) & 0xFF], 1u);
                        atomicAdd(&cnt[x[q] >> 24], 1u);
                    }
                }
            } else {
                for (int q = 0; q < 16; ++q)
                    if (p + q < n) {
                        uint32_t idx = text[p + q];
                        if constexpr (PAIR) idx |= (p + q + 1 < n ? ((uint32_t)text[p + q + 1] << 1) & 0x100u : 0u);
                        atomicAdd(&cnt[idx], 1u);
                    }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int dgt = q * 64 + lane;
            uint32_t c = call[dgt] + call[CS + dgt] + call[2 * CS + dgt] + call[3 * CS + dgt];
            if constexpr (PAIR) {
                const uint32_t hi = call[256 + dgt] + call[CS + 256 + dgt] + call[2 * CS + 256 + dgt] + call[3 * CS + 256 + dgt];
                c += hi;
                pc[q] += hi;  // (a wavefront counts at most 8 x 8192 positions: no overflow)
            }
            counts[(size_t)tile * 256 + dgt] = c;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (PAIR) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (pc[q]) atomicAdd(&pair_hi[q * 64 + lane], (unsigned long long)pc[q]);
    }
}
// any byte >= 0x80 among the first `len` bytes?  (decides whether the byte count is worth its PAIR form before anything is known
// about the text; a text whose high bytes only start later pays sa_pairclass_kernel's separate sweep as before)
__global__ __launch_bounds__(256) void sa_sample_high_kernel(const uint8_t* __restrict__ text, uint64_t len, uint32_t* __restrict__ flag) {
    bool hi = false;
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w * 16 + 16 <= len; w += (uint64_t)gridDim.x * 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + w * 16);
        hi = hi || ((v.x | v.y | v.z | v.w) & 0x80808080u);
    }
    if (__builtin_amdgcn_ballot_w64(hi) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// MSD-first initial sort, pair form (radix_sort.h: TextGen::msd_pair): how often every pair of symbol CODES (c0, c1) starts
// a suffix, as the number c0 * base + c1 — c1 = 0 where the document ends behind c0.  The sweep ignores document ends (and
// counts nothing for the last byte of the text); sa_docend_pair_kernel moves the last position of every document from the
// pair it was counted under to (c0, 0).  cnt[base^2], wrapping 64-bit adds.
__global__ __launch_bounds__(1024) void sa_paircode_kernel(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ symmap,
                                                           uint32_t kbase, unsigned long long* __restrict__ cnt) {
    extern __shared__ uint32_t s_pair[];  // [kbase * kbase]
    __shared__ uint16_t s_map[256];
    const uint32_t np = kbase * kbase;
    for (uint32_t i = threadIdx.x; i < np; i += 1024) s_pair[i] = 0;
    if (threadIdx.x < 256) s_map[threadIdx.x] = symmap[threadIdx.x];
    __syncthreads();
    const uint64_t words = n / 16;
    const uint4* t4 = reinterpret_cast<const uint4*>(text);
    const uint64_t stride = (uint64_t)gridDim.x * 1024;
    for (uint64_t w = (uint64_t)blockIdx.x * 1024 + threadIdx.x; w < words; w += stride) {
        const uint4 v = t4[w];
        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
        const bool has_next = w * 16 + 16 < n;
        const uint32_t nxt = has_next ? (uint32_t)s_map[text[w * 16 + 16]] : 0u;
        uint32_t c[17];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c[4 * q] = s_map[x[q] & 0xFF];
            c[4 * q + 1] = s_map[(x[q] >> 8) & 0xFF];
            c[4 * q + 2] = s_map[(x[q] >> 16) & 0xFF];
            c[4 * q + 3] = s_map[x[q] >> 24];
        }
        c[16] = nxt;
#pragma unroll
        for (int q = 0; q < 15; ++q) atomicAdd(&s_pair[__umul24(c[q], kbase) + c[q + 1]], 1u);
        if (has_next) atomicAdd(&s_pair[__umul24(c[15], kbase) + c[16]], 1u);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 15)) {
        const uint64_t p = words * 16 + threadIdx.x;
        if (p + 1 < n) atomicAdd(&s_pair[__umul24((uint32_t)s_map[text[p]], kbase) + (uint32_t)s_map[text[p + 1]]], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < np; i += 1024)
        if (s_pair[i]) atomicAdd(&cnt[i], (unsigned long long)s_pair[i]);
}
__global__ __launch_bounds__(256) void sa_docend_pair_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ doc_start,
                                                             uint64_t ndocs, uint64_t n, const uint16_t* __restrict__ symmap, uint32_t kbase,
                                                             unsigned long long* __restrict__ cnt) {
    __shared__ uint32_t s_end[257];  // documents by the code of their last symbol (few hot counters: not straight to memory)
    for (int i = threadIdx.x; i < 257; i += 256) s_end[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x; d < ndocs; d += stride) {
        const uint64_t b0 = doc_start[d], e = doc_start[d + 1];
        if (e == b0) continue;
        const uint32_t c0 = symmap[text[e - 1]];
        atomicAdd(&s_end[c0], 1u);
        if (e < n) atomicAdd(&cnt[c0 * kbase + (uint32_t)symmap[text[e]]], ~0ull);  // (- 1)
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < kbase; c += 256)
        if (s_end[c]) atomicAdd(&cnt[c * kbase], (unsigned long long)s_end[c]);
}

// Look-back-free generated pass of the MSD-first sort (radix_sort.h: TextGen::tile_base): per tile of RS_GEN8_TILE positions, how many
// suffixes carry each TOP digit — the digit is floor((c0 B + c1) / span) of the first two symbol codes, (a R) >> s in the pass's own
// arithmetic (rs_pair_setup).  One workgroup per tile, wave-private LDS counters (the digit has ~200 values: shared counters would
// serialise), the pair of the last position of every document counted with the byte BEHIND the document and moved to (c0, end) by
// sa_tile_docend_fix_kernel.  The column sums of the table are the top-digit histogram (it replaces the pair count).
constexpr uint32_t TPC_TILES_PER_WAVE = 8;  // tiles a wavefront counts one after the other (a workgroup: 4 x 8 tiles)
__global__ __launch_bounds__(256) void sa_tile_paircount_kernel(const uint8_t* __restrict__ text, uint64_t n, const uint16_t* __restrict__ symmap,
                                                                uint32_t kbase, uint32_t pair_r, uint32_t pair_s, uint32_t tiles,
                                                                uint32_t* __restrict__ counts) {
    // every wavefront counts its own tiles into its own counters: no workgroup barrier after the code table is staged
    // (four copies of a wavefront's counters, by lane: the digit has only ~200 values, 64 lanes on one copy collide three deep)
    __shared__ uint32_t s_cnt[4][4][256];
    __shared__ uint16_t s_map[256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    s_map[tid] = symmap[tid];
    __syncthreads();
    uint32_t* cnt = s_cnt[wave][lane & 3];
    uint32_t* call = &s_cnt[wave][0][0];
    for (uint32_t k = 0; k < TPC_TILES_PER_WAVE; ++k) {
        const uint32_t tile = (blockIdx.x * 4u + (uint32_t)wave) * TPC_TILES_PER_WAVE + k;
        if (tile >= tiles) break;  // (uniform per wavefront)
#pragma unroll
        for (int q = 0; q < 16; ++q) call[q * 64 + lane] = 0;
        const uint64_t b0 = (uint64_t)tile * RS_GEN8_TILE;
        constexpr int VEC = RS_GEN8_TILE / 16 / 64;  // 16-byte vectors per lane and tile
#pragma unroll 2
        for (int r = 0; r < VEC; ++r) {
            const uint64_t p = b0 + ((uint64_t)r * 64 + lane) * 16;
            if (p >= n) continue;
            uint32_t c[17];
            if (p + 16 <= n) {
                const uint4 v = *reinterpret_cast<const uint4*>(text + p);
                const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    c[4 * q] = s_map[x[q] & 0xFF];
                    c[4 * q + 1] = s_map[(x[q] >> 8) & 0xFF];
                    c[4 * q + 2] = s_map[(x[q] >> 16) & 0xFF];
                    c[4 * q + 3] = s_map[x[q] >> 24];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) c[q] = p + q < n ? (uint32_t)s_map[text[p + q]] : 0u;
            }
            c[16] = p + 16 < n ? (uint32_t)s_map[text[p + 16]] : 0u;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (p + q < n) {
                    const uint32_t a = __umul24(c[q], kbase) + c[q + 1];
                    atomicAdd(&cnt[__umul24(a, pair_r) >> pair_s], 1u);
                }
            }
        }
        // (the wavefront's own LDS traffic is ordered: its counters are complete when its atomics have returned)
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int dgt = q * 64 + lane;
            counts[(size_t)tile * 256 + dgt] = call[dgt] + call[256 + dgt] + call[512 + dgt] + call[768 + dgt];
        }
        __builtin_amdgcn_wave_barrier();
    }
}
__global__ __launch_bounds__(256) void sa_tile_docend_fix_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ doc_start,
                                                                 uint64_t ndocs, uint64_t n, const uint16_t* __restrict__ symmap, uint32_t kbase,
                                                                 uint32_t pair_r, uint32_t pair_s, uint32_t* __restrict__ counts) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x; d < ndocs; d += stride) {
        const uint64_t b0 = doc_start[d], e = doc_start[d + 1];
        if (e == b0 || e >= n) continue;  // (the last position of the text was counted with "end" already)
        const uint32_t c0 = symmap[text[e - 1]], c1 = symmap[text[e]];
        const uint32_t t_old = __umul24(__umul24(c0, kbase) + c1, pair_r) >> pair_s, t_new = __umul24(__umul24(c0, kbase), pair_r) >> pair_s;
        if (t_old != t_new) {
            uint32_t* row = counts + (size_t)((e - 1) / RS_GEN8_TILE) * 256;
            atomicAdd(row + t_old, ~0u);  // (- 1)
            atomicAdd(row + t_new, 1u);
        }
    }
}

// Reference order, one level below the root (bucket-wise build): how the suffixes of every first byte split by what
// FOLLOWS that byte — a byte below 0x80, a byte from 0x80, or the end of the document.  Inside a first-symbol bucket that
// the reference treats as a radix node its children come [end][0x80..0xFF][0x00..0x7F] (index.h:66-73); with these counts
// the last pass of the segmented sort writes the two byte blocks swapped straight away (no rotation afterwards).
// pair[b][c] = positions p with text[p] = b and text[p + 1] in class c (0: < 0x80 or p + 1 = n, 1: >= 0x80), document
// ends ignored; sa_docend_class_kernel counts, per last byte of a document, what has to move from those to "end".
__global__ __launch_bounds__(256) void sa_pairclass_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                           unsigned long long* __restrict__ pair /*[256][2]*/) {
    __shared__ uint32_t s[512];
    s[threadIdx.x] = 0;
    s[256 + threadIdx.x] = 0;
    __syncthreads();
    const uint64_t words = n / 16;
    const uint4* t4 = reinterpret_cast<const uint4*>(text);
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += stride) {
        const uint4 v = t4[w];
        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
        const uint32_t nxt = w * 16 + 16 < n ? (uint32_t)text[w * 16 + 16] : 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t follow = q < 3 ? x[q + 1] & 0xFFu : nxt;  // the byte behind this dword
            atomicAdd(&s[2 * (x[q] & 0xFF) + ((x[q] >> 15) & 1u)], 1u);
            atomicAdd(&s[2 * ((x[q] >> 8) & 0xFF) + ((x[q] >> 23) & 1u)], 1u);
            atomicAdd(&s[2 * ((x[q] >> 16) & 0xFF) + (x[q] >> 31)], 1u);
            atomicAdd(&s[2 * (x[q] >> 24) + (follow >> 7)], 1u);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 15)) {
        const uint64_t p = words * 16 + threadIdx.x;
        atomicAdd(&s[2 * text[p] + (p + 1 < n ? (uint32_t)(text[p + 1] >> 7) : 0u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256)
        if (s[i]) atomicAdd(&pair[i], (unsigned long long)s[i]);
}
// corr[b][c] (c = 0, 1): documents whose last byte is b and that are followed by a byte of class c (to be taken off
// pair[b][c]); corr[b][2]: documents whose last byte is b (their last suffix ends behind its first symbol)
__global__ __launch_bounds__(256) void sa_docend_class_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ doc_start,
                                                              uint64_t ndocs, uint64_t n, unsigned long long* __restrict__ corr /*[256][3]*/) {
    __shared__ uint32_t s[768];
    for (int i = threadIdx.x; i < 768; i += 256) s[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x; d < ndocs; d += stride) {
        const uint64_t b0 = doc_start[d], e = doc_start[d + 1];
        if (e == b0) continue;
        const uint32_t last = text[e - 1];
        atomicAdd(&s[3 * last + 2], 1u);
        atomicAdd(&s[3 * last + (e < n ? (uint32_t)(text[e] >> 7) : 0u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += 256)
        if (s[i]) atomicAdd(&corr[i], (unsigned long long)s[i]);
}

// Digit histograms of the initial sort without reading any key: with one symbol per digit, the digit of
// pass p is the symbol k = nsym-1-p positions into the suffix, so its histogram is the symbol histogram
// of the text minus the symbols sitting at offsets < k of their document, and the "end" code counts the
// suffixes shorter than k+1.  This kernel gathers those document-head corrections:
//   first[j][c] = documents whose byte at offset j has code c   (j < nsym-1)
//   lencnt[L]   = documents of length L                          (L < nsym)
constexpr int HC_MAXSYM = 16;
__global__ __launch_bounds__(256) void sa_headcorr_kernel(const uint8_t* __restrict__ text,
                                                          const uint64_t* __restrict__ doc_start, uint64_t ndocs,
                                                          const uint16_t* __restrict__ symmap, int nsym,
                                                          unsigned long long* __restrict__ first /*[HC_MAXSYM][257]*/,
                                                          unsigned long long* __restrict__ lencnt /*[HC_MAXSYM]*/) {
    __shared__ uint32_t s_first[HC_MAXSYM * 257];
    __shared__ uint32_t s_len[HC_MAXSYM];
    for (int i = threadIdx.x; i < HC_MAXSYM * 257; i += 256) s_first[i] = 0;
    if (threadIdx.x < HC_MAXSYM) s_len[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x; d < ndocs; d += stride) {
        const uint64_t ds = doc_start[d], len = doc_start[d + 1] - ds;
        if (len < (uint64_t)nsym) atomicAdd(&s_len[len], 1u);
        const int lim = (int)(len < (uint64_t)(nsym - 1) ? len : (uint64_t)(nsym - 1));
        for (int j = 0; j < lim; ++j) atomicAdd(&s_first[j * 257 + symmap[text[ds + j]]], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HC_MAXSYM * 257; i += 256)
        if (s_first[i]) atomicAdd(&first[i], (unsigned long long)s_first[i]);
    if (threadIdx.x < HC_MAXSYM && s_len[threadIdx.x]) atomicAdd(&lencnt[threadIdx.x], (unsigned long long)s_len[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// key-width estimate by sampling: keys of `kmax` symbols for `S` pseudo-random suffixes
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sa_sample_keys_kernel(const uint8_t* __restrict__ text,
                                                             const uint64_t* __restrict__ doc_start, uint64_t ndocs,
                                                             uint64_t n, const uint16_t* __restrict__ symmap, int symbits,
                                                             int kmax, uint64_t S, uint64_t* __restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;  // splitmix64 finaliser -> position
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const uint64_t stride = n / S;               // stratified: one position per stride, all distinct
    const uint64_t p = i * stride + z % stride;  // (a position drawn twice would count as a collision)
    uint64_t lo = 0, hi = ndocs - 1;  // document of p
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (doc_start[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const uint64_t rem = doc_start[lo + 1] - p;
    uint64_t key = 0;
    for (int k = 0; k < kmax; ++k) key = (key << symbits) | ((uint64_t)k < rem ? (uint64_t)symmap[text[p + k]] : 0ull);
    keys[i] = key;
}

// eq[k] = adjacent pairs of the sorted sample that agree on their first k symbols (k = 1 .. kmax)
__global__ __launch_bounds__(256) void sa_sample_count_kernel(const uint64_t* __restrict__ keys, uint64_t S, int symbits,
                                                              int kmax, unsigned long long* __restrict__ eq) {
    __shared__ unsigned int s_eq[32];
    if (threadIdx.x < 32) s_eq[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i + 1 < S) {
        const uint64_t x = keys[i] ^ keys[i + 1];
        // number of leading symbols that agree
        int same = kmax;
        if (x) same = (__clzll((long long)x) - (64 - kmax * symbits)) / symbits;
        // a k-prefix that already contains the end-of-document code identifies the whole suffix: such
        // pairs are equal suffixes (a final group), not unresolved ones — count only "live" prefixes
        int live = 0;
        while (live < kmax && ((keys[i] >> ((kmax - 1 - live) * symbits)) & ((1ull << symbits) - 1ull)) != 0) ++live;
        if (live < same) same = live;
        for (int k = 1; k <= same; ++k) atomicAdd(&s_eq[k], 1u);
    }
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x < 32 && s_eq[threadIdx.x]) atomicAdd(&eq[threadIdx.x], (unsigned long long)s_eq[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// 2. key generation
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t doc_upper(const uint64_t* __restrict__ doc_start, uint64_t lo, uint64_t hi,
                                              uint64_t p) {
    // largest d in [lo, hi] with doc_start[d] <= p (empty documents share a start; the last one wins,
    // which is the non-empty document that really contains p)
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (doc_start[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}

template <typename V>
__global__ __launch_bounds__(256) void sa_keygen_kernel(const uint8_t* __restrict__ text,
                                                        const uint64_t* __restrict__ doc_start, uint64_t ndocs,
                                                        uint64_t n, int bits, const uint16_t* __restrict__ symmap,
                                                        uint32_t kbase, int nsym, bool padded,
                                                        uint64_t* __restrict__ keys, V* __restrict__ vals) {
    __shared__ __attribute__((aligned(16))) uint8_t s_text[KG_TILE + KG_LOOK];
    __shared__ uint16_t s_map[256];
    __shared__ uint64_t s_drange[2];
    const int tid = threadIdx.x;
    const uint64_t p0 = (uint64_t)blockIdx.x * KG_TILE;
    const uint32_t cnt = (uint32_t)((n - p0) < (uint64_t)KG_TILE ? (n - p0) : (uint64_t)KG_TILE);
    s_map[tid] = symmap[tid];
    if (tid == 0) s_drange[0] = doc_upper(doc_start, 0, ndocs - 1, p0);
    if (tid == 1) s_drange[1] = doc_upper(doc_start, 0, ndocs - 1, p0 + cnt - 1);
    // stage the tile (+ look-ahead) in LDS with 16-byte loads; the tail of a caller-owned buffer that
    // has no padding is fetched bytewise
    for (uint32_t i = tid * 16; i < KG_TILE + KG_LOOK; i += 256 * 16) {
        const uint64_t g = p0 + i;
        if (padded ? (g < n + KG_LOOK) : (g + 16 <= n)) {
            *reinterpret_cast<uint4*>(&s_text[i]) = *reinterpret_cast<const uint4*>(text + g);
        } else {
#pragma unroll
            for (int b = 0; b < 16; ++b) s_text[i + b] = (g + b < n) ? text[g + b] : (uint8_t)0;
        }
    }
    __syncthreads();
    const uint64_t dlo = s_drange[0], dhi = s_drange[1];
#pragma unroll
    for (int j = 0; j < KG_TILE / 256; ++j) {
        const uint32_t li = j * 256 + tid;
        if (li >= cnt) break;
        const uint64_t p = p0 + li;
        const uint64_t d = doc_upper(doc_start, dlo, dhi, p);
        const uint64_t ds = doc_start[d];
        const uint64_t rem = doc_start[d + 1] - p;
        uint64_t key = 0;
        for (int k = 0; k < nsym; ++k) {
            const uint64_t sym = (uint64_t)k < rem ? (uint64_t)s_map[s_text[li + k]] : 0ull;
            key = key * kbase + sym;
        }
        keys[p] = key;
        vals[p] = (V)(((p - ds) << bits) | d);
    }
}

// Digit histograms of all LSD passes for keys whose digits are not whole symbols (dense base-(alphabet+1)
// keys): the keys are generated from the text exactly like the first radix pass generates them, counted
// and thrown away.  hist[pass][256], 8-bit digits.
constexpr int KH_TILE = 8192;
constexpr int KH_CHUNK = KH_TILE / 256;  // consecutive positions per thread
__global__ __launch_bounds__(256) void sa_keyhist_kernel(const uint8_t* __restrict__ text,
                                                         const uint64_t* __restrict__ doc_start, uint64_t ndocs,
                                                         uint64_t n, const uint16_t* __restrict__ symmap, uint32_t kbase,
                                                         int nsym, int npass, bool padded,
                                                         unsigned long long* __restrict__ hist, uint32_t digit_mask) {
    // digit_mask: bit q set = digit q is counted (the MSD-first sort only needs the top digit)
    __shared__ __attribute__((aligned(16))) uint8_t s_code[KH_TILE + KG_LOOK];
    __shared__ uint16_t s_map[256];
    __shared__ uint64_t s_drange[2];
    __shared__ uint32_t s_hist[8][256];  // <= 8 digits of 8 bits in a 64-bit key
    const int tid = threadIdx.x;
    s_map[tid] = symmap[tid];
    for (int p = 0; p < npass; ++p) s_hist[p][tid] = 0;
    uint64_t top = 1;  // base^(nsym-1): weight of the symbol that leaves the window
    for (int q = 1; q < nsym; ++q) top *= kbase;
    const uint64_t tiles = (n + KH_TILE - 1) / KH_TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint64_t p0 = tile * KH_TILE;
        const uint32_t cnt = (uint32_t)((n - p0) < (uint64_t)KH_TILE ? (n - p0) : (uint64_t)KH_TILE);
        __syncthreads();  // s_map ready / previous tile consumed
        if (tid == 0) s_drange[0] = doc_upper(doc_start, 0, ndocs - 1, p0);
        if (tid == 64) s_drange[1] = doc_upper(doc_start, 0, ndocs - 1, p0 + cnt - 1);
        for (uint32_t i = tid * 16; i < KH_TILE + KG_LOOK; i += 256 * 16) {
            const uint64_t g = p0 + i;
            uint32_t x[4];
            if (padded ? (g < n + KG_LOOK) : (g + 16 <= n)) {
                const uint4 w = *reinterpret_cast<const uint4*>(text + g);
                x[0] = w.x; x[1] = w.y; x[2] = w.z; x[3] = w.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x[q] = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        x[q] |= (uint32_t)((g + 4 * q + b < n) ? text[g + 4 * q + b] : (uint8_t)0) << (8 * b);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                x[q] = (uint32_t)s_map[x[q] & 0xFF] | ((uint32_t)s_map[(x[q] >> 8) & 0xFF] << 8) |
                       ((uint32_t)s_map[(x[q] >> 16) & 0xFF] << 16) | ((uint32_t)s_map[x[q] >> 24] << 24);
            *reinterpret_cast<uint4*>(&s_code[i]) = make_uint4(x[0], x[1], x[2], x[3]);
        }
        __syncthreads();
        // Each thread walks KH_CHUNK consecutive positions with a rolling key: inside one document
        //   key(p + 1) = (key(p) - code(p) * base^(nsym-1)) * base + code(p + nsym)   [0 behind the document end]
        // so only the first position of a chunk or of a document pays for a full Horner evaluation.
        const uint64_t dhi = s_drange[1];
        const uint32_t* s_words = reinterpret_cast<const uint32_t*>(s_code);
        const uint32_t q0 = tid * KH_CHUNK;
        if (q0 < cnt) {
            uint64_t d = doc_upper(doc_start, s_drange[0], dhi, p0 + q0);
            uint64_t dend = doc_start[d + 1];
            uint64_t key = 0;
            bool fresh = true;
            const uint32_t qe = q0 + KH_CHUNK < cnt ? q0 + KH_CHUNK : cnt;
            for (uint32_t li = q0; li < qe; ++li) {
                const uint64_t p = p0 + li;
                if (p >= dend) {  // next non-empty document
                    do { ++d; dend = doc_start[d + 1]; } while (p >= dend);
                    fresh = true;
                }
                if (fresh) {
                    key = rs_pack_key(s_words, li, nsym, kbase, dend - p < (1ull << 30) ? (uint32_t)(dend - p) : (1u << 30));
                    fresh = false;
                } else {
                    const uint64_t cin = p + (uint64_t)nsym <= dend ? (uint64_t)s_code[li + nsym - 1] : 0ull;
                    key = (key - (uint64_t)s_code[li - 1] * top) * kbase + cin;
                }
                const uint64_t dk = key;
                for (int q = 0; q < npass; ++q)
                    if ((digit_mask >> q) & 1u) atomicAdd(&s_hist[q][(uint32_t)(dk >> (8 * q)) & 0xFFu], 1u);
            }
        }
    }
    __syncthreads();
    for (int p = 0; p < npass; ++p)
        if (s_hist[p][tid]) atomicAdd(&hist[p * 256 + tid], (unsigned long long)s_hist[p][tid]);
}

// The same histograms for keys of 3 P symbols (P = 2..5; what the key-width choice produces for byte alphabets:
// 6 symbols for printable ASCII and UTF-8 text, 9 for the 64-symbol Zipf alphabet), with most of the arithmetic in
// 24 bits: a key is the number (G(p) W + G(p + 3)) W + ... with W = base^3 <= 2^24 and G(p) = the three codes at p
// as a base-`base` number — two full-rate v_mad_u32_u24 per position and ONE 64-bit multiply-add per further part
// instead of a rolling 64-bit key (64-bit multiplies run at quarter rate).  G(p) is evaluated afresh for every
// position, so a document end only spoils the nsym - 1 keys whose window crosses it: a thread whose 32 positions
// (+ look-ahead) meet at most one document end runs straight-line code and re-does those few keys with the masked
// Horner evaluation; threads that see more ends (tiny documents, empty documents) and the ragged last tile walk
// position by position like sa_keyhist_kernel.
constexpr int KH3_PER = 32;
static_assert(KH_TILE == 256 * KH3_PER, "sa_keyhist3_kernel: one thread walks KH3_PER positions");
template <int P>
__global__ __launch_bounds__(256) void sa_keyhist3_kernel(const uint8_t* __restrict__ text,
                                                          const uint64_t* __restrict__ doc_start, uint64_t ndocs,
                                                          uint64_t n, const uint16_t* __restrict__ symmap, uint32_t kbase,
                                                          int npass, bool padded, unsigned long long* __restrict__ hist,
                                                          uint32_t digit_mask) {
    constexpr int NSYM = 3 * P;
    constexpr int NG = KH3_PER + 3 * (P - 1);  // G values a thread needs
    static_assert(NG + 2 <= 48, "window of three 16-byte reads");
    __shared__ __attribute__((aligned(16))) uint8_t s_code[KH_TILE + KG_LOOK];
    __shared__ uint16_t s_map[256];
    __shared__ uint64_t s_drange[2];
    __shared__ uint32_t s_hist[8][256];
    const int tid = threadIdx.x;
    s_map[tid] = symmap[tid];
    for (int p = 0; p < npass; ++p) s_hist[p][tid] = 0;
    const uint32_t W = kbase * kbase * kbase;
    const uint64_t tiles = (n + KH_TILE - 1) / KH_TILE;
    auto count = [&](uint64_t key) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < npass && ((digit_mask >> q) & 1u)) atomicAdd(&s_hist[q][(uint32_t)(key >> (8 * q)) & 0xFFu], 1u);
    };
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint64_t p0 = tile * KH_TILE;
        const uint32_t cnt = (uint32_t)((n - p0) < (uint64_t)KH_TILE ? (n - p0) : (uint64_t)KH_TILE);
        __syncthreads();  // s_map ready / previous tile consumed
        if (tid == 0) s_drange[0] = doc_upper(doc_start, 0, ndocs - 1, p0);
        if (tid == 64) s_drange[1] = doc_upper(doc_start, 0, ndocs - 1, p0 + cnt - 1);
        for (uint32_t i = tid * 16; i < KH_TILE + KG_LOOK; i += 256 * 16) {
            const uint64_t g = p0 + i;
            uint32_t x[4];
            if (padded ? (g < n + KG_LOOK) : (g + 16 <= n)) {
                const uint4 w = *reinterpret_cast<const uint4*>(text + g);
                x[0] = w.x; x[1] = w.y; x[2] = w.z; x[3] = w.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x[q] = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        x[q] |= (uint32_t)((g + 4 * q + b < n) ? text[g + 4 * q + b] : (uint8_t)0) << (8 * b);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                x[q] = (uint32_t)s_map[x[q] & 0xFF] | ((uint32_t)s_map[(x[q] >> 8) & 0xFF] << 8) |
                       ((uint32_t)s_map[(x[q] >> 16) & 0xFF] << 16) | ((uint32_t)s_map[x[q] >> 24] << 24);
            *reinterpret_cast<uint4*>(&s_code[i]) = make_uint4(x[0], x[1], x[2], x[3]);
        }
        __syncthreads();
        const uint32_t q0 = tid * KH3_PER;
        if (q0 >= cnt) continue;
        const uint64_t dhi = s_drange[1];
        const uint32_t* s_words = reinterpret_cast<const uint32_t*>(s_code);
        const uint64_t pq = p0 + q0;
        uint64_t d = doc_upper(doc_start, s_drange[0], dhi, pq);
        uint64_t dend = doc_start[d + 1];
        const uint64_t dend2 = d + 2 <= ndocs ? doc_start[d + 2] : ~0ull;
        constexpr uint32_t REACH = KH3_PER + NSYM;
        const uint32_t E = dend - pq < (1ull << 20) ? (uint32_t)(dend - pq) : (1u << 20);  // positions left in this document
        const bool fast = q0 + KH3_PER <= cnt && (E >= REACH || dend2 - pq >= (uint64_t)REACH);
        if (fast) {
            uint32_t w[12];
            {
                const uint4 a = *reinterpret_cast<const uint4*>(&s_code[q0]);
                const uint4 b = *reinterpret_cast<const uint4*>(&s_code[q0 + 16]);
                const uint4 c = *reinterpret_cast<const uint4*>(&s_code[q0 + 32]);
                w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
                w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
            }
            auto code = [&](int t) -> uint32_t { return (w[t >> 2] >> (8 * (t & 3))) & 0xFFu; };
            uint32_t G[NG];
#pragma unroll
            for (int t = 0; t < NG; ++t) G[t] = __umul24(__umul24(code(t), kbase) + code(t + 1), kbase) + code(t + 2);
#pragma unroll
            for (int t = 0; t < KH3_PER; ++t) {
                // the keys of the nsym - 1 positions in front of the document end are counted below
                if ((uint32_t)(E - 1u - (uint32_t)t) >= (uint32_t)(NSYM - 1)) {
                    uint64_t key = G[t];
#pragma unroll
                    for (int k = 1; k < P; ++k) key = key * W + G[t + 3 * k];
                    count(key);
                }
            }
            if (E < REACH) {
                const uint32_t lo = E > (uint32_t)(NSYM - 1) ? E - (uint32_t)(NSYM - 1) : 0u;
                const uint32_t hi = E < (uint32_t)KH3_PER ? E : (uint32_t)KH3_PER;
                for (uint32_t t = lo; t < hi; ++t) count(rs_pack_key(s_words, q0 + t, NSYM, kbase, E - t));
            }
        } else {
            const uint32_t qe = q0 + KH3_PER < cnt ? q0 + KH3_PER : cnt;
            for (uint32_t li = q0; li < qe; ++li) {
                const uint64_t p = p0 + li;
                if (p >= dend) {  // next non-empty document
                    do { ++d; dend = doc_start[d + 1]; } while (p >= dend);
                }
                count(rs_pack_key(s_words, li, NSYM, kbase, dend - p < (1ull << 30) ? (uint32_t)(dend - p) : (1u << 30)));
            }
        }
    }
    __syncthreads();
    for (int p = 0; p < npass; ++p)
        if (s_hist[p][tid]) atomicAdd(&hist[p * 256 + tid], (unsigned long long)s_hist[p][tid]);
}

// ---------------------------------------------------------------------------------------------
// group flags: bit0 = first entry of a group of equal prefixes, bit1 = still unresolved
// ---------------------------------------------------------------------------------------------
// k[0] = predecessor, k[1..4] = the four keys of slots i0..i0+3, k[5] = successor -> four flag bytes
__device__ __forceinline__ uint32_t sa_flags_of(const uint64_t (&k)[6], uint64_t i0, uint64_t n, uint32_t kbase,
                                                uint64_t kmagic) {
    uint32_t out = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint64_t i = i0 + q;
        if (i >= n) break;
        const bool head = i == 0 || k[q] != k[q + 1];
        const bool tail = i + 1 == n || k[q + 2] != k[q + 1];
        // an end-of-document code (0) as the last symbol: key = 0 mod base.  kmagic = floor(2^64 / base) + 1
        // makes the quotient one multiply (exact for keys < 2^56); 0 for a power-of-two base
        const uint64_t kq = k[q + 1];
        const bool exhausted = kmagic ? (kq - __umul64hi(kq, kmagic) * kbase) == 0 : (kq & (uint64_t)(kbase - 1u)) == 0;
        out |= (uint32_t)((head ? 1 : 0) | ((!(head && tail) && !exhausted) ? 2 : 0)) << (8 * q);
    }
    return out;
}

__global__ __launch_bounds__(256) void sa_initflags_kernel(const uint64_t* __restrict__ keys, uint64_t n,
                                                           uint32_t kbase, uint64_t kmagic,
                                                           uint8_t* __restrict__ flags, bool flags_aligned) {
    // four consecutive keys per thread (two 16-byte loads), four flag bytes in one store (bytewise when the
    // flag range of a bucket does not start on a 4-byte boundary)
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    uint64_t k[6];  // k[0] = predecessor, k[1..4] = own, k[5] = successor
    if (i0 + 4 <= n) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(keys + i0);
        const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(keys + i0 + 2);
        k[1] = a.x; k[2] = a.y; k[3] = b.x; k[4] = b.y;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) k[1 + q] = i0 + q < n ? keys[i0 + q] : 0;
    }
    k[0] = i0 > 0 ? keys[i0 - 1] : ~k[1];
    k[5] = i0 + 4 < n ? keys[i0 + 4] : 0;
    const uint32_t out = sa_flags_of(k, i0, n, kbase, kmagic);
    if (flags_aligned && i0 + 4 <= n) {
        *reinterpret_cast<uint32_t*>(flags + i0) = out;
    } else {
        for (int q = 0; q < 4 && i0 + q < n; ++q) flags[i0 + q] = (uint8_t)(out >> (8 * q));
    }
}

// the same for 32-bit keys, optionally extended by the separately stored low digit (split sort):
// key = (k32 << low_bits) | low
template <typename W>
__global__ __launch_bounds__(256) void sa_initflags32_kernel(const uint32_t* __restrict__ k32,
                                                             const W* __restrict__ low, int low_bits, uint64_t n,
                                                             uint32_t kbase, uint64_t kmagic, uint8_t* __restrict__ flags,
                                                             bool flags_aligned, U2* __restrict__ tile_sums = nullptr) {
    // tile_sums (optional; `flags` is then the whole array): per scan tile of SC_TILE flags the number of unresolved
    // entries and of unresolved group heads — the sums the first compaction would otherwise re-read every flag for
    __shared__ uint32_t s_sum[2];
    if (tile_sums) {  // (uniform)
        if (threadIdx.x < 2) s_sum[threadIdx.x] = 0;
        __syncthreads();
    }
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 < n) {
        const uint64_t lmask = (1ull << low_bits) - 1ull;  // (bytes of W above the digits carry other data: packed entries)
        auto full = [&](uint64_t i) -> uint64_t {
            return low ? (((uint64_t)k32[i] << low_bits) | ((uint64_t)low[i] & lmask)) : (uint64_t)k32[i];
        };
        uint64_t k[6];
        if (i0 + 4 <= n) {
            const uint4 a = *reinterpret_cast<const uint4*>(k32 + i0);
            const uint32_t hi[4] = {a.x, a.y, a.z, a.w};
            W l4[4] = {};
            if (low) {
                if constexpr (sizeof(W) == 1) *reinterpret_cast<uint32_t*>(l4) = *reinterpret_cast<const uint32_t*>(low + i0);
                else if constexpr (sizeof(W) == 2) *reinterpret_cast<uint2*>(l4) = *reinterpret_cast<const uint2*>(low + i0);
                else *reinterpret_cast<uint4*>(l4) = *reinterpret_cast<const uint4*>(low + i0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) k[1 + q] = low ? (((uint64_t)hi[q] << low_bits) | ((uint64_t)l4[q] & lmask)) : (uint64_t)hi[q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) k[1 + q] = i0 + q < n ? full(i0 + q) : 0;
        }
        k[0] = i0 > 0 ? full(i0 - 1) : ~k[1];
        k[5] = i0 + 4 < n ? full(i0 + 4) : 0;
        const uint32_t out = sa_flags_of(k, i0, n, kbase, kmagic);
        if (flags_aligned && i0 + 4 <= n) {
            *reinterpret_cast<uint32_t*>(flags + i0) = out;
        } else {
            for (int q = 0; q < 4 && i0 + q < n; ++q) flags[i0 + q] = (uint8_t)(out >> (8 * q));
        }
        if (tile_sums) {
            const uint32_t unres = (out >> 1) & 0x01010101u;  // (sa_flags_of leaves the bytes behind n zero)
            const uint32_t a = __popc(unres), b = __popc(unres & out);
            if (a) atomicAdd(&s_sum[0], a);
            if (b) atomicAdd(&s_sum[1], b);
        }
    }
    if (tile_sums) {
        __syncthreads();
        if (threadIdx.x == 0 && s_sum[0]) {  // (a workgroup's 1024 flags lie inside one scan tile)
            U2* t = tile_sums + ((uint64_t)blockIdx.x * 1024) / SC_TILE;
            atomicAdd(reinterpret_cast<unsigned long long*>(&t->a), (unsigned long long)s_sum[0]);
            if (s_sum[1]) atomicAdd(reinterpret_cast<unsigned long long*>(&t->b), (unsigned long long)s_sum[1]);
        }
    }
}

// keys of one first-symbol bucket, gathered from the text for entries that already sit in text order
// (streamed bucket-wise sort: the n keys are never materialised together)
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;  // gfx950 global loads need no alignment
template <typename V>
__global__ __launch_bounds__(256) void sa_bucket_keys_kernel(const V* __restrict__ ent, uint64_t cnt,
                                                             const uint8_t* __restrict__ text, uint64_t n,
                                                             const uint64_t* __restrict__ doc_start,
                                                             const uint16_t* __restrict__ symmap, int bits, uint64_t mask,
                                                             int nsym, int symbits, uint64_t* __restrict__ keys) {
    __shared__ uint16_t s_map[256];
    s_map[threadIdx.x] = symmap[threadIdx.x];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt) return;
    const uint64_t e = (uint64_t)ent[i];
    const uint64_t d = e & mask;
    const uint64_t pos = doc_start[d] + (e >> bits);
    const uint64_t rem = doc_start[d + 1] - pos;
    uint64_t key = 0;
    if (pos + 16 <= n) {  // two 8-byte windows instead of nsym byte loads
        uint64_t w = *reinterpret_cast<const u64_unaligned*>(text + pos);
        if (nsym > 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                key = (key << symbits) | ((uint64_t)k < rem ? (uint64_t)s_map[w & 0xff] : 0ull);
                w >>= 8;
            }
            w = *reinterpret_cast<const u64_unaligned*>(text + pos + 8);
            for (int k = 8; k < nsym; ++k) {
                key = (key << symbits) | ((uint64_t)k < rem ? (uint64_t)s_map[w & 0xff] : 0ull);
                w >>= 8;
            }
        } else {
            for (int k = 0; k < nsym; ++k) {
                key = (key << symbits) | ((uint64_t)k < rem ? (uint64_t)s_map[w & 0xff] : 0ull);
                w >>= 8;
            }
        }
    } else {
        for (int k = 0; k < nsym; ++k) {
            const uint64_t sym = (uint64_t)k < rem ? (uint64_t)s_map[text[pos + k]] : 0ull;
            key = (key << symbits) | sym;
        }
    }
    keys[i] = key;
}

// Bucket records for the bucket-wise sort of big corpora.  Inside a first-symbol bucket the sort key is the
// symbols BEHIND the first one as a number in base kbase, split into (u32 key >> low_bits, low digits) like
// the records of the single-sort path.  Gathering them bucket by bucket would sweep the whole text once per
// bucket (every 64-byte line holds suffixes of ~50 different buckets: 64 n bytes of traffic), so the gather
// runs over GROUPS of buckets and in text order: a work item is (bucket, <= BR_ITEM consecutive entries of
// it) and the items are sorted by text chunk first — all buckets walk one 2 MiB stretch of text (which
// then sits in every XCD's L2) before anybody moves on.  The per-bucket digit histograms of all sort
// passes fall out of the same kernel.
constexpr uint32_t BR_ITEM = 4096;
struct BucketItem {
    unsigned long long begin;  // index into the entry array
    uint32_t count, bucket;
};

// bounds[b * (nch + 1) + x] = first index in bucket b's entry range whose text position is >= x * chunk
template <typename V>  // (V = storage tag of the partitioned entries: uint32_t / uint64_t / Packed40)
__global__ __launch_bounds__(256) void sa_bucket_bounds_kernel(typename SaOf<V>::ptr ent,
                                                               const unsigned long long* __restrict__ bstart /*[nb + 1]*/,
                                                               uint32_t nb, uint32_t nch, uint64_t chunk,
                                                               const uint64_t* __restrict__ doc_start, int bits, uint64_t mask,
                                                               unsigned long long* __restrict__ bounds) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)nb * (nch + 1)) return;
    const uint32_t b = (uint32_t)(t / (nch + 1)), x = (uint32_t)(t % (nch + 1));
    uint64_t lo = bstart[b], hi = bstart[b + 1];
    const uint64_t target = (uint64_t)x * chunk;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        const uint64_t e = (uint64_t)ent[mid];
        const uint64_t pos = doc_start[e & mask] + (e >> bits);
        if (pos < target) lo = mid + 1; else hi = mid;
    }
    bounds[t] = lo;
}

template <typename V, typename W, typename K>
__global__ __launch_bounds__(256) void sa_bucket_records_kernel(const V* __restrict__ ent,
                                                                const BucketItem* __restrict__ items,
                                                                const uint8_t* __restrict__ text, uint64_t n,
                                                                const uint64_t* __restrict__ doc_start,
                                                                const uint16_t* __restrict__ symmap, int bits, uint64_t mask,
                                                                int nsym, uint32_t kbase, int low_bits, int npass,
                                                                uint64_t gstart, uint32_t bucket0, K* __restrict__ k32,
                                                                W* __restrict__ low, unsigned long long* __restrict__ hist) {
    __shared__ uint16_t s_map[256];
    __shared__ uint32_t s_hist[8][256];
    s_map[threadIdx.x] = symmap[threadIdx.x];
    for (int p = 0; p < npass; ++p) s_hist[p][threadIdx.x] = 0;
    __syncthreads();
    const BucketItem it = items[blockIdx.x];
    for (uint32_t r = threadIdx.x; r < it.count; r += 256) {
        const uint64_t i = it.begin + r;
        const uint64_t e = (uint64_t)ent[i];
        const uint64_t d = e & mask;
        const uint64_t pos = doc_start[d] + (e >> bits);
        const uint64_t rem = doc_start[d + 1] - pos;
        uint64_t key = 0;
        if (pos + 24 <= n) {  // symbols 1 .. nsym-1 from two 8-byte windows behind the first byte
            uint64_t w = *reinterpret_cast<const u64_unaligned*>(text + pos + 1);
            for (int k = 1; k < nsym && k <= 8; ++k) {
                key = key * kbase + ((uint64_t)k < rem ? (uint64_t)s_map[w & 0xff] : 0ull);
                w >>= 8;
            }
            if (nsym > 9) {
                w = *reinterpret_cast<const u64_unaligned*>(text + pos + 9);
                for (int k = 9; k < nsym; ++k) {
                    key = key * kbase + ((uint64_t)k < rem ? (uint64_t)s_map[w & 0xff] : 0ull);
                    w >>= 8;
                }
            }
        } else {
            for (int k = 1; k < nsym; ++k) key = key * kbase + ((uint64_t)k < rem ? (uint64_t)s_map[text[pos + k]] : 0ull);
        }
        k32[i - gstart] = (K)(key >> low_bits);
        if constexpr (!std::is_same<W, NoVal>::value) low[i - gstart] = (W)(key & ((1ull << low_bits) - 1ull));
        for (int q = 0; q < npass; ++q) atomicAdd(&s_hist[q][(uint32_t)(key >> (8 * q)) & 0xFFu], 1u);
    }
    __syncthreads();
    unsigned long long* h = hist + (size_t)(it.bucket - bucket0) * 8 * 256;
    for (int p = 0; p < npass; ++p)
        if (s_hist[p][threadIdx.x]) atomicAdd(&h[p * 256 + threadIdx.x], (unsigned long long)s_hist[p][threadIdx.x]);
}

// The same records with PACKED entries (entries below 2^40, i.e. every corpus that fits a GPU): the sorts of a
// bucket then move (u32 key, u32 entry bits 0..31, W = low digits | entry bits 32..39 above them) — 9 to 12 bytes per
// suffix and pass instead of 13 to 16 with 8-byte entries — and sa_assemble_entries_kernel puts the sorted
// entries back together.
template <typename W>
__global__ __launch_bounds__(256) void sa_bucket_records_packed_kernel(const uint64_t* __restrict__ ent,
                                                                       const BucketItem* __restrict__ items,
                                                                       const uint8_t* __restrict__ text, uint64_t n,
                                                                       const uint64_t* __restrict__ doc_start,
                                                                       const uint16_t* __restrict__ symmap, int bits,
                                                                       uint64_t mask, int nsym, uint32_t kbase, int low_bits,
                                                                       int npass, uint64_t gstart, uint32_t bucket0,
                                                                       uint32_t* __restrict__ k32, W* __restrict__ low,
                                                                       uint32_t* __restrict__ elo,
                                                                       unsigned long long* __restrict__ hist) {
    __shared__ uint16_t s_map[256];
    __shared__ uint32_t s_hist[8][256];
    s_map[threadIdx.x] = symmap[threadIdx.x];
    for (int p = 0; p < npass; ++p) s_hist[p][threadIdx.x] = 0;
    __syncthreads();
    const BucketItem it = items[blockIdx.x];
    // Four entries per thread and round, stage by stage: entry -> document table -> text window are dependent loads,
    // and the four chains of a thread are in flight together.
    constexpr int U = 4;
    for (uint32_t r0 = threadIdx.x; r0 < it.count; r0 += 256 * U) {
        uint64_t e[U], pos[U], rem[U], w0[U], w1[U];
        bool live[U], windowed[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = r0 + 256u * u < it.count;
            e[u] = live[u] ? ent[it.begin + r0 + 256u * u] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t d = live[u] ? (e[u] & mask) : 0ull;
            const uint64_t ds = doc_start[d], de = doc_start[d + 1];
            pos[u] = ds + (e[u] >> bits);
            rem[u] = de - pos[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            windowed[u] = live[u] && pos[u] + 24 <= n;  // symbols 1 .. nsym-1 from two 8-byte windows behind the first byte
            w0[u] = windowed[u] ? *reinterpret_cast<const u64_unaligned*>(text + pos[u] + 1) : 0ull;
            w1[u] = windowed[u] && nsym > 9 ? *reinterpret_cast<const u64_unaligned*>(text + pos[u] + 9) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            uint64_t key = 0;
            if (windowed[u]) {
                uint64_t w = w0[u];
                for (int k = 1; k < nsym && k <= 8; ++k) {
                    key = key * kbase + ((uint64_t)k < rem[u] ? (uint64_t)s_map[w & 0xff] : 0ull);
                    w >>= 8;
                }
                w = w1[u];
                for (int k = 9; k < nsym; ++k) {
                    key = key * kbase + ((uint64_t)k < rem[u] ? (uint64_t)s_map[w & 0xff] : 0ull);
                    w >>= 8;
                }
            } else {
                for (int k = 1; k < nsym; ++k)
                    key = key * kbase + ((uint64_t)k < rem[u] ? (uint64_t)s_map[text[pos[u] + k]] : 0ull);
            }
            const uint64_t i = it.begin + r0 + 256u * u;
            k32[i - gstart] = (uint32_t)(key >> low_bits);
            low[i - gstart] = (W)((key & ((1ull << low_bits) - 1ull)) | ((e[u] >> 32) << low_bits));
            elo[i - gstart] = (uint32_t)e[u];
            for (int q = 0; q < npass; ++q) atomicAdd(&s_hist[q][(uint32_t)(key >> (8 * q)) & 0xFFu], 1u);
        }
    }
    __syncthreads();
    unsigned long long* h = hist + (size_t)(it.bucket - bucket0) * 8 * 256;
    for (int p = 0; p < npass; ++p)
        if (s_hist[p][threadIdx.x]) atomicAdd(&h[p * 256 + threadIdx.x], (unsigned long long)s_hist[p][threadIdx.x]);
}

// ---- device-side planning of the record gather + a persistent, XCD-aware gather (segmented bucket-wise build) ----
// The work items of a bucket group are (bucket, <= BR_ITEM consecutive entries of it inside one text chunk), walked in
// text-chunk order so that the chunk sits in the L2 while every bucket reads it.  The 8 XCDs have private L2s, so the
// chunks are dealt out to them round robin: list x = the chunks x, x + 8, ... (chunk-major, bucket-minor), and only the
// workgroups running on XCD x (blockIdx % 8, as dispatched) draw from list x — a chunk crosses the fabric once instead
// of once per XCD.  Item lists are laid out on the device (no host loop over millions of items, no copy of the bounds
// back to the host, no synchronisation per group).
//   cells of list x: c = ci * gb + bi  <->  chunk x + 8 ci, bucket b0 + bi;  cell_base[x] = cells in front of list x
struct GatherPlan {
    uint32_t cell_base[9];
};

// block x scans the item counts of its list: cell_off[cell_base[x] + c] = items in front of cell c, list_len[x] = total
__global__ __launch_bounds__(1024) void sa_gather_plan_kernel(const unsigned long long* __restrict__ bounds, uint32_t nch, uint32_t b0,
                                                              uint32_t gb, GatherPlan plan, uint32_t* __restrict__ cell_off,
                                                              uint32_t* __restrict__ list_len) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const uint32_t x = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cells = plan.cell_base[x + 1] - plan.cell_base[x];
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < cells; c0 += 1024) {
        const uint32_t c = c0 + tid;
        uint32_t v = 0;
        if (c < cells) {
            const uint32_t ch = x + 8u * (c / gb), b = b0 + c % gb;
            const unsigned long long lo = bounds[(size_t)b * (nch + 1) + ch], hi = bounds[(size_t)b * (nch + 1) + ch + 1];
            v = (uint32_t)((hi - lo + BR_ITEM - 1) / BR_ITEM);
        }
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += y;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint32_t pre = s_carry;
        for (uint32_t w = 0; w < wave; ++w) pre += s_w[w];
        if (c < cells) cell_off[plan.cell_base[x] + c] = pre + incl - v;
        __syncthreads();
        if (tid == 1023) s_carry = pre + incl;
        __syncthreads();
    }
    if (tid == 0) list_len[x] = s_carry;
}

// one thread per cell writes the cell's items; items of list x start behind the lists in front of it
__global__ __launch_bounds__(256) void sa_gather_emit_kernel(const unsigned long long* __restrict__ bounds, uint32_t nch, uint32_t b0,
                                                             uint32_t gb, GatherPlan plan, const uint32_t* __restrict__ cell_off,
                                                             const uint32_t* __restrict__ list_len, BucketItem* __restrict__ items) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= plan.cell_base[8]) return;
    uint32_t x = 0;
    while (t >= plan.cell_base[x + 1]) ++x;
    const uint32_t c = t - plan.cell_base[x];
    uint32_t at = cell_off[t];
    for (uint32_t y = 0; y < x; ++y) at += list_len[y];
    const uint32_t ch = x + 8u * (c / gb), b = b0 + c % gb;
    const unsigned long long lo = bounds[(size_t)b * (nch + 1) + ch], hi = bounds[(size_t)b * (nch + 1) + ch + 1];
    for (unsigned long long o = lo; o < hi; o += BR_ITEM, ++at)
        items[at] = BucketItem{o, (uint32_t)((hi - o) < (unsigned long long)BR_ITEM ? (hi - o) : (unsigned long long)BR_ITEM), b};
}

// the gather of sa_bucket_records_packed_kernel, persistent: workgroup w serves list w % 8 until it is empty
template <typename W, typename ET = uint64_t>  // (ET = storage tag of the partitioned entries: uint64_t or Packed40)
__global__ __launch_bounds__(256) void sa_bucket_records_lists_kernel(typename SaOf<ET>::ptr ent, const BucketItem* __restrict__ items,
                                                                      const uint32_t* __restrict__ list_len, uint32_t* __restrict__ tickets,
                                                                      const uint8_t* __restrict__ text, uint64_t n,
                                                                      const uint64_t* __restrict__ doc_start,
                                                                      const uint16_t* __restrict__ symmap, int bits, uint64_t mask, int nsym,
                                                                      uint32_t kbase, int low_bits, int npass, uint64_t gstart,
                                                                      uint32_t bucket0, uint32_t* __restrict__ k32, W* __restrict__ low,
                                                                      uint32_t* __restrict__ elo, unsigned long long* __restrict__ hist) {
    constexpr int abl = RS_GATHER_ABL;  // (0 in every product build: timing-only ablations are compiled in with -DRS_GATHER_ABL=<bits>)
    __shared__ uint16_t s_map[256];
    __shared__ uint32_t s_hist[8][256];
    __shared__ uint32_t s_slot;
    const uint32_t x = blockIdx.x & 7u;
    s_map[threadIdx.x] = symmap[threadIdx.x];
    uint32_t first = 0;
    for (uint32_t y = 0; y < x; ++y) first += list_len[y];
    const uint32_t len = list_len[x];
    constexpr int U = 4;
    for (;;) {
        __syncthreads();  // (s_map ready / previous item's histogram flushed)
        if (threadIdx.x == 0) s_slot = atomicAdd(tickets + x, 1u);
        for (int p = 0; p < npass; ++p) s_hist[p][threadIdx.x] = 0;
        __syncthreads();
        const uint32_t slot = s_slot;
        if (slot >= len) break;
        const BucketItem it = items[first + slot];
        for (uint32_t r0 = threadIdx.x; r0 < it.count; r0 += 256 * U) {
            uint64_t e[U], pos[U], rem[U], w0[U], w1[U];
            bool live[U], windowed[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                live[u] = r0 + 256u * u < it.count;
                e[u] = live[u] ? ent[it.begin + r0 + 256u * u] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t d = live[u] ? (e[u] & mask) : 0ull;
                if (abl & 2) {  // (timing experiment: no document table)
                    pos[u] = (e[u] >> bits) + d * 1024;
                    rem[u] = 1u << 20;
                    continue;
                }
                const uint64_t ds = doc_start[d], de = doc_start[d + 1];
                pos[u] = ds + (e[u] >> bits);
                rem[u] = de - pos[u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                windowed[u] = live[u] && pos[u] + 24 <= n;  // symbols 1 .. nsym-1 from two 8-byte windows behind the first byte
                if (abl & 4) {  // (timing experiment: no text)
                    w0[u] = e[u] * 0x9E3779B97F4A7C15ull;
                    w1[u] = w0[u] >> 7;
                    continue;
                }
                w0[u] = windowed[u] ? *reinterpret_cast<const u64_unaligned*>(text + pos[u] + 1) : 0ull;
                w1[u] = windowed[u] && nsym > 9 ? *reinterpret_cast<const u64_unaligned*>(text + pos[u] + 9) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!live[u]) continue;
                uint64_t key = 0;
                if (windowed[u]) {
                    uint64_t w = w0[u];
                    for (int k = 1; k < nsym && k <= 8; ++k) {
                        key = key * kbase + ((uint64_t)k < rem[u] ? (uint64_t)s_map[w & 0xff] : 0ull);
                        w >>= 8;
                    }
                    w = w1[u];
                    for (int k = 9; k < nsym; ++k) {
                        key = key * kbase + ((uint64_t)k < rem[u] ? (uint64_t)s_map[w & 0xff] : 0ull);
                        w >>= 8;
                    }
                } else {
                    for (int k = 1; k < nsym; ++k)
                        key = key * kbase + ((uint64_t)k < rem[u] ? (uint64_t)s_map[text[pos[u] + k]] : 0ull);
                }
                const uint64_t i = it.begin + r0 + 256u * u;
                k32[i - gstart] = (uint32_t)(key >> low_bits);
                low[i - gstart] = (W)((key & ((1ull << low_bits) - 1ull)) | ((e[u] >> 32) << low_bits));
                if (!(abl & 8)) elo[i - gstart] = (uint32_t)e[u];
                if (!(abl & 1))
                    for (int q = 0; q < npass; ++q) atomicAdd(&s_hist[q][(uint32_t)(key >> (8 * q)) & 0xFFu], 1u);
            }
        }
        __syncthreads();
        unsigned long long* h = hist + (size_t)(it.bucket - bucket0) * 8 * 256;
        for (int p = 0; p < npass; ++p)
            if (s_hist[p][threadIdx.x]) atomicAdd(&h[p * 256 + threadIdx.x], (unsigned long long)s_hist[p][threadIdx.x]);
    }
}

template <typename W>
__global__ __launch_bounds__(256) void sa_assemble_entries_kernel(const uint32_t* __restrict__ elo, const W* __restrict__ low,
                                                                  int hi_shift, uint64_t cnt, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < cnt) out[i] = ((uint64_t)((uint32_t)low[i] >> hi_shift) << 32) | (uint64_t)elo[i];
}

struct FlagIn {
    const uint8_t* flags;
    static __device__ __forceinline__ U2 decode(uint32_t f) {
        const uint64_t u = (f >> 1) & 1u;
        return U2{u, u & (uint64_t)(f & 1u)};
    }
    __device__ __forceinline__ U2 operator()(uint64_t i) const { return decode(flags[i]); }
    // a thread's 16 consecutive flag bytes in one load (tiles start at multiples of 16; the flag array
    // is a library allocation, so it is 16-byte aligned)
    __device__ __forceinline__ void load8(uint64_t base, uint64_t n, const U2& identity, U2 (&v)[SC_IPT]) const {
        static_assert(SC_IPT == 16, "flag loader assumes 16 items per thread");
        if (base + 16 <= n) {
            const uint4 w = *reinterpret_cast<const uint4*>(flags + base);
            const uint32_t x[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = decode((x[k >> 2] >> (8 * (k & 3))) & 0xFFu);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = base + k < n ? decode(flags[base + k]) : identity;
        }
    }
};

// compaction of the unresolved entries + construction of their sort keys
// compaction of the unresolved entries: slot number, entry and group id of every unresolved entry
// (the sort key's minor part is filled in by sa_round_keys_kernel, one dense thread per entry)
// (SAW = how the suffix array is stored: SaRW<V> or the packed Sa40RW, index_impl.h)
template <typename SAW, typename I>
struct CompactOut {
    using V = typename SAW::val;
    SAW sa;
    I* U;
    uint64_t* skey;
    V* sval;
    int kbits;
    __device__ __forceinline__ void operator()(uint64_t i, const U2& ex, const U2& in) const {
        if (in.a == ex.a) return;  // not an unresolved entry
        const uint64_t j = ex.a;
        U[j] = (I)i;
        skey[j] = (in.b - 1) << kbits;  // group id
        sval[j] = sa.load(i);
    }
};

// The same compaction from the PREVIOUS round's list (text-extension rounds behind the first): only entries of that list can still be
// open, and sa_update left their new flag bytes in list order (`lf`), so the wave-autonomous flag sweeps run over m bytes instead of
// the array's n, and the entries come from the sorted list instead of a gather through the suffix array.
template <typename V, typename I>
struct CompactListOut {
    const I* Uo;
    const V* svo;
    I* U;
    uint64_t* skey;
    V* sval;
    int kbits;
    __device__ __forceinline__ void operator()(uint64_t j, const U2& ex, const U2& in) const {
        if (in.a == ex.a) return;  // not an unresolved entry
        const uint64_t jj = ex.a;
        U[jj] = Uo[j];
        skey[jj] = (in.b - 1) << kbits;  // group id
        sval[jj] = svo[j];
    }
};

// minor sort key of every compacted entry: the rank of the suffix h symbols further on (prefix
// doubling) or the next nsym2 symbols read from the text (text extension)
template <typename V, typename R, bool USE_ISA>
__global__ __launch_bounds__(256) void sa_round_keys_kernel(const V* __restrict__ sval, uint64_t m,
                                                            const uint64_t* __restrict__ doc_start,
                                                            const uint8_t* __restrict__ text,
                                                            const uint16_t* __restrict__ symmap,
                                                            const R* __restrict__ rank, int bits, uint64_t mask, uint64_t h,
                                                            int nsym2, int symbits, uint64_t* __restrict__ skey, uint64_t n_text,
                                                            const uint8_t* __restrict__ vl_len = nullptr, uint32_t vl_kb1 = 0,
                                                            uint32_t h_extra = 0, uint8_t* __restrict__ hcov = nullptr) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const V v = sval[j];
    const uint64_t d = (uint64_t)v & mask, off = (uint64_t)v >> bits;
    const uint64_t ds = doc_start[d];
    if (!USE_ISA && vl_len) {
        // variable-length keys (vl_code.h): the symbols the initial key covered IN FULL — the bucket symbol + the code words that fit
        // its vl_kb1 bits — are equal throughout the entry's group (equal keys decode alike), so the round may start behind them
        const uint64_t len = doc_start[d + 1] - ds - off;
        const uint8_t* q = text + ds + off;
        // (the first symbols by aligned 8-byte loads, like the keys below: byte loads in suffix order cost a line each)
        const uint64_t pa = (uint64_t)q & ~7ull;
        const uint32_t sh = (uint32_t)((uint64_t)q & 7ull) * 8u;
        const uint64_t tend = (uint64_t)(text + n_text);
        const uint64_t w0 = *reinterpret_cast<const uint64_t*>(pa);
        const uint64_t w1 = pa + 8 < tend ? *reinterpret_cast<const uint64_t*>(pa + 8) : 0ull;
        const uint64_t w2 = pa + 16 < tend ? *reinterpret_cast<const uint64_t*>(pa + 16) : 0ull;
        const uint64_t b0 = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;  // bytes q[0] .. q[7]
        const uint64_t b1 = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;  // bytes q[8] .. q[15]
        uint32_t used = 0, cov = 1;
        while ((uint64_t)cov < len && cov < 200u) {
            const uint32_t byte = cov < 8u ? (uint32_t)(b0 >> (8u * cov)) & 0xFFu : (cov < 16u ? (uint32_t)(b1 >> (8u * (cov - 8u))) & 0xFFu : (uint32_t)q[cov]);
            used += vl_len[byte];
            if (used > vl_kb1) break;
            ++cov;
        }
        hcov[j] = (uint8_t)cov;
        h = (uint64_t)cov + h_extra;
    }
    uint64_t key2 = 0;
    if constexpr (USE_ISA) {
        key2 = (uint64_t)rank[ds + d + off + h];  // extended position: one end slot per document
    } else {
        const uint64_t rem = doc_start[d + 1] - ds - off - h;  // >= 0: unresolved => length >= h
        const uint8_t* p = text + ds + off + h;
        if (nsym2 <= 9 && rem > 0) {
            // the next symbols by TWO aligned 8-byte loads instead of one load per byte: the entries arrive in suffix order, every
            // load instruction of a wave touches 64 different lines, and with thousands of gathers in flight per CU a line does not
            // survive in the L1 from one byte to the next (16 GiB shard: 310 B of fabric traffic per entry with byte loads).  The
            // second word is read only where it holds a byte of the text (an aligned word never crosses a page)
            const uint64_t pa = (uint64_t)p & ~7ull;
            const uint32_t sh = (uint32_t)((uint64_t)p & 7ull) * 8u;
            const uint64_t w0 = *reinterpret_cast<const uint64_t*>(pa);
            const uint64_t w1 = pa + 8 < (uint64_t)(text + n_text) ? *reinterpret_cast<const uint64_t*>(pa + 8) : 0ull;
            const uint64_t lo = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;  // bytes p[0] .. p[7]
            const uint32_t b8 = (uint32_t)(w1 >> sh) & 0xFFu;                 // byte p[8]
            for (int k = 0; k < nsym2; ++k) {
                const uint32_t byte = k < 8 ? (uint32_t)(lo >> (8 * k)) & 0xFFu : b8;
                const uint64_t sym = (uint64_t)k < rem ? (uint64_t)symmap[byte] : 0ull;
                key2 = (key2 << symbits) | sym;
            }
        } else {
            for (int k = 0; k < nsym2; ++k) {
                const uint64_t sym = (uint64_t)k < rem ? (uint64_t)symmap[p[k]] : 0ull;
                key2 = (key2 << symbits) | sym;
            }
        }
    }
    skey[j] |= key2;
}

// Compaction of the unresolved entries, specialised for the byte-flag array (the generic scan spends most of its time carrying
// 16 (u64, u64) pairs per thread for entries that are almost all resolved).  Round 4: WAVE-autonomous — a wavefront owns whole
// tiles of the scan geometry (SC_TILE = 4096 flags = four 1 KiB chunks of 16 bytes per lane), counts with SWAR popcounts, scans
// with wave shuffles and takes its tiles one after the other: no LDS, no workgroup barrier, millions of 4 KiB workgroups less
// (the compaction itself stays bound by its gather of the unresolved entries).  sa_flag_count_kernel leaves the raw tile sums (unresolved entries, unresolved group
// heads) in the partials of scan.h's layout; sa_flag_compact_kernel uses their exclusive scan.
constexpr uint32_t FC_TILES_PER_WAVE = 4;
__device__ __forceinline__ void fc_load(const uint8_t* __restrict__ flags, uint64_t n, uint64_t base, uint32_t (&x)[4]) {
    x[0] = x[1] = x[2] = x[3] = 0;
    if (base + 16 <= n) {
        const uint4 w = *reinterpret_cast<const uint4*>(flags + base);
        x[0] = w.x; x[1] = w.y; x[2] = w.z; x[3] = w.w;
    } else if (base < n) {
        for (int k = 0; k < 16 && base + k < n; ++k) x[k >> 2] |= (uint32_t)flags[base + k] << (8 * (k & 3));
    }
}
__device__ __forceinline__ uint32_t fc_count(const uint32_t (&x)[4]) {  // unresolved entries | unresolved group heads << 16
    uint32_t cu = 0, ch = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t u = (x[q] >> 1) & 0x01010101u;
        cu += __popc(u);
        ch += __popc(u & x[q] & 0x01010101u);
    }
    return cu | (ch << 16);
}
__global__ __launch_bounds__(256) void sa_flag_count_kernel(const uint8_t* __restrict__ flags, uint64_t n, uint64_t tiles, U2* __restrict__ partials) {
    static_assert(SC_TILE == 4096, "flag compaction assumes 4096-flag tiles");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t t0 = ((uint64_t)blockIdx.x * 4 + wave) * FC_TILES_PER_WAVE;
#pragma unroll
    for (uint32_t k = 0; k < FC_TILES_PER_WAVE; ++k) {
        const uint64_t tile = t0 + k;
        if (tile >= tiles) break;  // (uniform per wavefront)
        uint32_t v = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t x[4];
            fc_load(flags, n, tile * SC_TILE + (uint64_t)c * 1024 + (uint64_t)lane * 16, x);
            v += fc_count(x);  // (both halves stay below 2^16: at most 4096 per tile)
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) partials[tile] = U2{(uint64_t)(v & 0xFFFFu), (uint64_t)(v >> 16)};
    }
}
// Round 6: the flagged entries of a tile are first LISTED (12-bit offset + head bit, in order, in a wave-private stretch of the
// LDS) and then handed to `out` DENSELY, 64 per wave instruction.  The round-4 form called `out` — a gather through the suffix
// array and three stores — from a 16-trip loop over the lane's bytes: with 2.4 % of the flags set (8 GiB of Zipf text) that were 64
// sparse trips per tile, ~450 memory instructions with one or two lanes alive each, and the sweep took 10.4 ms where the bare
// count over the same flags takes 1.5 ms.
template <typename Out>
__global__ __launch_bounds__(256) void sa_flag_compact_kernel(const uint8_t* __restrict__ flags, uint64_t n, uint64_t tiles,
                                                              const U2* __restrict__ partials, Out out) {
    static_assert(SC_TILE == 4096, "flag compaction assumes 4096-flag tiles");
    __shared__ uint16_t s_list[4][SC_TILE];  // per wave: the tile's unresolved positions (offset | head << 12), worst case all of them
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint16_t* const list = s_list[wave];
    const uint64_t t0 = ((uint64_t)blockIdx.x * 4 + wave) * FC_TILES_PER_WAVE;
    for (uint32_t k = 0; k < FC_TILES_PER_WAVE; ++k) {
        const uint64_t tile = t0 + k;
        if (tile >= tiles) break;  // (uniform per wavefront)
        uint32_t x[4][4], v[4], incl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) fc_load(flags, n, tile * SC_TILE + (uint64_t)c * 1024 + (uint64_t)lane * 16, x[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) incl[c] = v[c] = fc_count(x[c]);
        // element order inside the tile = (chunk, lane, byte): one wave scan per chunk (independent: their shuffles interleave)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t y = __shfl_up(incl[c], off);
                if (lane >= off) incl[c] += y;
            }
        }
        uint32_t any = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) any |= v[c];
        if (__builtin_amdgcn_ballot_w64(any != 0) == 0) continue;  // (nothing unresolved in the tile)
        const U2 tile_run = partials[tile];
        // ---- phase 1: list the unresolved positions in order (a lane walks the SET bits of its 16 bytes only)
        uint32_t cpre = 0;  // packed counts of the chunks in front
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t ctot = __shfl(incl[c], 63);
            if (v[c] & 0xFFFFu) {
                uint32_t slot = (cpre + incl[c] - v[c]) & 0xFFFFu;
                uint32_t um = 0, hm = 0;  // bit q: byte q is unresolved / an unresolved group head
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t u = (x[c][w] >> 1) & 0x01010101u;
                    um |= (((u * 0x01020408u) >> 24) & 0xFu) << (4 * w);
                    hm |= ((((u & x[c][w]) * 0x01020408u) >> 24) & 0xFu) << (4 * w);
                }
                const uint32_t off0 = (uint32_t)c * 1024u + (uint32_t)lane * 16u;
                while (um) {
                    const uint32_t q = (uint32_t)__builtin_ctz(um);
                    um &= um - 1u;
                    list[slot++] = (uint16_t)((off0 + q) | (((hm >> q) & 1u) << 12));
                }
            }
            cpre += ctot;
        }
        const uint32_t total = cpre & 0xFFFFu;
        // (the list is the wave's own: its LDS operations execute in program order — the fences keep the compiler from moving them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase 2: one unresolved entry per lane
        uint32_t heads = 0;  // unresolved group heads in front of this trip
        for (uint32_t b0 = 0; b0 < total; b0 += 64u) {
            const uint32_t t = b0 + (uint32_t)lane;
            const bool live = t < total;
            const uint32_t e = live ? (uint32_t)list[t] : 0u;
            const uint32_t head = (e >> 12) & 1u;
            const uint64_t hb = __builtin_amdgcn_ballot_w64(head != 0);
            if (live) {
                const U2 run{tile_run.a + t, tile_run.b + heads + (uint64_t)__popcll(hb & lt_mask)};
                const U2 nxt{run.a + 1, run.b + head};
                out(tile * SC_TILE + (uint64_t)(e & 0xFFFu), run, nxt);
            }
            heads += (uint32_t)__popcll(hb);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the next tile's list overwrites this one's)
        __builtin_amdgcn_wave_barrier();
    }
}

// The refinement's sort, specialised for what it sorts: the compacted list is ALREADY ordered by group (the group id sits in the
// key's high bits and ascends with the list), so "sort by (group, minor key)" only permutes entries INSIDE their group — runs of two
// or three entries on every named configuration (8 GiB of Zipf text: 28 242 of 2.07 x 10^8 entries sit in groups of more than 49,
// 6 in groups of more than 257).  One pass instead of the general sort's eight: an entry looks at the members of its group on both
// sides (neighbouring list slots: cached lines) and takes the slot
//     j - #{members in front with a greater key} + #{members behind with a smaller key}
// — the stable order, slot for slot what the LSD passes produce.  The work is the sum of the SQUARED group sizes, so it is bounded:
// an entry that walks more than GS_FREE members reports its walk to a global counter, and once the walks add up to more than
// `budget` members (16 per entry of the list), or a group exceeds `cap` members on one side of an entry, `state[0]` is raised — every
// thread gives up at its next look at it and the caller falls back to the general sort (duplicated documents: groups of thousands).
constexpr uint32_t GS_FREE = 32;  // (walks this short cost no more than the general sort would: never reported)
template <typename V>
__global__ __launch_bounds__(256) void sa_group_sort_kernel(const uint64_t* __restrict__ kin, const V* __restrict__ vin, uint64_t m, int kbits, uint32_t cap,
                                                            unsigned long long budget, uint64_t* __restrict__ kout, V* __restrict__ vout,
                                                            unsigned long long* __restrict__ state /* [0] gave up, [1] members walked by long walks */) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const volatile unsigned long long* vstate = state;
    if (vstate[0]) return;
    const uint64_t k = kin[j];
    const uint64_t g = kbits >= 64 ? 0ull : k >> kbits;
    auto gid = [&](uint64_t x) { return kbits >= 64 ? 0ull : x >> kbits; };
    uint64_t pos = j;
    uint32_t walked = 0;
    bool too = false;
    {
        uint32_t d = 1;
        for (; d <= cap && j >= d; ++d) {
            const uint64_t o = kin[j - d];
            if (gid(o) != g) break;
            pos -= (uint64_t)(o > k);
            if ((d & 127u) == 0 && vstate[0]) return;
        }
        too = d > cap && j >= d && gid(kin[j - d]) == g;
        walked += d - 1;
    }
    if (!too) {
        uint32_t d = 1;
        for (; d <= cap && j + d < m; ++d) {
            const uint64_t o = kin[j + d];
            if (gid(o) != g) break;
            pos += (uint64_t)(o < k);
            if ((d & 127u) == 0 && vstate[0]) return;
        }
        too = d > cap && j + d < m && gid(kin[j + d]) == g;
        walked += d - 1;
    }
    if (walked > GS_FREE && !too) too = atomicAdd(&state[1], (unsigned long long)walked) + walked > budget;
    if (too) {
        atomicExch(&state[0], 1ull);
        return;
    }
    kout[pos] = k;
    vout[pos] = vin[j];
}

template <typename I>
__global__ __launch_bounds__(256) void sa_newhead_kernel(const uint64_t* __restrict__ skey, const I* __restrict__ U,
                                                         const uint8_t* __restrict__ flags, uint64_t m,
                                                         uint8_t* __restrict__ nh) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const bool oldhead = flags[U[j]] & 1;
    nh[j] = (uint8_t)(oldhead || j == 0 || skey[j] != skey[j - 1]);
}

template <typename SAW, typename I, bool WANT_POS = true>
__device__ __forceinline__ void sa_place(uint64_t j, uint64_t m, const typename SAW::val* sval, const I* U, const uint8_t* nh,
                                         const uint64_t* doc_start, int bits, uint64_t mask, uint64_t hnew, SAW sa,
                                         uint8_t* flags, uint64_t& ext_pos, unsigned long long* still_open,
                                         const uint8_t* hcov = nullptr, uint32_t h_add = 0, uint8_t* lf = nullptr) {
    if (hcov) hnew = (uint64_t)hcov[j] + h_add;  // (variable-length keys: the group's own depth; slot j stays inside its group through the sort)
    const typename SAW::val v = sval[j];
    const uint64_t i = U[j];
    const bool head = nh[j];
    const bool last = j + 1 == m || nh[j + 1];
    const uint64_t d = (uint64_t)v & mask, off = (uint64_t)v >> bits;
    // (a group of one is settled whatever is left of its document: most entries of a round, and their two random loads from the
    //  document table are the kernel's only gathers — skipped unless the caller wants the entry's text position)
    uint64_t ds = 0;
    bool exhausted = false;
    if (WANT_POS || !(head && last)) {
        ds = doc_start[d];
        exhausted = doc_start[d + 1] - ds - off < hnew;
    }
    const bool open = !(head && last) && !exhausted;
    sa.store(i, v);
    flags[i] = (uint8_t)((head ? 1 : 0) | (open ? 2 : 0));
    if (lf) lf[j] = (uint8_t)((head ? 1 : 0) | (open ? 2 : 0));  // (the same byte in list order: the next round compacts from the list)
    if (open) atomicAdd(still_open, 1ull);  // the compiler folds this into one add per wave
    ext_pos = ds + d + off;
}

template <typename SAW, typename I>
__global__ __launch_bounds__(256) void sa_update_kernel(const typename SAW::val* __restrict__ sval, const I* __restrict__ U,
                                                        const uint8_t* __restrict__ nh, uint64_t m,
                                                        const uint64_t* __restrict__ doc_start, int bits,
                                                        uint64_t mask, uint64_t hnew, SAW sa,
                                                        uint8_t* __restrict__ flags,
                                                        unsigned long long* __restrict__ still_open,
                                                        const uint8_t* __restrict__ hcov = nullptr, uint32_t h_add = 0, uint8_t* __restrict__ lf = nullptr) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    uint64_t q;
    sa_place<SAW, I, false>(j, m, sval, U, nh, doc_start, bits, mask, hnew, sa, flags, q, still_open, hcov, h_add, lf);
}

template <typename I>
struct HeadIn {  // group start (+1) of compacted entry j, 0 for non-heads  -> max-scan
    const uint8_t* nh;
    const I* U;
    __device__ __forceinline__ uint64_t operator()(uint64_t j) const { return nh[j] ? (uint64_t)U[j] + 1 : 0ull; }
};
// Prefix doubling compares RANKS, and a rank is a slot number of the array: it orders suffixes correctly only while the
// array's slot order is the plain unsigned order.  When the bucket-wise build lays blocks out in the reference's order
// (first-symbol buckets by signed byte, the two byte blocks of radix-node buckets swapped) a rank is taken through the
// inverse of that block permutation first: slot -> where the slot would sit in unsigned order.  nseg = 0: identity.
struct SlotOrder {
    const unsigned long long* a_start = nullptr;  // [nseg] block starts in the array, ascending
    const unsigned long long* u_start = nullptr;  // [nseg] where each block starts in unsigned order
    uint32_t nseg = 0;
    __device__ __forceinline__ uint64_t operator()(uint64_t slot) const {
        if (nseg == 0) return slot;
        uint32_t lo = 0, hi = nseg - 1;  // last block with a_start <= slot
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo + 1) / 2;
            if (a_start[mid] <= slot) lo = mid; else hi = mid - 1;
        }
        return slot - a_start[lo] + u_start[lo];
    }
};

template <typename SAW, typename I, typename R>
struct UpdateOut {
    using V = typename SAW::val;
    const V* sval;
    const I* U;
    const uint8_t* nh;
    uint64_t m;
    const uint64_t* doc_start;
    int bits;
    uint64_t mask, hnew;
    SAW sa;
    uint8_t* flags;
    R* rank;
    unsigned long long* still_open;
    SlotOrder order;
    __device__ __forceinline__ void operator()(uint64_t j, uint64_t, uint64_t incl) const {
        uint64_t q;
        sa_place<SAW, I>(j, m, sval, U, nh, doc_start, bits, mask, hnew, sa, flags, q, still_open);
        rank[q] = (R)(order(incl - 1) + 1);  // (incl = the group's first slot + 1)
    }
};

// inverse array from the current grouping: rank[ext(sa[i])] = group start + 1
struct AllHeadIn {
    const uint8_t* flags;
    __device__ __forceinline__ uint64_t operator()(uint64_t i) const { return (flags[i] & 1) ? i + 1 : 0ull; }
};
template <typename SAW, typename R>
struct IsaOut {
    SAW sa;
    const uint64_t* doc_start;
    int bits;
    uint64_t mask;
    R* rank;
    SlotOrder order;
    __device__ __forceinline__ void operator()(uint64_t i, uint64_t, uint64_t incl) const {
        const auto v = sa.load(i);
        const uint64_t d = (uint64_t)v & mask, off = (uint64_t)v >> bits;
        rank[doc_start[d] + d + off] = (R)(order(incl - 1) + 1);
    }
};

// ---------------------------------------------------------------------------------------------
// reference-compatible ordering for text with bytes >= 0x80
// ---------------------------------------------------------------------------------------------
// The reference buckets radix nodes by `(int)char - CHAR_MIN + 1` with a SIGNED char (index.h:72), so
// while a bucket holds more than chuck_size suffixes its children are laid out
//     [end of document][bytes 0x80..0xFF][bytes 0x00..0x7F]
// whereas leaves (<= chuck_size, index.cpp:86-95) and both binary searches compare unsigned.  Given the
// plain unsigned order built above, the reference's order is reached by rotating the two byte blocks
// of every "big" bucket, level by level down the trie.  compat_bounds_kernel finds, for one big bucket
// at depth `depth`, where each of the 257 unsigned symbols starts (the bucket is sorted by that symbol).
struct CompatBucket {
    unsigned long long lo, hi;
};

template <typename V>
__global__ __launch_bounds__(320) void compat_bounds_kernel(typename SaOf<V>::ptr sa,
                                                            const uint8_t* __restrict__ text,
                                                            const uint64_t* __restrict__ doc_start, int bits,
                                                            uint64_t mask, const CompatBucket* __restrict__ buckets,
                                                            uint64_t depth, unsigned long long* __restrict__ bounds) {
    const int c = threadIdx.x;  // symbol 0 = end of document, 1..256 = byte + 1; 257 = one past
    if (c > 257) return;
    const CompatBucket b = buckets[blockIdx.x];
    uint64_t lo = b.lo, hi = b.hi;
    if (c == 257) {
        bounds[(uint64_t)blockIdx.x * 258 + c] = b.hi;
        return;
    }
    while (lo < hi) {  // first slot whose symbol at `depth` is >= c
        const uint64_t mid = lo + (hi - lo) / 2;
        const auto e = sa[mid];
        const uint64_t d = (uint64_t)e & mask, off = (uint64_t)e >> bits;
        const uint64_t p = doc_start[d] + off + depth;
        const uint32_t sym = p == doc_start[d + 1] ? 0u : (uint32_t)text[p] + 1u;
        if (sym < (uint32_t)c) lo = mid + 1; else hi = mid;
    }
    bounds[(uint64_t)blockIdx.x * 258 + c] = lo;
}

// in-place reversal of x[0, len): [A|B] -> [B|A] is reverse(A), reverse(B), reverse(A|B) — the same traffic as
// a round trip through a scratch copy, but without the scratch (up to n entries at the root bucket)
template <typename V>
__global__ __launch_bounds__(256) void compat_reverse_kernel(V* __restrict__ x, uint64_t len) {
    // grid-stride: the root bucket of a 16 GiB shard has more than 2^32 pairs, which one launch cannot address
    // with a thread each (found by tests/test_gpu_fullsize.py: the last of the three reversals silently did not run)
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < len / 2; i += stride) {
        const V a = x[i], b = x[len - 1 - i];
        x[i] = b;
        x[len - 1 - i] = a;
    }
}

// roots (optional): the bucket-wise build already laid its first-symbol buckets out in the reference's root order —
// the walk starts one level down, with those buckets as nodes
// (V = storage tag: uint32_t / uint64_t entries in sa_buf, or Packed40 — low words in sa_buf, high bytes in *sa_hi)
template <typename V>
void apply_reference_order(Index& ix, DevBuf& sa_buf, DevBuf* sa_hi, const std::vector<CompatBucket>* roots = nullptr) {
    constexpr bool PK = std::is_same<V, Packed40>::value;
    using EV = typename std::conditional<PK, uint32_t, typename SaOf<V>::val>::type;  // element type of sa_buf
    EV* sa = sa_buf.as<EV>();
    typename SaOf<V>::ptr sa_r;
    if constexpr (PK) sa_r = Sa40{sa_buf.as<uint32_t>(), sa_hi->as<uint8_t>()};
    else sa_r = sa_buf.as<EV>();
    hipStream_t s = ix.stream;
    const uint64_t n = ix.size;
    const uint64_t chuck = std::max<uint64_t>(4096, n / 256);  // index.cpp:218
    std::vector<CompatBucket> level{{0ull, (unsigned long long)n}};
    if (n <= chuck) return;
    DevBuf d_buckets, d_bounds;
    uint64_t depth = 0;
    if (roots) {
        level.clear();
        for (const CompatBucket& b : *roots)
            if (b.hi - b.lo > chuck) level.push_back(b);
        depth = 1;
    }
    uint64_t moved = 0;  // elements reversed (each one read and written)
    const int tprof = ix.prof.begin(s);
    auto reverse = [&](uint64_t at, uint64_t len) {
        if (len > 1) {
            hipLaunchKernelGGL((compat_reverse_kernel<EV>), dim3((unsigned)std::min<uint64_t>(ceil_div(len / 2, 256), 1u << 20)), dim3(256),
                               0, s, sa + at, len);
            if constexpr (PK)
                hipLaunchKernelGGL((compat_reverse_kernel<uint8_t>), dim3((unsigned)std::min<uint64_t>(ceil_div(len / 2, 256), 1u << 20)),
                                   dim3(256), 0, s, sa_hi->as<uint8_t>() + at, len);
            moved += len;
        }
    };
    while (!level.empty()) {
        const size_t nb = level.size();
        d_buckets.ensure(nb * sizeof(CompatBucket));
        d_bounds.ensure(nb * 258 * sizeof(uint64_t));
        CDB_HIP(hipMemcpyAsync(d_buckets.p, level.data(), nb * sizeof(CompatBucket), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL((compat_bounds_kernel<V>), dim3((unsigned)nb), dim3(320), 0, s, sa_r, ix.d_text,
                           (const uint64_t*)ix.d_doc_start.as<uint64_t>(), (int)ix.bits, ix.mask,
                           (const CompatBucket*)d_buckets.as<CompatBucket>(), depth,
                           d_bounds.as<unsigned long long>());
        std::vector<unsigned long long> hb(nb * 258);
        CDB_HIP(hipMemcpyAsync(hb.data(), d_bounds.p, hb.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        std::vector<CompatBucket> next;
        for (size_t b = 0; b < nb; ++b) {
            const unsigned long long* bd = &hb[b * 258];
            const uint64_t a0 = bd[1], b0 = bd[129], end = bd[257];  // [a0,b0) = 0x00..0x7F, [b0,end) = 0x80..0xFF
            const uint64_t lenA = b0 - a0, lenB = end - b0;
            if (lenA && lenB) {
                reverse(a0, lenA);
                reverse(b0, lenB);
                reverse(a0, lenA + lenB);
                ix.bstats.compat_rotations++;
            }
            for (int v = 0; v < 256; ++v) {
                const uint64_t len = bd[v + 2] - bd[v + 1];
                if (len <= chuck) continue;
                const uint64_t start = v >= 128 ? a0 + (bd[v + 1] - b0) : a0 + lenB + (bd[v + 1] - a0);
                next.push_back(CompatBucket{(unsigned long long)start, (unsigned long long)(start + len)});
            }
        }
        level.swap(next);
        ++depth;
    }
    ix.bstats.compat_depth = depth;
    ix.prof.end(tprof, "sa_compat_rotate", 2 * moved * (PK ? 5 : sizeof(EV)), s);
    CDB_HIP(hipStreamSynchronize(s));
}

// The same order reached in ONE out-of-place pass (when a second array fits): the rotations only move whole
// child buckets, so every bucket of the rotated array is still a contiguous range of the plainly sorted one.
// The bucket tree is therefore walked on the UNROTATED array — each bucket carries (its range there, where it
// ends up) — and the leaves become copy segments.  1 read + 1 write per entry instead of 2 + 2 for every
// level of three reversals; the kept search keys can make the same trip.
struct CompatSeg {
    unsigned long long src, dst;
    uint32_t count, pad;
};
constexpr uint32_t CS_ITEM = 16384;
template <typename T>
__global__ __launch_bounds__(256) void compat_segcopy_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                             const CompatSeg* __restrict__ segs) {
    const CompatSeg sg = segs[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < sg.count; i += 256) out[sg.dst + i] = in[sg.src + i];
}

struct CompatNode {
    unsigned long long lo, hi, fin;  // range in the sorted array, first slot in the reference order
};

// returns false (nothing done) when the scratch array cannot be had; sa_buf is replaced by the reordered array
template <typename V>
bool apply_reference_order_oop(Index& ix, DevBuf& sa_buf, DevBuf* sa_hi, const std::vector<CompatBucket>* roots = nullptr) {
    constexpr bool PK = std::is_same<V, Packed40>::value;
    using EV = typename std::conditional<PK, uint32_t, typename SaOf<V>::val>::type;  // element type of sa_buf
    constexpr size_t ESZ = PK ? 5 : sizeof(EV);
    hipStream_t s = ix.stream;
    const uint64_t n = ix.size;
    const uint64_t chuck = std::max<uint64_t>(4096, n / 256);  // index.cpp:218
    if (n <= chuck) return true;
    {   // only when the second array fits comfortably: a failed allocation would flush the block cache for nothing
        size_t fre = 0, tot = 0;
        CDB_HIP(hipMemGetInfo(&fre, &tot));
        if ((double)n * ESZ > 0.8 * ((double)fre + (double)DevPool::get().cached_bytes())) return false;
        // ... and only when the pool can hand it out as it stands (a cached block of that size, or untouched VRAM): a hipMalloc
        // that fails first makes the pool return its whole cache to the driver — 2.5-5 s per 16 GiB build when it happened
        if (!DevPool::get().can_serve(n * sizeof(EV), ix.device)) return false;
    }
    DevBuf dst, dst_hi;
    try {
        dst.alloc(n * sizeof(EV));
        if (PK) dst_hi.alloc(n);
    } catch (const Error&) {
        return false;
    }
    typename SaOf<V>::ptr sa;
    if constexpr (PK) sa = Sa40{sa_buf.as<uint32_t>(), sa_hi->as<uint8_t>()};
    else sa = sa_buf.as<EV>();
    std::vector<CompatNode> level{{0ull, (unsigned long long)n, 0ull}};
    std::vector<CompatSeg> segs;
    auto emit = [&](uint64_t src, uint64_t dstpos, uint64_t len) {
        if (!len) return;
        for (uint64_t o = 0; o < len; o += CS_ITEM)
            segs.push_back(CompatSeg{(unsigned long long)(src + o), (unsigned long long)(dstpos + o),
                                     (uint32_t)std::min<uint64_t>(CS_ITEM, len - o), 0u});
    };
    DevBuf d_buckets, d_bounds, d_segs;
    uint64_t depth = 0;
    if (roots) {  // first-symbol buckets already in the reference's root order: they stay where they are
        level.clear();
        for (const CompatBucket& b : *roots) {
            if (b.hi - b.lo > chuck) level.push_back(CompatNode{b.lo, b.hi, b.lo});
            else emit(b.lo, b.lo, b.hi - b.lo);
        }
        depth = 1;
    }
    const uint64_t rotations_before = ix.bstats.compat_rotations;
    std::vector<CompatBucket> cb;
    while (!level.empty()) {
        const size_t nb = level.size();
        cb.resize(nb);
        for (size_t b = 0; b < nb; ++b) cb[b] = CompatBucket{level[b].lo, level[b].hi};
        d_buckets.ensure(nb * sizeof(CompatBucket));
        d_bounds.ensure(nb * 258 * sizeof(uint64_t));
        CDB_HIP(hipMemcpyAsync(d_buckets.p, cb.data(), nb * sizeof(CompatBucket), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL((compat_bounds_kernel<V>), dim3((unsigned)nb), dim3(320), 0, s, sa, ix.d_text,
                           (const uint64_t*)ix.d_doc_start.as<uint64_t>(), (int)ix.bits, ix.mask,
                           (const CompatBucket*)d_buckets.as<CompatBucket>(), depth, d_bounds.as<unsigned long long>());
        std::vector<unsigned long long> hb(nb * 258);
        CDB_HIP(hipMemcpyAsync(hb.data(), d_bounds.p, hb.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        std::vector<CompatNode> next;
        for (size_t b = 0; b < nb; ++b) {
            const unsigned long long* bd = &hb[b * 258];
            const CompatNode nd = level[b];
            const uint64_t a0 = bd[1], b0 = bd[129], end = bd[257];  // [a0,b0) = 0x00..0x7F, [b0,end) = 0x80..0xFF
            const uint64_t lenEnd = a0 - nd.lo, lenA = b0 - a0, lenB = end - b0;
            if (lenA && lenB) ix.bstats.compat_rotations++;
            emit(nd.lo, nd.fin, lenEnd);  // suffixes that end here stay in front
            // children in reference order: 0x80..0xFF first, then 0x00..0x7F; runs of small children are copied
            // as one segment
            uint64_t run_src = 0, run_dst = 0, run_len = 0;
            auto flush = [&] {
                emit(run_src, run_dst, run_len);
                run_len = 0;
            };
            for (int k = 0; k < 256; ++k) {
                const int v = k < 128 ? 128 + k : k - 128;
                const uint64_t lo = bd[v + 1], len = bd[v + 2] - lo;
                if (!len) continue;
                const uint64_t fin = v >= 128 ? nd.fin + lenEnd + (lo - b0) : nd.fin + lenEnd + lenB + (lo - a0);
                if (len > chuck) {
                    flush();
                    next.push_back(CompatNode{(unsigned long long)lo, (unsigned long long)(lo + len), (unsigned long long)fin});
                } else if (run_len && run_src + run_len == lo && run_dst + run_len == fin) {
                    run_len += len;
                } else {
                    flush();
                    run_src = lo;
                    run_dst = fin;
                    run_len = len;
                }
            }
            flush();
        }
        level.swap(next);
        ++depth;
    }
    ix.bstats.compat_depth = depth;
    if (ix.bstats.compat_rotations == rotations_before) return true;  // every node had its children on one side: nothing moves
    if (!segs.empty()) {
        d_segs.alloc(segs.size() * sizeof(CompatSeg));
        CDB_HIP(hipMemcpyAsync(d_segs.p, segs.data(), segs.size() * sizeof(CompatSeg), hipMemcpyHostToDevice, s));
        int t = ix.prof.begin(s);
        hipLaunchKernelGGL((compat_segcopy_kernel<EV>), dim3((unsigned)segs.size()), dim3(256), 0, s, (const EV*)sa_buf.as<EV>(), dst.as<EV>(),
                           (const CompatSeg*)d_segs.as<CompatSeg>());
        if constexpr (PK)
            hipLaunchKernelGGL((compat_segcopy_kernel<uint8_t>), dim3((unsigned)segs.size()), dim3(256), 0, s, (const uint8_t*)sa_hi->as<uint8_t>(),
                               dst_hi.as<uint8_t>(), (const CompatSeg*)d_segs.as<CompatSeg>());
        ix.prof.end(t, "sa_compat_copy", 2 * n * ESZ, s);
    }
    // the kept search keys make the same trip, so that probes on the reordered array can still be decided from
    // one load (query.hip follows the reference's probe sequence there, with the same comparisons)
    if (!segs.empty() && ix.key_nsym) {
        auto permute = [&](DevBuf& buf, size_t esz) {
            if (!buf.p) return;
            DevBuf nb;
            nb.alloc(n * esz);
            const dim3 grid((unsigned)segs.size());
            const CompatSeg* sg = d_segs.as<CompatSeg>();
            if (esz == 8) hipLaunchKernelGGL((compat_segcopy_kernel<uint64_t>), grid, dim3(256), 0, s, (const uint64_t*)buf.as<uint64_t>(), nb.as<uint64_t>(), sg);
            else if (esz == 4) hipLaunchKernelGGL((compat_segcopy_kernel<uint32_t>), grid, dim3(256), 0, s, (const uint32_t*)buf.as<uint32_t>(), nb.as<uint32_t>(), sg);
            else if (esz == 2) hipLaunchKernelGGL((compat_segcopy_kernel<uint16_t>), grid, dim3(256), 0, s, (const uint16_t*)buf.as<uint16_t>(), nb.as<uint16_t>(), sg);
            else hipLaunchKernelGGL((compat_segcopy_kernel<uint8_t>), grid, dim3(256), 0, s, (const uint8_t*)buf.as<uint8_t>(), nb.as<uint8_t>(), sg);
            CDB_HIP(hipStreamSynchronize(s));
            buf = std::move(nb);
        };
        permute(ix.d_keys, 8);
        permute(ix.d_keys32, 4);
        permute(ix.d_keylow, (size_t)std::max(ix.key_low_bytes, 1));
    }
    CDB_HIP(hipStreamSynchronize(s));
    sa_buf = std::move(dst);
    if constexpr (PK) *sa_hi = std::move(dst_hi);
    return true;
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// I = type of suffix-array slot numbers, R = type of ranks / extended text positions: u32 while
// n + D < 2^32, u64 beyond.  `big` additionally selects the bucket-wise initial sort that keeps the
// working set inside HBM for multi-GiB corpora with 8-byte entries.
template <typename V, typename I, typename R>
void build_typed(Index& ix, bool big) {
    hipStream_t s = ix.stream;
    const uint64_t n = ix.size, D = ix.ndocs;
    BuildStats& st = ix.bstats;
    st = BuildStats{};
    {
        // the previous suffix array and kept keys go back to the block cache first: a rebuild of the same
        // corpus shape then finds every buffer it needs there instead of asking the driver for fresh memory
        const double tf = now_ms();
        ix.release_sa();
        ix.drop_keys();
        ix.d_pivots.release();
        ix.pivot_levels = 0;
        st.free_ms += now_ms() - tf;
    }
    if (n == 0) {
        ix.d_sa.alloc(16);
        ix.sa_sorted = true;
        ix.pivot_levels = 0;
        ix.drop_keys();
        return;
    }
    if (sizeof(R) == 4 && n + D + 2 >= (1ull << 32)) throw Error("internal: 32-bit ranks selected for a corpus >= 2^32");
    const uint8_t* text = ix.d_text;
    const uint64_t* doc_start = ix.d_doc_start.as<uint64_t>();

    // ---- 1. alphabet -> order-preserving dense codes
    DevBuf d_counts, d_symmap;
    d_counts.alloc(256 * sizeof(uint64_t));
    d_symmap.alloc(256 * sizeof(uint16_t));
    CDB_HIP(hipMemsetAsync(d_counts.p, 0, 256 * sizeof(uint64_t), s));
    // bucket-wise builds with look-back-free generated passes count the bytes per 8 Ki-position tile (the rows become the tile
    // bases of the records / partition pass, radix_sort.h: TextGen::tile_base); their column sums are the histogram
    DevBuf d_tbc;
    const uint32_t tiles8 = (uint32_t)ceil_div(n, (uint64_t)RS_GEN8_TILE);
    const bool tile_bytes = big && sizeof(V) == 8 && ix.gen_prebased && ix.segmented_sort && rs_atomic_rank_ok(s);
    const unsigned long long* d_count_src = d_counts.as<unsigned long long>();
    DevBuf d_pair_hi;      // per byte value: positions whose next byte is >= 0x80 (counted beside the bytes, PAIR form)
    bool pair_hi_counted = false;
    if (tile_bytes) {
        d_tbc.alloc((size_t)tiles8 * 256 * sizeof(uint32_t));
        if (ix.reference_compat && ix.fold_root && ix.fold_depth1 && ix.fuse_pairclass) {
            // the reference-order build will want the next-byte classes if the text holds bytes >= 0x80: ask its first MiB
            d_pair_hi.alloc(257 * sizeof(unsigned long long));
            CDB_HIP(hipMemsetAsync(d_pair_hi.p, 0, 257 * sizeof(unsigned long long), s));
            hipLaunchKernelGGL(sa_sample_high_kernel, dim3(64), dim3(256), 0, s, text, std::min<uint64_t>(n, 1u << 20), reinterpret_cast<uint32_t*>(d_pair_hi.as<unsigned long long>() + 256));
            uint32_t any = 0;
            CDB_HIP(hipMemcpyAsync(&any, d_pair_hi.as<unsigned long long>() + 256, sizeof(any), hipMemcpyDeviceToHost, s));
            CDB_HIP(hipStreamSynchronize(s));
            pair_hi_counted = any != 0;
        }
        int t = ix.prof.begin(s);
        if (pair_hi_counted)
            hipLaunchKernelGGL(sa_tile_bytecount_kernel<true>, dim3((unsigned)ceil_div((uint64_t)tiles8, (uint64_t)(4 * TBC_TILES_PER_WAVE))), dim3(256), 0, s, text, n,
                               tiles8, d_tbc.as<uint32_t>(), d_pair_hi.as<unsigned long long>());
        else
        hipLaunchKernelGGL(sa_tile_bytecount_kernel<false>, dim3((unsigned)ceil_div((uint64_t)tiles8, (uint64_t)(4 * TBC_TILES_PER_WAVE))), dim3(256), 0, s, text, n,
                           tiles8, d_tbc.as<uint32_t>(), (unsigned long long*)nullptr);
        d_count_src = rs_tile_totals(s, ix.tbw, d_tbc.as<uint32_t>(), tiles8, nullptr);
        ix.prof.end(t, "sa_tile_bytecount", n + (uint64_t)tiles8 * 2048, s);
    } else {
        const int grid = (int)std::min<uint64_t>(ceil_div(n, 256 * 16 * 4), 256 * 8);
        int t = ix.prof.begin(s);
        hipLaunchKernelGGL(sa_bytecount_kernel, dim3(std::max(grid, 1)), dim3(256), 0, s, text, n,
                           d_counts.as<unsigned long long>());
        ix.prof.end(t, "sa_bytecount", n, s);
    }
    uint64_t h_counts[256];
    CDB_HIP(hipMemcpyAsync(h_counts, d_count_src, sizeof(h_counts), hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
    uint16_t h_map[256];
    int sigma = 0;
    bool high_bytes = false;
    for (int b = 0; b < 256; ++b) {
        h_map[b] = h_counts[b] ? (uint16_t)(++sigma) : (uint16_t)0;
        if (b >= 128 && h_counts[b]) high_bytes = true;
    }
    const int symbits = std::max(1, bit_width64((uint64_t)sigma));
    CDB_HIP(hipMemcpyAsync(d_symmap.p, h_map, sizeof(h_map), hipMemcpyHostToDevice, s));

    // Speculative pair count of the MSD-first sort (pair form) on a SECOND stream, beside the key-width sample below: both only
    // need the symbol codes, the sample is a chain of small latency-bound kernels (0.7 ms) and the count a 0.4 ms sweep.
    // Whether the sort takes the pair form is known only after the sample (6-symbol keys); if not, the counts are dropped.
    DevBuf d_pc_spec, d_tc;  // d_tc: per-tile top-digit counts of the look-back-free generated pass (option gen_prebased)
    bool pc_spec = false, tc_spec = false;
    const unsigned long long* d_tc_totals = nullptr;
    struct AuxJoin {  // (declared after the buffer: an exception on the way still waits for the second stream before the
        hipStream_t a = nullptr;  //  buffer goes back to the pool)
        ~AuxJoin() { if (a) (void)hipStreamSynchronize(a); }
    } aux_join;
    if (!big && sizeof(V) == 4 && n >= (1ull << 24) && ix.overlap_paircount && ix.msd_first && ix.msd_pair && ix.narrow_keys &&
        ix.fuse_keygen && ix.flags_in_last_pass && ix.initial_passes == 0 && ix.digit_bits == 0 && ix.key_coding != 1 && sigma + 1 <= 128 &&
        (ix.sort_variant == 0 || ix.sort_variant == 31 || ix.sort_variant == 33) && rs_variant_has_gen(ix.sort_variant) && rs_atomic_rank_ok(s)) {
        if (!ix.aux_stream) CDB_HIP(hipStreamCreateWithFlags(&ix.aux_stream, hipStreamNonBlocking));
        for (hipEvent_t& e : ix.aux_ev)
            if (!e) CDB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const uint32_t pb = (uint32_t)sigma + 1u, np = pb * pb;
        // look-back-free generated pass (radix_sort.h: TextGen::tile_base): count the top digit per 8 Ki-position tile instead of
        // the pairs of the whole text — same sweep, and the column sums are the top-digit histogram
        struct { uint32_t pair_span = 0, pair_r = 0, pair_s = 0; } pcs;
        uint32_t span_spec = 0;
        if (ix.gen_prebased) {
            const uint64_t P4 = (uint64_t)pb * pb * pb * pb;
            span_spec = (uint32_t)std::min<uint64_t>((1ull << 32) / P4, (uint64_t)np);
            if (span_spec == 0 || ceil_div((uint64_t)np, (uint64_t)span_spec) > 256 || !rs_pair_setup(pcs, pb, span_spec)) span_spec = 0;
        }
        if (span_spec) {
            const uint32_t tiles8 = (uint32_t)ceil_div(n, (uint64_t)RS_GEN8_TILE);
            d_tc.alloc((size_t)tiles8 * 256 * sizeof(uint32_t));
            CDB_HIP(hipEventRecord(ix.aux_ev[0], s));
            CDB_HIP(hipStreamWaitEvent(ix.aux_stream, ix.aux_ev[0], 0));
            aux_join.a = ix.aux_stream;
            int t = ix.prof.begin(ix.aux_stream);
            hipLaunchKernelGGL(sa_tile_paircount_kernel, dim3((unsigned)ceil_div((uint64_t)tiles8, (uint64_t)(4 * TPC_TILES_PER_WAVE))), dim3(256), 0, ix.aux_stream,
                               text, n, (const uint16_t*)d_symmap.as<uint16_t>(), pb, pcs.pair_r, pcs.pair_s, tiles8, d_tc.as<uint32_t>());
            hipLaunchKernelGGL(sa_tile_docend_fix_kernel, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(D, 256), 1024))), dim3(256), 0,
                               ix.aux_stream, text, doc_start, D, n, (const uint16_t*)d_symmap.as<uint16_t>(), pb, pcs.pair_r, pcs.pair_s, d_tc.as<uint32_t>());
            d_tc_totals = rs_tile_totals(ix.aux_stream, ix.tbw, d_tc.as<uint32_t>(), tiles8, nullptr);
            ix.prof.end(t, "sa_tile_paircount", n + D * 18 + (uint64_t)tiles8 * 2048, ix.aux_stream);
            CDB_HIP(hipEventRecord(ix.aux_ev[1], ix.aux_stream));
            pc_spec = tc_spec = true;
        } else {
        d_pc_spec.alloc((size_t)np * sizeof(uint64_t));
        CDB_HIP(hipMemsetAsync(d_pc_spec.p, 0, (size_t)np * sizeof(uint64_t), s));
        CDB_HIP(hipEventRecord(ix.aux_ev[0], s));
        CDB_HIP(hipStreamWaitEvent(ix.aux_stream, ix.aux_ev[0], 0));
        aux_join.a = ix.aux_stream;
        int t = ix.prof.begin(ix.aux_stream);
        // (one workgroup per CU: the sample's kernels find room beside it)
        hipLaunchKernelGGL(sa_paircode_kernel, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(n, 1024 * 16 * 4), 256))), dim3(1024),
                           np * sizeof(uint32_t), ix.aux_stream, text, n, (const uint16_t*)d_symmap.as<uint16_t>(), pb, d_pc_spec.as<unsigned long long>());
        hipLaunchKernelGGL(sa_docend_pair_kernel, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(D, 256), 1024))), dim3(256), 0,
                           ix.aux_stream, text, doc_start, D, n, (const uint16_t*)d_symmap.as<uint16_t>(), pb, d_pc_spec.as<unsigned long long>());
        ix.prof.end(t, "sa_paircode", n + D * 18, ix.aux_stream);
        CDB_HIP(hipEventRecord(ix.aux_ev[1], ix.aux_stream));
        pc_spec = true;
        }
    }

    // Key width of the initial sort.  Every suffix left unresolved costs a refinement round (compaction,
    // gathers, a further sort, inverse-array traffic for doubling) that is an order of magnitude more
    // expensive per element than one more radix pass, so the key takes as many symbols as it needs for
    // the expected unresolved share to drop below ~1/64.  The share is estimated from the text itself:
    // S pseudo-random suffixes are keyed with the maximal width and sorted, and the number of adjacent
    // sample pairs agreeing on their first k symbols gives the pair-collision probability c_k, hence
    // ~ n * c_k of all suffixes share their k-prefix with another one.  (An order-0 symbol model is far
    // too optimistic for correlated text such as multi-byte UTF-8.)  Small corpora use the order-0 model.
    int nsym;
    double est_k0 = 0;        // the sample's measured pair-collision probability at est_k0n symbols (>= 50 colliding pairs: meaningful)
    int est_k0n = 0;
    uint64_t refine_depth0 = 0;  // symbols every key of the initial sort covers for certain (0: nsym; variable-length keys: fewer)
    DevBuf d_vl_bytelen;         // variable-length keys: text byte -> code-word length (the refinement recovers every group's exact depth)
    uint32_t vl_kb1 = 0;
    const int kmax = std::min(64 / symbits, 16);
    if (ix.initial_passes > 0) {
        const int passes = std::min(ix.initial_passes, 8);
        nsym = std::min((8 * passes) / symbits, 64 / symbits);
    } else if (n >= (1ull << 24)) {
        const uint64_t S = n >= (1ull << 32) ? 1ull << 22 : 1ull << 21;  // S^2 / 2 sample pairs must resolve 1 / (64 n)
        DevBuf sk0, sk1, d_eq;
        sk0.alloc(S * 8);
        sk1.alloc(S * 8);
        d_eq.alloc(32 * 8);
        CDB_HIP(hipMemsetAsync(d_eq.p, 0, 32 * 8, s));
        hipLaunchKernelGGL(sa_sample_keys_kernel, dim3((unsigned)ceil_div(S, 256)), dim3(256), 0, s, text, doc_start, D, n,
                           (const uint16_t*)d_symmap.as<uint16_t>(), symbits, kmax, S, sk0.as<uint64_t>());
        const int ssel = radix_sort<uint64_t, NoVal>(s, ix.rws, ix.prof, sk0.as<uint64_t>(), sk1.as<uint64_t>(), (NoVal*)nullptr,
                                                     (NoVal*)nullptr, S, 0, kmax * symbits, nullptr);
        hipLaunchKernelGGL(sa_sample_count_kernel, dim3((unsigned)ceil_div(S, 256)), dim3(256), 0, s,
                           (const uint64_t*)(ssel ? sk1 : sk0).as<uint64_t>(), S, symbits, kmax, d_eq.as<unsigned long long>());
        uint64_t h_eq[32];
        CDB_HIP(hipMemcpyAsync(h_eq, d_eq.p, sizeof(h_eq), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        const double pairs = (double)S * (double)S / 2.0;
        if (getenv("CDB_DEBUG_SAMPLE")) {
            for (int k = 1; k <= kmax; ++k) std::fprintf(stderr, "[sample] k=%d adjacent-equal=%llu\n", k, (unsigned long long)h_eq[k]);
        }
        nsym = kmax;
        for (int k = 1; k <= kmax; ++k) {
            if ((double)n * ((double)h_eq[k] / pairs) <= 1.0 / 64.0) {
                nsym = k;
                break;
            }
        }
        // Bucket-wise build: one symbol fewer can save a whole radix pass AND a narrower auxiliary word in every record (16 GiB of
        // UTF-8: 7 symbols = 6 passes over 12-byte records, 6 symbols = 5 passes over 10-byte records), which is worth more than
        // the refinement of the extra unresolved suffixes as long as those stay few.  Compared in bytes moved per suffix:
        // passes x 2 x record bytes against (unresolved share) x ~2000 B — what a text-extension round costs per unresolved
        // suffix (compaction, a 64-bit sort, gathers; measured on that corpus: 694 instead of 729 ms with 1.9 % unresolved).
        if (big && ix.key_cost_model && nsym >= 3 && sizeof(V) == 8) {
            auto plan_bytes = [&](int k) -> double {
                unsigned __int128 v = 1;
                for (int i = 0; i + 1 < k; ++i) {
                    v *= (unsigned)(sigma + 1);
                    if (v > ((unsigned __int128)1 << 56)) return 1e9;
                }
                const int bb = bit_width64((uint64_t)(v - 1));
                const int w = bb <= 32 ? 1 : (bb <= 40 ? 2 : 4);
                return (double)std::max(1, (bb + 7) / 8) * 2.0 * (8.0 + w);
            };
            const double refine_bytes = 2000.0;
            const int k1 = nsym - 1;
            const double u0 = (double)n * ((double)h_eq[nsym] / pairs), u1 = (double)n * ((double)h_eq[k1] / pairs);
            if (u1 <= 1.0 / 40.0 && plan_bytes(k1) + u1 * refine_bytes < plan_bytes(nsym) + u0 * refine_bytes) nsym = k1;
        }
        for (int k = 1; k <= kmax; ++k)  // the longest prefix the sample still sees often enough to measure
            if (h_eq[k] >= 50) {
                est_k0n = k;
                est_k0 = (double)h_eq[k] / pairs;
            }
    } else {
        double pc = 0;
        for (int b = 0; b < 256; ++b) {
            const double p = (double)h_counts[b] / (double)n;
            pc += p * p;
        }
        const double need = pc < 0.999 ? std::log(64.0 * (double)n) / -std::log(pc) : 1e9;
        nsym = (int)std::min<double>(std::ceil(need), 64.0);
    }
    if (ix.debug_starve_group == 2 && !ix.rws.plain_order) ix.rws.debug_poison = true;  // (test hook: the initial sort "starves")
    if (ix.key_symbols > 0 && ix.initial_passes == 0) nsym = ix.key_symbols;
    nsym = std::min(std::max(nsym, 1), std::min(64 / symbits, 32));
    // digit width of the initial sort: whole symbols per digit when that costs no extra pass — the
    // digit then takes at most alphabet+1 values, which lengthens the per-digit runs of a tile
    // (better write coalescing) for small alphabets such as ASCII text
    int dbits = 8;
    if (symbits <= 8) {
        const int dsym = 8 / symbits;
        if ((int)ceil_div(nsym, dsym) <= (int)ceil_div(nsym * symbits, 8)) dbits = dsym * symbits;
    }
    if (ix.digit_bits > 0) dbits = ix.digit_bits;
    if (big) {
        // the bucket-wise sort partitions by the FIRST symbol and gathers the keys behind it bucket by bucket
        nsym = std::min(nsym, HC_MAXSYM);  // (the record gather reads two 8-byte windows behind the first symbol)
        dbits = std::min(symbits, 8);
        // Symbols that ride along for free: a bucket's key is the dense number of the nsym - 1 symbols behind the first,
        // sorted in whole 8-bit passes — one more symbol that still fits the last pass costs nothing and leaves fewer
        // suffixes unresolved (8 GiB of Zipf text: 10 instead of 9 symbols in the same 7 passes).
        if (ix.narrow_keys && ix.initial_passes == 0 && nsym > 1) {
            auto bits_of = [&](int k) {  // bits of (alphabet + 1)^k - 1; 999 beyond 56 bits
                unsigned __int128 v = 1;
                for (int i = 0; i < k; ++i) {
                    v *= (unsigned)sigma + 1u;
                    if (v > ((unsigned __int128)1 << 56)) return 999;
                }
                return bit_width64((uint64_t)(v - 1));
            };
            const int passes = (int)ceil_div(bits_of(nsym - 1), 8);
            while (bits_of(nsym - 1) <= 56 && nsym < HC_MAXSYM && bits_of(nsym) <= 8 * passes) ++nsym;
        }
    }
    // Key coding.  Bit-aligned symbols (base 2^symbits) waste log2(2^symbits / (alphabet + 1)) bits per
    // symbol; the dense base-(alphabet + 1) number is used when it saves a whole radix pass (95-symbol
    // ASCII: 6 symbols = 40 bits = 5 passes instead of 42 bits = 6).  Its digits are not whole symbols, so
    // the per-pass histograms come from a counting pre-pass over the text instead of the byte counts.
    int key_bits = nsym * symbits;
    uint32_t kbase = 1u << symbits;
    bool dense = false;
    if (!big && ix.digit_bits == 0 && ix.initial_passes == 0 && ix.key_coding != 1 && sigma < 255) {
        const unsigned __int128 B = (unsigned)sigma + 1u;
        auto bits_of = [&](int k) {  // bits of B^k - 1; 999 when beyond 56 bits
            unsigned __int128 v = 1;
            for (int i = 0; i < k; ++i) {
                v *= B;
                if (v > ((unsigned __int128)1 << 56)) return 999;
            }
            return bit_width64((uint64_t)(v - 1));
        };
        const int bd = bits_of(nsym);
        const int passes_dense = (int)ceil_div(bd, 8), passes_aligned = (int)ceil_div(key_bits, dbits);
        if (bd <= 56 && (passes_dense < passes_aligned || ix.key_coding == 2)) {
            dense = true;
            while (nsym < HC_MAXSYM && bits_of(nsym + 1) <= 8 * passes_dense) ++nsym;  // symbols that ride along for free
            key_bits = bits_of(nsym);
            kbase = (uint32_t)sigma + 1u;
            dbits = 8;
        }
    }
    const uint64_t kmagic = (kbase & (kbase - 1u)) ? (uint64_t)(~0ull / kbase) + 1ull : 0ull;
    st.dense_keys = dense ? 1 : 0;
    st.digit_bits = dbits;
    st.key_symbols = nsym;
    st.symbol_bits = symbits;
    st.alphabet = sigma;

    // ---- 2 + 3. keys + entries, initial sort
    DevBuf sorted_keys, sa_buf, flags;
    DevBuf sa_hi_buf;         // packed output (index_impl.h: Sa40): bits 32..39 of every entry; sa_buf then holds the low words
    bool packed_out = false;
    struct Depth1Fold {
        uint64_t nend = 0, nlow = 0, nhigh = 0;
        bool fold = false, done = false;  // done: the segmented sort really wrote the bucket with its blocks swapped
    };
    std::vector<Depth1Fold> depth1;          // ... per bucket: how its suffixes split by the class of their second symbol
    std::vector<CompatBucket> folded_roots;  // bucket-wise build in the reference's root order: the first-symbol buckets
    SortStats ss;
    const bool fused = big || (ix.fuse_keygen && (dense || dbits == symbits) && nsym <= HC_MAXSYM &&
                               rs_variant_has_gen(ix.sort_variant));
    st.fused_keygen = fused ? 1 : 0;
    std::vector<uint64_t> h_hist;  // [nsym][256] digit histograms of the LSD passes (fused path)
    // digit histograms of the dense keys, counted in one sweep over the text
    auto key_histograms = [&](int npass, uint32_t digit_mask) {
        DevBuf d_kh;
        d_kh.alloc((size_t)npass * 256 * sizeof(uint64_t));
        CDB_HIP(hipMemsetAsync(d_kh.p, 0, (size_t)npass * 256 * sizeof(uint64_t), s));
        const int grid = (int)std::min<uint64_t>(ceil_div(n, KH_TILE), 256 * 8);
        int t = ix.prof.begin(s);
        const bool by3 = ix.keyhist3 && nsym % 3 == 0 && nsym >= 6 && nsym <= 15;
#define CDB_KH3(PARTS)                                                                                              \
    hipLaunchKernelGGL((sa_keyhist3_kernel<PARTS>), dim3(std::max(grid, 1)), dim3(256), 0, s, text, doc_start, D, n, \
                       (const uint16_t*)d_symmap.as<uint16_t>(), kbase, npass, ix.text_padded, d_kh.as<unsigned long long>(), digit_mask)
        if (by3 && nsym == 6) CDB_KH3(2);
        else if (by3 && nsym == 9) CDB_KH3(3);
        else if (by3 && nsym == 12) CDB_KH3(4);
        else if (by3 && nsym == 15) CDB_KH3(5);
        else
            hipLaunchKernelGGL(sa_keyhist_kernel, dim3(std::max(grid, 1)), dim3(256), 0, s, text, doc_start, D, n,
                               (const uint16_t*)d_symmap.as<uint16_t>(), kbase, nsym, npass, ix.text_padded,
                               d_kh.as<unsigned long long>(), digit_mask);
#undef CDB_KH3
        ix.prof.end(t, "sa_keyhist", n, s);
        h_hist.assign((size_t)npass * 256, 0);
        CDB_HIP(hipMemcpyAsync(h_hist.data(), d_kh.p, h_hist.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
    };
    // MSD-first form of the split sort (radix_sort.h: radix_sort_msd): dense keys of 33..40 bits, 16 Ki-key tiles with the
    // one-atomic ranking, flags written by the last pass.  Only the TOP digit's histogram comes from the text sweep.
    const bool use_msd = fused && dense && !big && sizeof(V) == 4 && ix.narrow_keys && ix.msd_first && ix.flags_in_last_pass &&
                         dbits == 8 && key_bits > 32 && key_bits <= 40 && rs_atomic_rank_ok(s) &&
                         ((ix.sort_variant == 0 && n >= (1ull << 23)) || ix.sort_variant == 31 || ix.sort_variant == 33);
    st.msd_first = use_msd ? 1 : 0;
    // pair form: 6-symbol keys whose top digit is a function of the first two symbols (TextGen::msd_pair); otherwise the
    // top digit is key >> 32, counted by the key sweep
    unsigned long long msd_m = 1ull << 32;
    uint32_t msd_span = 0;
    struct { uint32_t pair_span = 0, pair_r = 0, pair_s = 0; } pair_consts;
    std::vector<uint64_t> h_top;
    if (use_msd && ix.msd_pair && nsym == 6 && kbase <= 128) {
        const uint64_t P4 = (uint64_t)kbase * kbase * kbase * kbase;  // (< 2^28)
        msd_span = (uint32_t)std::min<uint64_t>((1ull << 32) / P4, (uint64_t)kbase * kbase);
        if (ceil_div((uint64_t)kbase * kbase, (uint64_t)msd_span) > 256) msd_span = 0;
        if (msd_span && !rs_pair_setup(pair_consts, kbase, msd_span)) msd_span = 0;  // (24-bit arithmetic of the pair generator)
    }
    if (pc_spec) {  // the second stream joins the first (whether or not its counts are used)
        CDB_HIP(hipStreamWaitEvent(s, ix.aux_ev[1], 0));
        aux_join.a = nullptr;
    }
    if (msd_span && tc_spec) {  // the column sums of the per-tile counts ARE the top-digit histogram
        msd_m = (unsigned long long)msd_span * kbase * kbase * kbase * kbase;
        h_top.assign(256, 0);
        CDB_HIP(hipMemcpyAsync(h_top.data(), d_tc_totals, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        st.msd_first = 2;
    } else if (msd_span && pc_spec) {
        msd_m = (unsigned long long)msd_span * kbase * kbase * kbase * kbase;
        const uint32_t np = kbase * kbase;
        std::vector<uint64_t> pc(np);
        CDB_HIP(hipMemcpyAsync(pc.data(), d_pc_spec.p, (size_t)np * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        h_top.assign(256, 0);
        for (uint32_t a = 0; a < np; ++a) h_top[a / msd_span] += pc[a];
        st.msd_first = 2;
    } else if (msd_span && ix.gen_prebased) {  // (no speculative count on the second stream: the same kernels on this one)
        msd_m = (unsigned long long)msd_span * kbase * kbase * kbase * kbase;
        const uint32_t tiles8 = (uint32_t)ceil_div(n, (uint64_t)RS_GEN8_TILE);
        d_tc.alloc((size_t)tiles8 * 256 * sizeof(uint32_t));
        int t = ix.prof.begin(s);
        hipLaunchKernelGGL(sa_tile_paircount_kernel, dim3((unsigned)ceil_div((uint64_t)tiles8, (uint64_t)(4 * TPC_TILES_PER_WAVE))), dim3(256), 0, s, text, n,
                           (const uint16_t*)d_symmap.as<uint16_t>(), kbase, pair_consts.pair_r, pair_consts.pair_s, tiles8, d_tc.as<uint32_t>());
        hipLaunchKernelGGL(sa_tile_docend_fix_kernel, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(D, 256), 1024))), dim3(256), 0, s,
                           text, doc_start, D, n, (const uint16_t*)d_symmap.as<uint16_t>(), kbase, pair_consts.pair_r, pair_consts.pair_s,
                           d_tc.as<uint32_t>());
        d_tc_totals = rs_tile_totals(s, ix.tbw, d_tc.as<uint32_t>(), tiles8, nullptr);
        ix.prof.end(t, "sa_tile_paircount", n + D * 18 + (uint64_t)tiles8 * 2048, s);
        h_top.assign(256, 0);
        CDB_HIP(hipMemcpyAsync(h_top.data(), d_tc_totals, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        tc_spec = true;
        st.msd_first = 2;
    } else if (msd_span) {
        msd_m = (unsigned long long)msd_span * kbase * kbase * kbase * kbase;
        const uint32_t np = kbase * kbase;
        DevBuf d_pc;
        d_pc.alloc((size_t)np * sizeof(uint64_t));
        CDB_HIP(hipMemsetAsync(d_pc.p, 0, (size_t)np * sizeof(uint64_t), s));
        int t = ix.prof.begin(s);
        hipLaunchKernelGGL(sa_paircode_kernel, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(n, 1024 * 16 * 4), 512))), dim3(1024),
                           np * sizeof(uint32_t), s, text, n, (const uint16_t*)d_symmap.as<uint16_t>(), kbase, d_pc.as<unsigned long long>());
        hipLaunchKernelGGL(sa_docend_pair_kernel, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(D, 256), 1024))), dim3(256), 0, s,
                           text, doc_start, D, n, (const uint16_t*)d_symmap.as<uint16_t>(), kbase, d_pc.as<unsigned long long>());
        ix.prof.end(t, "sa_paircode", n + D * 18, s);
        std::vector<uint64_t> pc(np);
        CDB_HIP(hipMemcpyAsync(pc.data(), d_pc.p, (size_t)np * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        h_top.assign(256, 0);
        for (uint32_t a = 0; a < np; ++a) h_top[a / msd_span] += pc[a];
        st.msd_first = 2;
    } else if (fused && dense) {
        const int npass = (int)ceil_div(key_bits, 8);
        key_histograms(npass, use_msd ? 1u << (npass - 1) : 0xFFFFFFFFu);
        if (use_msd) h_top.assign(h_hist.begin() + (size_t)(npass - 1) * 256, h_hist.begin() + (size_t)npass * 256);
    } else if (fused && !big) {
        // per-pass digit histograms from the byte counts + document-head corrections (no key is read)
        DevBuf d_corr;
        const size_t corr_words = (size_t)HC_MAXSYM * 257 + HC_MAXSYM;
        d_corr.alloc(corr_words * sizeof(uint64_t));
        CDB_HIP(hipMemsetAsync(d_corr.p, 0, corr_words * sizeof(uint64_t), s));
        {
            const int grid = (int)std::min<uint64_t>(ceil_div(D, 256), 256 * 4);
            int t = ix.prof.begin(s);
            hipLaunchKernelGGL(sa_headcorr_kernel, dim3(std::max(grid, 1)), dim3(256), 0, s, text, doc_start, D,
                               (const uint16_t*)d_symmap.as<uint16_t>(), nsym, d_corr.as<unsigned long long>(),
                               d_corr.as<unsigned long long>() + (size_t)HC_MAXSYM * 257);
            ix.prof.end(t, "sa_headcorr", D * (16 + (uint64_t)nsym), s);
        }
        std::vector<uint64_t> h_corr(corr_words);
        CDB_HIP(hipMemcpyAsync(h_corr.data(), d_corr.p, corr_words * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        const uint64_t* first = h_corr.data();
        const uint64_t* lencnt = h_corr.data() + (size_t)HC_MAXSYM * 257;
        uint64_t code_count[257] = {0};
        for (int b = 0; b < 256; ++b)
            if (h_map[b]) code_count[h_map[b]] = h_counts[b];
        h_hist.assign((size_t)nsym * 256, 0);
        for (int p = 0; p < nsym; ++p) {
            const int k = nsym - 1 - p;  // LSD: pass 0 sorts on the last symbol of the key
            uint64_t* hp = &h_hist[(size_t)p * 256];
            uint64_t shorter = 0, ends = 0;  // documents shorter than k, bytes they hold
            for (int L = 0; L < k; ++L) {
                shorter += lencnt[L];
                ends += (uint64_t)L * lencnt[L];
            }
            hp[0] = ends + (uint64_t)k * (D - shorter);  // suffixes with fewer than k+1 symbols left
            for (int c = 1; c <= sigma; ++c) {
                uint64_t head = 0;
                for (int j = 0; j < k; ++j) head += first[(size_t)j * 257 + c];
                hp[c] = code_count[c] - head;
            }
        }
    }
    TextGen gen{text, doc_start, d_symmap.as<uint16_t>(), D, (int)ix.bits, kbase, nsym, 0, ix.text_padded};
    double ta = now_ms();
    flags.alloc(n);
    // Layout of the sort records.  WIDE: (u64 key, entry).  When the generated first pass can drop the digit it
    // sorts on (it travels as one byte per element) and the rest of the key fits 32 bits, the other passes
    // move 9 instead of 12 bytes per suffix (SPLIT); keys of <= 32 bits need no extra byte at all (NARROW).
    enum { WIDE, NARROW, SPLIT, SPLIT2 } layout = WIDE;
    if (fused && !big && sizeof(V) == 4 && ix.narrow_keys) {
        if (key_bits <= 32) layout = NARROW;
        else if (key_bits - dbits <= 32) layout = SPLIT;
        else if (key_bits - 2 * dbits <= 32) layout = SPLIT2;  // two low digits in a u16: 10 bytes per suffix
    }
    st.key_layout = (int)layout;
    DevBuf sorted_k32, sorted_low;
    bool flags_by_sort = false;  // the sort's last pass wrote the group flags and the tile sums of the first compaction
    const int low_bits = layout == SPLIT ? dbits : (layout == SPLIT2 ? 2 * dbits : 0);
    const int low_bytes = layout == SPLIT ? 1 : (layout == SPLIT2 ? 2 : 0);
    if (!big && layout != WIDE) {
        if constexpr (sizeof(V) == 4) {
            DevBuf k32[2], vals[2], low[2];
            k32[0].alloc(n * sizeof(uint32_t));
            k32[1].alloc(n * sizeof(uint32_t));
            vals[0].alloc(n * sizeof(V));
            vals[1].alloc(n * sizeof(V));
            if (low_bytes) {
                if (!use_msd) low[0].alloc(n * low_bytes);  // (the MSD-first sort has no travelling byte: only its last pass writes one)
                low[1].alloc(n * low_bytes);
            }
            st.alloc_ms += now_ms() - ta;
            if (getenv("CDB_DEBUG_BUFS"))
                std::fprintf(stderr, "[bufs] k32 %p %p vals %p %p low %p %p flags %p\n", k32[0].p, k32[1].p, vals[0].p, vals[1].p, low[0].p,
                             low[1].p, flags.p);
            int sel = 0;
            // The last pass writes the group flags itself (radix_sort.h: SegFinalKeepArgs) where its kernel configuration
            // can (16 Ki tiles, one-atomic ranking): the flag kernel — 5-6 B read per suffix — is then not needed, and the
            // per-tile counts of unresolved entries for the first compaction are counted on the way.
            SegFinalKeepArgs keep;
            DevBuf edges;
            const bool want_keep = ix.flags_in_last_pass && dbits == 8 &&
                                   (n >= (1ull << 23) || ix.sort_variant == 31 || ix.sort_variant == 33);  // (16 Ki-tile configurations)
            if (want_keep) {
                const uint64_t nbt = ceil_div(n, (uint64_t)SC_TILE);
                ix.scan_partials.ensure(scan_partials_slots(nbt) * sizeof(U2));
                CDB_HIP(hipMemsetAsync(ix.scan_partials.p, 0, nbt * sizeof(U2), s));
                edges.alloc((ceil_div(n, (uint64_t)RS_SEG_TILE) + 256) * 256 * sizeof(SegEdge));  // (MSD-first: one ragged tile per bucket)
                keep.flags = flags.as<uint8_t>();
                keep.edges = edges.as<SegEdge>();
                keep.low_bits = low_bits;
                keep.kbase = kbase;
                keep.kmagic = kmagic;
                keep.tile_sums = ix.scan_partials.as<unsigned long long>();
                keep.sums_tile = SC_TILE;
            }
            if (layout == SPLIT && use_msd && want_keep) {
                // the final pass writes the kept keys in the split layout (u32 = key >> 8, low byte), entries and flags
                if (msd_span) {
                    gen.msd_pair = true;
                    gen.pair_span = pair_consts.pair_span;
                    gen.pair_r = pair_consts.pair_r;
                    gen.pair_s = pair_consts.pair_s;
                } else {
                    gen.msd_shift = 32;
                }
                const bool prebased = tc_spec && msd_span != 0;  // (the counts were made with this span: same rule, same alphabet)
                radix_sort_msd(s, ix.rws, ix.msd_ws, ix.prof, k32[0].as<uint32_t>(), k32[1].as<uint32_t>(), vals[0].as<uint32_t>(),
                               vals[1].as<uint32_t>(), low[1].as<uint8_t>(), n, h_top.data(), gen, msd_m, keep, &ss,
                               prebased ? (const uint32_t*)d_tc.as<uint32_t>() : nullptr, prebased ? &ix.tbw : nullptr, ix.sweep_records);
                st.gen_prebased = prebased ? 1 : 0;
                st.sweep_records = prebased && ix.sweep_records && gen.msd_pair ? 1 : 0;
                d_tc.release();
                ix.tbw.base.release();
                sel = 1;
                sorted_low = std::move(low[1]);
                flags_by_sort = ix.rws.keep_applied;
            } else if (layout == SPLIT) {
                gen.low_bits = low_bits;
                sel = radix_sort_split<V, uint8_t>(s, ix.rws, ix.prof, k32[0].as<uint32_t>(), k32[1].as<uint32_t>(),
                                                   vals[0].as<V>(), vals[1].as<V>(), low[0].as<uint8_t>(), low[1].as<uint8_t>(), n,
                                                   key_bits - low_bits, &ss, ix.sort_variant, dbits, h_hist.data(), &gen, 0, nullptr, -1,
                                                   want_keep ? &keep : nullptr);
                sorted_low = std::move(low[sel]);
                flags_by_sort = ix.rws.keep_applied;
            } else if (layout == SPLIT2) {
                gen.low_bits = low_bits;
                sel = radix_sort_split<V, uint16_t>(s, ix.rws, ix.prof, k32[0].as<uint32_t>(), k32[1].as<uint32_t>(),
                                                    vals[0].as<V>(), vals[1].as<V>(), low[0].as<uint16_t>(), low[1].as<uint16_t>(),
                                                    n, key_bits - low_bits, &ss, ix.sort_variant, dbits, h_hist.data(), &gen, 0, nullptr, -1,
                                                    want_keep ? &keep : nullptr);
                sorted_low = std::move(low[sel]);
                flags_by_sort = ix.rws.keep_applied;
            } else {
                sel = radix_sort<uint32_t, V>(s, ix.rws, ix.prof, k32[0].as<uint32_t>(), k32[1].as<uint32_t>(), vals[0].as<V>(),
                                              vals[1].as<V>(), n, 0, key_bits, &ss, ix.sort_variant, dbits, h_hist.data(), &gen);
            }
            CDB_HIP(hipStreamSynchronize(s));
            sorted_k32 = std::move(k32[sel]);
            sa_buf = std::move(vals[sel]);
        }
    } else if (!big) {
        DevBuf keys[2], vals[2];
        keys[0].alloc(n * sizeof(uint64_t));
        keys[1].alloc(n * sizeof(uint64_t));
        vals[0].alloc(n * sizeof(V));
        vals[1].alloc(n * sizeof(V));
        st.alloc_ms += now_ms() - ta;
        int sel;
        if (fused) {
            sel = radix_sort<uint64_t, V>(s, ix.rws, ix.prof, keys[0].as<uint64_t>(), keys[1].as<uint64_t>(), vals[0].as<V>(),
                                          vals[1].as<V>(), n, 0, key_bits, &ss, ix.sort_variant, dbits, h_hist.data(), &gen);
        } else {
            int t = ix.prof.begin(s);
            hipLaunchKernelGGL((sa_keygen_kernel<V>), dim3((unsigned)ceil_div(n, KG_TILE)), dim3(256), 0, s, text,
                               doc_start, D, n, (int)ix.bits, d_symmap.as<uint16_t>(), kbase, nsym, ix.text_padded,
                               keys[0].as<uint64_t>(), vals[0].as<V>());
            ix.prof.end(t, "sa_keygen", n * (1 + sizeof(uint64_t) + sizeof(V)), s);
            sel = radix_sort<uint64_t, V>(s, ix.rws, ix.prof, keys[0].as<uint64_t>(), keys[1].as<uint64_t>(), vals[0].as<V>(),
                                          vals[1].as<V>(), n, 0, key_bits, &ss, ix.sort_variant, dbits);
        }
        CDB_HIP(hipStreamSynchronize(s));
        sorted_keys = std::move(keys[sel]);
        sa_buf = std::move(vals[sel]);
    } else {
        // Streamed bucket-wise sort (a double-buffered LSD sort of (u64 key, u64 entry) pairs would need 32 n
        // bytes — 256 GiB at n = 2^33): ONE generated pass partitions the ENTRIES by first symbol — its keys exist only
        // inside the tile — and every bucket then gathers its keys from the text (its entries are still in
        // text order, so the gather walks the text front to back), LSD-sorts (key, entry) on the remaining
        // symbols and writes its group flags.  Peak = 8 n (entries) + 24 x the largest bucket.
        DevBuf E, KT[2], ET;  // (E: allocated below, once it is known whether the entries are partitioned at all)
        const int top_shift = (nsym - 1) * symbits;
        // Every position starts a suffix, and a suffix is never empty: the first symbol is never "end of document", so
        // the partition digit is code - 1 (0 .. alphabet - 1: 8 bits even for all 256 byte values, where the codes
        // 1 .. 256 themselves need 9) and its histogram is simply the byte histogram.
        // Bucket order = ascending code, or — reference_compat order on text with bytes >= 0x80 — the reference's child
        // order of the ROOT radix node (index.h:66-73: bytes 0x80..0xFF in front of 0x00..0x7F): the first-symbol buckets
        // then sit where apply_reference_order's root rotation (three reversals of the whole array) would put them.
        const bool root_folded = ix.reference_compat && high_bytes && ix.fold_root && n > std::max<uint64_t>(4096, n / 256);
        std::vector<int> border;  // symbol codes in bucket order
        if (root_folded) {
            for (int b = 128; b < 256; ++b)
                if (h_map[b]) border.push_back(h_map[b]);
            for (int b = 0; b < 128; ++b)
                if (h_map[b]) border.push_back(h_map[b]);
        } else {
            for (int c = 1; c <= sigma; ++c) border.push_back(c);
        }
        std::vector<uint64_t> first_by_code(258, 0), first_digit(256, 0);
        uint16_t h_map_first[256];
        uint8_t h_slotmap[260] = {0};  // symbol code -> bucket slot (the fused records pass)
        {
            int slot_of_code[258] = {0};
            for (int k = 0; k < sigma; ++k) slot_of_code[border[k]] = k;
            for (int c = 1; c <= sigma; ++c) h_slotmap[c] = (uint8_t)slot_of_code[c];
            for (int b = 0; b < 256; ++b) {
                h_map_first[b] = h_map[b] ? (uint16_t)slot_of_code[h_map[b]] : (uint16_t)0;
                if (h_map[b]) {
                    first_by_code[h_map[b]] = h_counts[b];
                    first_digit[slot_of_code[h_map[b]]] = h_counts[b];
                }
            }
        }
        const uint64_t* h_first = first_by_code.data();  // (indexed by symbol code 1 .. alphabet)
        DevBuf d_symmap_first;
        d_symmap_first.alloc(256 * sizeof(uint16_t));
        CDB_HIP(hipMemcpyAsync(d_symmap_first.p, h_map_first, sizeof(h_map_first), hipMemcpyHostToDevice, s));
        // (look-back-free generated passes: which byte's per-tile counts feed bucket slot d; 0xFFFF = no such slot)
        DevBuf d_src_col;
        uint16_t h_src_col[256];
        for (int d = 0; d < 256; ++d) h_src_col[d] = 0xFFFFu;
        for (int b = 0; b < 256; ++b)
            if (h_map[b]) h_src_col[h_map_first[b]] = (uint16_t)b;
        if (tile_bytes) {
            d_src_col.alloc(sizeof(h_src_col));
            CDB_HIP(hipMemcpyAsync(d_src_col.p, h_src_col, sizeof(h_src_col), hipMemcpyHostToDevice, s));
        }
        CDB_HIP(hipStreamSynchronize(s));  // (h_map_first is a stack array)
        uint64_t maxb = 0;
        for (int c = 1; c <= sigma; ++c) maxb = std::max(maxb, h_first[c]);
        // Bucket records.  Inside a bucket the first symbol is constant, so the sort key is the remaining
        // nsym - 1 symbols — as a dense base-(alphabet + 1) number, split like the records of the single-sort
        // path when it fits 32 bits + one or two low digits: (u32, entry, u8 / u16) instead of (u64, entry).
        const uint32_t bbase = (uint32_t)sigma + 1u;
        int bbits = 0;  // bits of bbase^(nsym-1) - 1; 999 beyond 56 bits
        {
            unsigned __int128 v = 1;
            for (int i = 0; i + 1 < nsym && bbits != 999; ++i) {
                v *= bbase;
                if (v > ((unsigned __int128)1 << 56)) bbits = 999;
            }
            if (bbits != 999) bbits = bit_width64((uint64_t)(v - 1));
        }
        // ---- variable-length keys (vl_code.h): the first B - 1 bits of the suffix's alphabetic code stream + a "continues" bit instead
        // of the dense number, when the model of key_cost_model says they are cheaper — skewed text, where equal prefixes are made of
        // frequent symbols with short code words (8 GiB of Zipf-64 text: 40 key bits = 5 passes over 10-byte records instead of 54 bits
        // = 7 passes over 12-byte records).  Needs the sweep form (the key of a position is a window into the tile's bit stream).
        VlCode vlc;
        int vl_bits = 0;
        DevBuf d_vl_sym, d_vl_dec;
        if (sizeof(V) == 8 && ix.vl_keys != 0 && !ix.vl_off_once && tile_bytes && ix.sweep_records && ix.segmented_sort && ix.pack_entries &&
            ix.narrow_keys && (int)ix.bits + ix.off_bits <= 40 && sigma >= 2 && sigma <= 127 && nsym > 1 && rs_atomic_rank_ok(s) &&
            (ix.vl_keys != 2 || (ix.initial_passes == 0 && ix.key_symbols == 0))) {
            uint64_t cnts[257] = {0};
            cnts[0] = D;
            for (int c = 1; c <= sigma; ++c) cnts[c] = h_first[c];
            if (vl_build(cnts, sigma, vlc)) {
                auto rec_bytes = [](int b) { return 8.0 + (b <= 32 ? 1.0 : (b <= 40 ? 2.0 : 4.0)); };
                const double refine_bytes = 2000.0;  // (what a text-extension round costs per unresolved suffix: key_cost_model above)
                // the sample measured the dense key's unresolved share; the order-0 model q^k says what it "should" be — their ratio
                // carries the text's correlation over to the estimate for the code stream: share(B) = ratio n q 2^(-rate (B - 1))
                // (measured where the sample SEES collisions — the longest prefix with >= 50 colliding sample pairs; at the key width
                //  that was chosen it sees none by construction, and "none" is no measurement: a first version divided 0 by the model
                //  and took 32-bit code-stream keys for 16 GiB of printable ASCII, 12 % unresolved)
                const double ratio = est_k0n > 0 ? std::min(1e3, std::max(0.2, est_k0 / std::max(std::pow(vlc.q, (double)est_k0n), 1e-300))) : 1.0;
                const double u_fixed = std::min(1.0, ratio * (double)n * std::pow(vlc.q, (double)nsym));
                const double cost_fixed = bbits != 999 ? std::ceil(bbits / 8.0) * 2.0 * rec_bytes(bbits) + u_fixed * refine_bytes : 1e9;
                auto share = [&](int B) { return std::min(1.0, ratio * (double)n * vlc.q * std::pow(2.0, -vlc.rate * (double)(B - 1))); };
                int best_b = 0;
                double best = 1e18;
                // (with these keys the text-extension rounds start every group at the depth its key really covered, so the open
                //  suffixes are settled in one or two rounds: ~1000 B each — 8 GiB of Zipf text, B = 40: 46 ms for 206 M of them)
                const double refine_bytes_vl = 1000.0;
                for (int B : {32, 40, 48, 56}) {
                    const double u = share(B);
                    if (u > 1.0 / 24.0) continue;  // (text extension wants few open suffixes; the estimate errs on the high side)
                    const double c = (double)(B / 8) * 2.0 * rec_bytes(B) + u * refine_bytes_vl + 6.0;  // (+ 6: the tile's bit offsets)
                    if (c < best) {
                        best = c;
                        best_b = B;
                    }
                }
                if (ix.vl_keys >= 16) best_b = std::min(56, (ix.vl_keys + 7) / 8 * 8);
                else if (ix.vl_keys == 1 && !best_b) best_b = 56;
                // (automatic mode also wants a code that really is shorter than the dense number's log2(alphabet + 1) bits per symbol:
                //  on a flat alphabet the code stream can only lose)
                const bool shorter = vlc.avg_len <= 0.93 * std::log2((double)sigma + 1.0);
                if (best_b && (ix.vl_keys != 2 || (shorter && best < 0.93 * cost_fixed))) {
                    vl_bits = best_b;
                    bbits = vl_bits;
                    std::vector<uint16_t> sym(256, 0), dec(128, (uint16_t)(0xFFu | ((unsigned)vlc.end_len << 8)));
                    for (int c = 0; c <= sigma; ++c) {
                        sym[c] = (uint16_t)(vlc.bits[c] | ((unsigned)vlc.len[c] << 8));
                        if (c == 0) continue;
                        const int l = vlc.len[c];
                        for (unsigned x = 0; x < (1u << (VL_MAX_LEN - l)); ++x)
                            dec[((unsigned)vlc.bits[c] << (VL_MAX_LEN - l)) | x] = (uint16_t)(h_slotmap[c] | ((unsigned)l << 8));
                    }
                    d_vl_sym.alloc(256 * sizeof(uint16_t));
                    d_vl_dec.alloc(128 * sizeof(uint16_t));
                    CDB_HIP(hipMemcpyAsync(d_vl_sym.p, sym.data(), 256 * sizeof(uint16_t), hipMemcpyHostToDevice, s));
                    CDB_HIP(hipMemcpyAsync(d_vl_dec.p, dec.data(), 128 * sizeof(uint16_t), hipMemcpyHostToDevice, s));
                    CDB_HIP(hipStreamSynchronize(s));  // (sym, dec)
                    st.vl_key_bits = vl_bits;
                    st.vl_avg_len = vlc.avg_len;
                    st.vl_rate = vlc.rate;
                    st.vl_est_unresolved = share(vl_bits);
                    st.fixed_est_unresolved = u_fixed;
                    // every key covers at least this many whole symbols behind the bucket symbol: where the refinement starts
                    refine_depth0 = 1 + (uint64_t)((vl_bits - 1) / vlc.max_len);
                    // (the text-extension rounds start every group at the depth its key really covered: code-word length per text byte)
                    std::vector<uint8_t> blen(256, 0);
                    for (int b = 0; b < 256; ++b) blen[b] = h_map[b] ? vlc.len[h_map[b]] : (uint8_t)0;
                    d_vl_bytelen.alloc(256);
                    CDB_HIP(hipMemcpyAsync(d_vl_bytelen.p, blen.data(), 256, hipMemcpyHostToDevice, s));
                    CDB_HIP(hipStreamSynchronize(s));  // (blen)
                    vl_kb1 = (uint32_t)(vl_bits - 1);
                }
            }
        }
        // ---- leftover key bits (sweep form): the dense number of nsym - 1 symbols is sorted in whole 8-bit passes; when its range
        // leaves a factor part_m >= 2 of room, the NEXT symbol quantised to part_m levels rides along below it — order-preserving
        // (monotone in the symbol), no extra pass or byte, fewer unresolved suffixes (16 GiB of UTF-8: 206^5 = 2^38.4 in 40 bits ->
        // part_m = 2).  "The key ends inside the document" = key mod (base * part_m) == 0.  Like the variable-length keys it exists
        // in the sweep kernels only; should the build fall back to another records form it is redone without.
        TextGen part_gen{};
        uint32_t part_m = 1;
        if (!vl_bits && sizeof(V) == 8 && ix.partial_symbol && !ix.vl_off_once && tile_bytes && ix.sweep_records && ix.segmented_sort &&
            ix.pack_entries && ix.narrow_keys && (int)ix.bits + ix.off_bits <= 40 && sigma <= 254 && nsym > 1 && nsym - 1 <= 10 &&
            bbits != 999 && bbits <= 56 && rs_atomic_rank_ok(s) && rs_sweep_records_ok(bbase, nsym) && true) {
            const int pbits = 8 * (int)ceil_div(bbits, 8);
            unsigned __int128 range = 1;
            for (int i = 0; i + 1 < nsym; ++i) range *= bbase;
            const uint64_t room = (uint64_t)((((unsigned __int128)1) << pbits) / range);
            const uint32_t m = (uint32_t)std::min<uint64_t>(room, bbase - 1);
            if (m >= 2 && rs_part_setup(part_gen, bbase, m)) {
                part_m = m;
                bbits = bit_width64((uint64_t)(range * m - 1));
                st.partial_levels = (int)m;
            }
        }
        // (wider keys, up to 56 bits: dense u64 keys without low digits — still fewer passes than bit-aligned
        //  symbols, the grouped gather and no separate histogram pass)
        // packed entries (sa_bucket_records_packed_kernel): 8-byte entries below 2^40 travel as u32 + one byte on top of the
        // low digits; up to three low digits then keep every key of <= 56 bits in (u32, u32, u8 / u16 / u32) records
        const bool packed = sizeof(V) == 8 && ix.pack_entries && (int)ix.bits + ix.off_bits <= 40 && ix.narrow_keys && nsym > 1 &&
                            bbits <= 56;
        const int blow = bbits <= 32 ? 0 : (bbits <= 40 ? 8 : (bbits <= 48 ? 16 : (bbits <= 56 ? (packed ? 24 : 0) : -1)));
        const bool bwide = !packed && bbits > 48 && bbits <= 56;
        // (a one-symbol key leaves nothing behind the bucket symbol: its "ends inside the key" test would look at an
        //  empty remainder, so that corner keeps the plain (u64 key, entry) records)
        const bool brecords = ix.narrow_keys && blow >= 0 && nsym > 1;  // (codes up to 256 are u16 in the record kernel)
        // FUSED form: when the records of ALL buckets fit the memory at once (one bucket group), the generated pass writes the
        // records itself (radix_sort.h: radix_gen_records) — the entries are never partitioned on their own, no gather walks
        // the text bucket by bucket (4 GiB of UTF-8: 22 ms partition + 32 ms gather -> one 28 ms pass + a 6 ms histogram
        // sweep of the records), and the finished entries land in the record buffer the last pass does not read.
        bool fuse_rec = false;
        // (the tile stages symbol CODES as bytes: alphabets of all 256 byte values — codes up to 256 — keep the gather)
        if (brecords && packed && sigma <= 255 && ix.segmented_sort && ix.fuse_records && sizeof(V) == 8 && rs_atomic_rank_ok(s) &&
            (ix.bucket_group_limit == 0 || ix.bucket_group_limit >= n)) {
            size_t fre = 0, tot = 0;
            CDB_HIP(hipMemGetInfo(&fre, &tot));
            const double avail = (double)fre + (double)DevPool::get().cached_bytes();
            const int recb0 = 4 + (blow == 0 ? 1 : (blow == 8 ? 2 : 4)) + 4;
            // (no separate entry array: the last pass writes the finished entries over the record buffer it does not read)
            fuse_rec = avail * 0.85 / (2.0 * recb0 + 1.0 + (sa_packable(ix) ? 1.0 : 0.0)) >= (double)n;
        }
        // packed output (index_impl.h: Sa40): the last pass writes u32 low words over the KEY half it does not read and the high
        // bytes into their own array — 5 instead of 8 bytes written per suffix, and the index keeps 5 n bytes instead of 8 n
        const bool want_pack = sizeof(V) == 8 && sa_packable(ix);
        // (allocate the record buffers NOW: if the device cannot provide them after all — fragmentation, other handles —
        //  nothing has happened yet and the build takes partition + gather instead of failing)
        DevBuf fr_kv[2], fr_e[2], fr_w[2];
        if (fuse_rec) {
            const size_t auxb0 = blow == 0 ? 1 : (blow == 8 ? 2 : 4);
            try {
                for (int q = 0; q < 2; ++q) {
                    if (want_pack) {  // key and entry halves as blocks of their own: the key half of the dead buffer BECOMES the array
                        fr_kv[q].alloc(n * sizeof(uint32_t));
                        fr_e[q].alloc(n * sizeof(uint32_t));
                    } else {
                        fr_kv[q].alloc(n * 2 * sizeof(uint32_t));
                    }
                    fr_w[q].alloc(n * auxb0);
                }
                if (want_pack) sa_hi_buf.alloc(n);
            } catch (const std::exception&) {
                for (int q = 0; q < 2; ++q) {
                    fr_kv[q].release();
                    fr_e[q].release();
                    fr_w[q].release();
                }
                sa_hi_buf.release();
                (void)hipGetLastError();
                fuse_rec = false;
            }
        }
        packed_out = fuse_rec && want_pack;
        // SWEEP form (records_sweep.h): the records do not fit at once, but with counted tile bases a sweep over the text can write
        // the records of any bucket group in place — the entries are not partitioned and no gather re-reads them and the text
        bool sweep_rec = !fuse_rec && tile_bytes && ix.sweep_records && brecords && packed && sigma <= 255 && ix.segmented_sort &&
                         sizeof(V) == 8 && rs_atomic_rank_ok(s) && (vl_bits || rs_sweep_records_ok(bbase, nsym));
        // (the fused form writes its records with the same kernel where it applies: one sweep that keeps every bucket — it beats the
        //  generated pass of radix_gen_records, which carries whole records through the LDS: 4 GiB UTF-8 18.8 against 21.4 ms)
        const bool sweep_fused = fuse_rec && tile_bytes && ix.sweep_records && (vl_bits || rs_sweep_records_ok(bbase, nsym));
        // (variable-length keys exist in the sweep kernels only: build_suffix_array redoes the build with dense keys)
        const char* vl_retry = "variable-length keys: the sweep form does not apply (retry with dense keys)";
        if ((vl_bits || part_m > 1) && !sweep_rec && !sweep_fused) throw RetryWithDenseKeys(vl_retry);
        // partition + gather with packed output: the partitioned entries ARE stored packed from the start (E = low words,
        // sa_hi_buf = bits 32..39): the gather reads them through Sa40, the last pass of every group writes the finished
        // entries back in that form — 16 GiB of text: 80 instead of 128 GiB of suffix array, during the build and after it
        const bool pack_E = !fuse_rec && want_pack && packed && ix.segmented_sort && rs_atomic_rank_ok(s);
        if (pack_E) {
            E.alloc(n * sizeof(uint32_t));
            sa_hi_buf.alloc(n);
        } else if (!fuse_rec) {
            E.alloc(n * sizeof(V));
        }
        st.alloc_ms += now_ms() - ta;
        st.fused_records = fuse_rec ? 1 : 0;
        DevBuf d_slotmap;
        if (fuse_rec || sweep_rec) {
            d_slotmap.alloc(260);
            CDB_HIP(hipMemcpyAsync(d_slotmap.p, h_slotmap, 260, hipMemcpyHostToDevice, s));
            CDB_HIP(hipStreamSynchronize(s));  // (h_slotmap is a stack array)
        }
        auto partition_entries = [&]() {
            gen.first_only = true;
            gen.symmap = d_symmap_first.as<uint16_t>();
            const int fbits = std::max(1, bit_width64((uint64_t)sigma - 1));
            if (pack_E) gen.vout_hi = sa_hi_buf.as<uint8_t>();
            (void)radix_sort<uint64_t, V>(s, ix.rws, ix.prof, (uint64_t*)nullptr, (uint64_t*)nullptr, (V*)nullptr, E.as<V>(), n, 0, fbits,
                                          &ss, ix.sort_variant, fbits, first_digit.data(), &gen);
            radix_check_error(s, ix.rws);  // (the gathers below read the text through these entries)
        };
        if (!fuse_rec && !sweep_rec) {
            // partition + gather reads no per-tile byte counts: they (1 KiB per 8 Ki positions: 2 GiB at 16 GiB of text) and the
            // tile-base workspace go back to the pool BEFORE the record memory of the groups is sized
            d_tbc.release();
            ix.tbw.partial.release();
            ix.tbw.blockbase.release();
            ix.tbw.base.release();
            partition_entries();
        }
        if (root_folded) {
            uint64_t at = 0;
            for (int k = 0; k < sigma; ++k) {
                folded_roots.push_back(CompatBucket{(unsigned long long)at, (unsigned long long)(at + h_first[border[k]])});
                at += h_first[border[k]];
            }
            st.root_folded = 1;
            bool lo_class = false;
            for (int b = 0; b < 128; ++b) lo_class |= h_map[b] != 0;
            if (lo_class) st.compat_rotations += 1;  // (what the root rotation would have counted: both classes present)
            // one level down: first-symbol buckets that are radix nodes of the reference (more than chuck suffixes) and
            // hold children on both sides of 0x80 — their blocks are swapped by the segmented sort's last pass
            if (ix.fold_depth1) {
                DevBuf d_cls;
                d_cls.alloc((512 + 768) * sizeof(uint64_t));
                CDB_HIP(hipMemsetAsync(d_cls.p, 0, (512 + 768) * sizeof(uint64_t), s));
                int t = ix.prof.begin(s);
                if (!pair_hi_counted)  // (counted beside the bytes otherwise: sa_tile_bytecount_kernel<true>)
                hipLaunchKernelGGL(sa_pairclass_kernel, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(n, 256 * 16 * 4), 2048))),
                                   dim3(256), 0, s, text, n, d_cls.as<unsigned long long>());
                hipLaunchKernelGGL(sa_docend_class_kernel, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ceil_div(D, 256), 1024))),
                                   dim3(256), 0, s, text, doc_start, D, n, d_cls.as<unsigned long long>() + 512);
                ix.prof.end(t, "sa_pairclass", n + D * 18, s);
                std::vector<uint64_t> cls(512 + 768);
                CDB_HIP(hipMemcpyAsync(cls.data(), d_cls.p, cls.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
                uint64_t h_pair_hi[256];
                if (pair_hi_counted) CDB_HIP(hipMemcpyAsync(h_pair_hi, d_pair_hi.p, sizeof(h_pair_hi), hipMemcpyDeviceToHost, s));
                CDB_HIP(hipStreamSynchronize(s));
                if (pair_hi_counted) {
                    for (int b = 0; b < 256; ++b) {
                        cls[2 * b + 1] = h_pair_hi[b];
                        cls[2 * b] = h_counts[b] - h_pair_hi[b];
                    }
                    st.pairclass_fused = 1;
                }
                const uint64_t chuck1 = std::max<uint64_t>(4096, n / 256);
                depth1.assign((size_t)sigma, Depth1Fold{});
                for (int b = 0; b < 256; ++b) {
                    if (!h_map[b]) continue;
                    const uint64_t* pr = &cls[2 * b];
                    const uint64_t* co = &cls[512 + 3 * b];
                    Depth1Fold f;
                    f.nend = co[2];
                    f.nlow = pr[0] - co[0];
                    f.nhigh = pr[1] - co[1];
                    f.fold = h_counts[b] > chuck1 && f.nlow && f.nhigh && f.nend + f.nlow + f.nhigh == h_counts[b];
                    int slot = 0;
                    while (border[slot] != (int)h_map[b]) ++slot;
                    depth1[slot] = f;
                }
            }
        }
        st.bucket_low_digits = brecords && blow > 0 ? blow / 8 : 0;
        st.key_layout = brecords ? (packed ? 5 : (bwide ? 4 : (blow == 0 ? 1 : (blow == 8 ? 2 : 3)))) : 0;
        if (brecords) {
            // "the key ends inside the document" = key mod base == 0 (dense number) / its lowest bit clear (variable-length keys)
            const uint32_t fbase = vl_bits ? 2u : bbase * part_m;
            const uint64_t bmagic = (fbase & (fbase - 1u)) ? (uint64_t)(~0ull / fbase) + 1ull : 0ull;
            const int bpass = (int)ceil_div(bbits, 8);
            const int lowb = blow / 8;
            const int keyb = bwide ? 8 : 4;  // bytes of the key part of a record
            // non-empty buckets and their entry ranges
            const uint32_t nb = (uint32_t)sigma;  // (every code stands for a byte that occurs)
            std::vector<uint64_t> bstart(nb + 1, 0);  // entry ranges of the buckets, in bucket order
            for (uint32_t b = 0; b < nb; ++b) bstart[b + 1] = bstart[b] + h_first[border[b]];
            // where every bucket crosses the text chunks: 2 MiB, so that the chunk all concurrent work items read
            // fits every XCD's 4 MB L2
            const uint64_t chunk = 2ull << 20;  // (measured at 8 GiB: 1 MiB 61 ms, 2 MiB 60, 4 MiB 62, 16 MiB 88, 32 MiB 97)
            const uint32_t nch = (uint32_t)ceil_div(n, chunk);
            DevBuf d_bstart, d_bounds;
            d_bstart.alloc((nb + 1) * 8);
            d_bounds.alloc(fuse_rec ? 8 : (size_t)nb * (nch + 1) * 8);
            CDB_HIP(hipMemcpyAsync(d_bstart.p, bstart.data(), (nb + 1) * 8, hipMemcpyHostToDevice, s));
            auto bucket_bounds = [&]() {
            if (pack_E)
            hipLaunchKernelGGL((sa_bucket_bounds_kernel<Packed40>), dim3((unsigned)ceil_div((uint64_t)nb * (nch + 1), 256)), dim3(256), 0, s,
                               Sa40{E.as<uint32_t>(), sa_hi_buf.as<uint8_t>()}, (const unsigned long long*)d_bstart.as<unsigned long long>(), nb,
                               nch, chunk, doc_start, (int)ix.bits, ix.mask, d_bounds.as<unsigned long long>());
            else if (!fuse_rec)
            hipLaunchKernelGGL((sa_bucket_bounds_kernel<V>), dim3((unsigned)ceil_div((uint64_t)nb * (nch + 1), 256)), dim3(256), 0, s,
                               (const V*)E.as<V>(), (const unsigned long long*)d_bstart.as<unsigned long long>(), nb, nch, chunk,
                               doc_start, (int)ix.bits, ix.mask, d_bounds.as<unsigned long long>());
            };
            if (!sweep_rec) bucket_bounds();
            // sweep form: array-wide tile bases of every bucket slot (4 GiB of them for 16 GiB of text: allocated BEFORE the record
            // memory is sized; the counts they were made from go back to the pool), the document of every tile's first position
            DevBuf sweep_doc, d_codeslot;
            const unsigned long long* sweep_base = nullptr;
            if (sweep_rec || sweep_fused) {
                std::vector<uint16_t> codeslot(256, 0);  // byte -> symbol code | bucket slot << 8 (one lookup per text byte)
                for (int b = 0; b < 256; ++b) codeslot[b] = (uint16_t)(h_map[b] | ((uint32_t)h_slotmap[h_map[b]] << 8));
                d_codeslot.alloc(256 * sizeof(uint16_t));
                CDB_HIP(hipMemcpyAsync(d_codeslot.p, codeslot.data(), 256 * sizeof(uint16_t), hipMemcpyHostToDevice, s));
                std::vector<uint64_t> slot_start(256, n);
                for (uint32_t b = 0; b < nb; ++b) slot_start[b] = bstart[b];
                DevBuf d_slot_start;
                d_slot_start.alloc(256 * sizeof(uint64_t));
                CDB_HIP(hipMemcpyAsync(d_slot_start.p, slot_start.data(), 256 * sizeof(uint64_t), hipMemcpyHostToDevice, s));
                int t = ix.prof.begin(s);
                sweep_base = rs_tile_bases(s, ix.tbw, d_tbc.as<uint32_t>(), tiles8, d_src_col.as<uint16_t>(),
                                           (const unsigned long long*)d_slot_start.as<unsigned long long>());
                sweep_doc.alloc(((size_t)tiles8 + 1) * sizeof(uint64_t));
                hipLaunchKernelGGL(rs_tiledoc_kernel, dim3((unsigned)ceil_div((uint64_t)tiles8 + 1, 256)), dim3(256), 0, s, doc_start, D, n,
                                   (uint64_t)RS_SWEEP_TILE, (uint64_t)tiles8, sweep_doc.as<uint64_t>());
                ix.prof.end(t, "rs_tile_bases", (uint64_t)tiles8 * 256 * (4 + 8), s);
                CDB_HIP(hipStreamSynchronize(s));  // (slot_start, d_slot_start, codeslot)
                d_tbc.release();
                ix.tbw.partial.release();
                ix.tbw.blockbase.release();
            }
            // groups of consecutive buckets whose records (4 + lowb bytes per suffix) fit the memory left
            size_t fre = 0, tot = 0;
            CDB_HIP(hipMemGetInfo(&fre, &tot));
            const double avail = (double)fre + (double)DevPool::get().cached_bytes();
            const int auxb = packed ? (lowb == 0 ? 1 : (lowb == 1 ? 2 : 4)) : lowb;  // bytes of the auxiliary array per suffix
            const int recb = keyb + auxb + (packed ? 4 : 0);                         // record bytes per suffix of a group
            // ---- segmented form (round 3): ONE launch per radix pass sorts every bucket of a group (radix_sort.h:
            // radix_sort_segmented), the last pass writes entries + group flags itself, the gather is planned on the
            // device and XCD-aware, and nothing in the group loop waits for the host.  Needs two record buffers per
            // group (the passes ping-pong over the whole group) and a bucket is never split across groups.
            uint64_t seg_cap = 0;
            if (packed && ix.segmented_sort && sizeof(V) == 8 && rs_atomic_rank_ok(s)) {
                const double per = 2.0 * recb + 1.0;  // + edge records (32 B per tile and digit)
                uint64_t cap = (uint64_t)std::max(0.0, avail * 0.85 / per);
                if (cap >= maxb) {
                    if (ix.bucket_group_limit) cap = std::max<uint64_t>(std::min<uint64_t>(cap, ix.bucket_group_limit), maxb);
                    seg_cap = std::min<uint64_t>(cap, n);
                }
                if (fuse_rec) seg_cap = n;  // (decided with the same bound before the entries were NOT partitioned)
            }
            if (!fuse_rec && ix.debug_no_segcap) seg_cap = 0;  // (test hook: "a bucket does not fit the record memory")
            if (fuse_rec && !seg_cap) throw Error("bucket-wise build: fused records without the segmented sort (internal)");
            if ((vl_bits || part_m > 1) && !seg_cap) throw RetryWithDenseKeys(vl_retry);
            if (sweep_rec && !seg_cap) {  // (a bucket larger than the record memory: partition + gather, bucket by bucket)
                sweep_rec = false;
                sweep_doc.release();
                ix.tbw.base.release();
                partition_entries();
                bucket_bounds();
            }
            if (pack_E && !seg_cap) {
                // (a bucket larger than the record memory: the per-bucket forms below work on plain 8-byte entries)
                if constexpr (sizeof(V) == 8) {
                    DevBuf wide;
                    wide.alloc(n * sizeof(uint64_t));
                    Index tmp_view;  // (sa_expand wants an Index: borrow the two arrays for one call)
                    tmp_view.stream = s;
                    tmp_view.d_sa = std::move(E);
                    tmp_view.d_sa_hi = std::move(sa_hi_buf);
                    tmp_view.sa_packed = true;
                    sa_expand(tmp_view, 0, n, wide.as<uint64_t>());
                    CDB_HIP(hipStreamSynchronize(s));
                    tmp_view.stream = nullptr;
                    E = std::move(wide);
                }
            }
            const bool pack_seg = pack_E && seg_cap;
            auto run_segmented = [&](auto wtag) {
                using W = decltype(wtag);
                if constexpr (sizeof(V) == 8) {
                    struct Group {
                        uint32_t b0, b1, tiles, cells;
                        uint64_t gstart, elems;
                        GatherPlan plan;
                    };
                    std::vector<Group> groups;
                    std::vector<SegInfo> h_segs(nb);
                    uint32_t max_tiles = 0, max_gb = 0, max_cells = 0;
                    uint64_t max_items = 0;
                    for (uint32_t b0 = 0; b0 < nb;) {
                        uint32_t b1 = b0 + 1;
                        while (b1 < nb && bstart[b1 + 1] - bstart[b0] <= seg_cap) ++b1;
                        Group g;
                        g.b0 = b0;
                        g.b1 = b1;
                        g.gstart = bstart[b0];
                        g.elems = bstart[b1] - bstart[b0];
                        uint32_t tiles = 0;
                        for (uint32_t b = b0; b < b1; ++b) {
                            h_segs[b] = SegInfo{(unsigned long long)(bstart[b] - g.gstart), (unsigned long long)(bstart[b + 1] - g.gstart), tiles, 0u, 0ull, 0ull};
                            if (!depth1.empty() && depth1[b].fold) {  // [end][low][high] is written as [end][high][low]
                                h_segs[b].rot_a = depth1[b].nend;
                                h_segs[b].rot_b = depth1[b].nend + depth1[b].nlow;
                                depth1[b].done = true;
                                st.compat_rotations += 1;
                            }
                            tiles += (uint32_t)ceil_div(bstart[b + 1] - bstart[b], (uint64_t)RS_SEG_TILE);
                        }
                        g.tiles = tiles;
                        const uint32_t gb = b1 - b0;
                        g.plan.cell_base[0] = 0;
                        for (uint32_t x = 0; x < 8; ++x) g.plan.cell_base[x + 1] = g.plan.cell_base[x] + (x < nch ? (nch - x + 7) / 8 : 0u) * gb;
                        g.cells = g.plan.cell_base[8];
                        max_tiles = std::max(max_tiles, tiles);
                        max_gb = std::max(max_gb, gb);
                        max_cells = std::max(max_cells, g.cells);
                        max_items = std::max<uint64_t>(max_items, g.elems / BR_ITEM + g.cells + 8);
                        groups.push_back(g);
                        b0 = b1;
                    }
                    uint64_t max_elems = 0;
                    for (const Group& g : groups) max_elems = std::max(max_elems, g.elems);
                    DevBuf kb[2], eb[2], wb[2], edges, d_starts, d_segs, tile_seg, cell_off, lists, d_items2, d_bh2;
                    uint32_t *kbp[2], *ebp[2];
                    // fused form: key and entry halves of a record buffer are ONE block, so that the last pass can write the
                    // finished 8-byte entries over the buffer it does not read (the passes ping-pong: pass p reads buffer p & 1)
                    // — that block then IS the suffix array, and no separate 8 n bytes are ever allocated
                    const int dead = bpass & 1;
                    for (int q = 0; q < 2; ++q) {
                        if (fuse_rec) {
                            kb[q] = std::move(fr_kv[q]);  // (allocated when the fused form was chosen: max_elems = n)
                            kbp[q] = kb[q].as<uint32_t>();
                            if (packed_out) {
                                eb[q] = std::move(fr_e[q]);
                                ebp[q] = eb[q].as<uint32_t>();
                            } else {
                                ebp[q] = kbp[q] + max_elems;
                            }
                            wb[q] = std::move(fr_w[q]);
                            continue;
                        } else {
                            kb[q].alloc(max_elems * sizeof(uint32_t));
                            eb[q].alloc(max_elems * sizeof(uint32_t));
                            kbp[q] = kb[q].as<uint32_t>();
                            ebp[q] = eb[q].as<uint32_t>();
                        }
                        wb[q].alloc(max_elems * sizeof(W));
                    }
                    edges.alloc((size_t)max_tiles * 256 * sizeof(SegEdge));
                    d_starts.alloc((size_t)max_gb * 8 * 256 * sizeof(uint64_t));
                    d_bh2.alloc((size_t)max_gb * 8 * 256 * sizeof(uint64_t));
                    d_segs.alloc((size_t)nb * sizeof(SegInfo));
                    tile_seg.alloc((size_t)max_tiles * sizeof(uint32_t));
                    cell_off.alloc((size_t)max_cells * sizeof(uint32_t));
                    lists.alloc(groups.size() * 16 * sizeof(uint32_t));  // per group: 8 list lengths, 8 tickets
                    d_items2.alloc((size_t)max_items * sizeof(BucketItem));
                    st.alloc_ms += now_ms() - ta;
                    CDB_HIP(hipMemcpyAsync(d_segs.p, h_segs.data(), (size_t)nb * sizeof(SegInfo), hipMemcpyHostToDevice, s));
                    CDB_HIP(hipMemsetAsync(lists.p, 0, groups.size() * 16 * sizeof(uint32_t), s));
                    const unsigned long long* bnd = d_bounds.as<unsigned long long>();
                    for (size_t gi = 0; gi < groups.size(); ++gi) {
                        const Group& g = groups[gi];
                        const uint32_t gb = g.b1 - g.b0;
                        uint32_t* list_len = lists.as<uint32_t>() + gi * 16;
                        uint32_t* tickets = list_len + 8;
                        if (fuse_rec && !sweep_fused) {
                            {
                                hipLaunchKernelGGL(rs_seg_tilemap_kernel, dim3((unsigned)ceil_div(g.tiles, 256)), dim3(256), 0, s,
                                                   (const SegInfo*)(d_segs.as<SegInfo>() + g.b0), gb, g.tiles, tile_seg.as<uint32_t>());
                                TextGen rg{text, doc_start, d_symmap.as<uint16_t>(), D, (int)ix.bits, bbase, nsym, 0, ix.text_padded};
                                rg.slotmap = d_slotmap.as<uint8_t>();
                                rg.rec_low_bits = blow;
                                radix_gen_records<W>(s, ix.rws, ix.prof, kbp[0], ebp[0], wb[0].as<W>(), n,
                                                     first_digit.data(), rg, (const uint32_t*)tile_seg.as<uint32_t>(),
                                                     (const SegInfo*)(d_segs.as<SegInfo>() + g.b0), gb, g.tiles, lowb, bpass,
                                                     d_bh2.as<unsigned long long>(), &ss,
                                                     tile_bytes ? (const uint32_t*)d_tbc.as<uint32_t>() : nullptr,
                                                     tile_bytes ? (const uint16_t*)d_src_col.as<uint16_t>() : nullptr, tile_bytes ? &ix.tbw : nullptr);
                                st.gen_prebased = tile_bytes ? 1 : 0;
                            }
                        } else if (sweep_rec || sweep_fused) {
                            hipLaunchKernelGGL(rs_seg_tilemap_kernel, dim3((unsigned)ceil_div(g.tiles, 256)), dim3(256), 0, s,
                                               (const SegInfo*)(d_segs.as<SegInfo>() + g.b0), gb, g.tiles, tile_seg.as<uint32_t>());
                            TextGen rg{text, doc_start, d_symmap.as<uint16_t>(), D, (int)ix.bits, bbase, nsym, 0, ix.text_padded};
                            rg.slotmap = d_slotmap.as<uint8_t>();
                            rg.rec_low_bits = blow;
                            rg.part_m = part_m;
                            rg.part_r = part_gen.part_r;
                            rg.part_s = part_gen.part_s;
                            rg.tile_doc = sweep_doc.as<uint64_t>();
                            rg.tile_base = sweep_base;
                            if (vl_bits) {
                                VlTables vt;
                                vt.sym = d_vl_sym.as<uint16_t>();
                                vt.dec = d_vl_dec.as<uint16_t>();
                                vt.key_bits = vl_bits;
                                vt.end_len = vlc.end_len;
                                radix_sweep_records_vl<W>(s, ix.prof, kbp[0], ebp[0], wb[0].as<W>(), n, rg, (const uint16_t*)d_codeslot.as<uint16_t>(), vt,
                                                          g.b0, g.b1, g.gstart, g.elems, (const uint32_t*)tile_seg.as<uint32_t>(),
                                                          (const SegInfo*)(d_segs.as<SegInfo>() + g.b0), gb, g.tiles, lowb, bpass,
                                                          d_bh2.as<unsigned long long>(), &ss);
                            } else
                            radix_sweep_records<W>(s, ix.prof, kbp[0], ebp[0], wb[0].as<W>(), n, rg, (const uint16_t*)d_codeslot.as<uint16_t>(), g.b0, g.b1,
                                                   g.gstart, g.elems,
                                                   (const uint32_t*)tile_seg.as<uint32_t>(), (const SegInfo*)(d_segs.as<SegInfo>() + g.b0), gb,
                                                   g.tiles, lowb, bpass, d_bh2.as<unsigned long long>(), &ss);
                            st.sweep_records = 1;
                            st.gen_prebased = 1;
                        } else {
                        CDB_HIP(hipMemsetAsync(d_bh2.p, 0, (size_t)gb * 8 * 256 * sizeof(uint64_t), s));
                        int t = ix.prof.begin(s);
                        hipLaunchKernelGGL(sa_gather_plan_kernel, dim3(8), dim3(1024), 0, s, bnd, nch, g.b0, gb, g.plan, cell_off.as<uint32_t>(),
                                           list_len);
                        hipLaunchKernelGGL(sa_gather_emit_kernel, dim3((unsigned)ceil_div(g.cells, 256)), dim3(256), 0, s, bnd, nch, g.b0, gb,
                                           g.plan, (const uint32_t*)cell_off.as<uint32_t>(), (const uint32_t*)list_len,
                                           d_items2.as<BucketItem>());
                        ix.prof.end(t, "sa_gather_plan", (uint64_t)g.cells * 24 + (g.elems / BR_ITEM) * sizeof(BucketItem), s);
                        t = ix.prof.begin(s);
                        constexpr unsigned gather_wgs = 256u * 8u;
                        if (pack_seg)
                        hipLaunchKernelGGL((sa_bucket_records_lists_kernel<W, Packed40>), dim3(gather_wgs), dim3(256), 0, s,
                                           Sa40{E.as<uint32_t>(), sa_hi_buf.as<uint8_t>()},
                                           (const BucketItem*)d_items2.as<BucketItem>(), (const uint32_t*)list_len, tickets, text, n, doc_start,
                                           (const uint16_t*)d_symmap.as<uint16_t>(), (int)ix.bits, ix.mask, nsym, bbase, blow, bpass, g.gstart,
                                           g.b0, kbp[0], wb[0].as<W>(), ebp[0], d_bh2.as<unsigned long long>());
                        else
                        hipLaunchKernelGGL((sa_bucket_records_lists_kernel<W>), dim3(gather_wgs), dim3(256), 0, s, (const uint64_t*)E.as<uint64_t>(),
                                           (const BucketItem*)d_items2.as<BucketItem>(), (const uint32_t*)list_len, tickets, text, n, doc_start,
                                           (const uint16_t*)d_symmap.as<uint16_t>(), (int)ix.bits, ix.mask, nsym, bbase, blow, bpass, g.gstart,
                                           g.b0, kbp[0], wb[0].as<W>(), ebp[0], d_bh2.as<unsigned long long>());
                        ix.prof.end(t, "sa_bucket_records", g.elems * ((uint64_t)nsym + recb + sizeof(V)), s);
                        st.gather_items += g.elems / BR_ITEM;
                        hipLaunchKernelGGL(rs_seg_tilemap_kernel, dim3((unsigned)ceil_div(g.tiles, 256)), dim3(256), 0, s,
                                           (const SegInfo*)(d_segs.as<SegInfo>() + g.b0), gb, g.tiles, tile_seg.as<uint32_t>());
                        }
                        SegFinalArgs fin;
                        if (pack_seg) {
                            fin.elo = E.as<uint32_t>() + g.gstart;
                            fin.ehi = sa_hi_buf.as<uint8_t>() + g.gstart;
                        } else if (packed_out) {
                            fin.elo = kbp[dead] + g.gstart;
                            fin.ehi = sa_hi_buf.as<uint8_t>() + g.gstart;
                        } else {
                            fin.eout = (fuse_rec ? kb[dead].as<uint64_t>() : E.as<uint64_t>()) + g.gstart;
                        }
                        fin.flags = flags.as<uint8_t>() + g.gstart;
                        fin.edges = edges.as<SegEdge>();
                        fin.hi_shift = blow;
                        fin.low_bits = blow;
                        fin.kbase = fbase;
                        fin.kmagic = bmagic;
                        radix_sort_segmented<W>(s, ix.rws, ix.prof, kbp[0], kbp[1], ebp[0], ebp[1], wb[0].as<W>(), wb[1].as<W>(), g.elems,
                                                (const SegInfo*)(d_segs.as<SegInfo>() + g.b0), (const uint32_t*)tile_seg.as<uint32_t>(), gb,
                                                g.tiles, (const unsigned long long*)d_bh2.as<unsigned long long>(),
                                                d_starts.as<unsigned long long>(), bbits - blow, lowb, fin, &ss);
                        st.bucket_groups++;
                    }
                    CDB_HIP(hipStreamSynchronize(s));  // (h_segs and the group scratch go out of scope)
                    if (sweep_rec || sweep_fused) ix.tbw.base.release();
                    if (fuse_rec) E = std::move(kb[dead]);
                    if (pack_seg) packed_out = true;
                    st.segmented = 1;
                }
            };
            if (seg_cap) {
                if (lowb == 0) run_segmented(uint8_t{});
                else if (lowb == 1) run_segmented(uint16_t{});
                else run_segmented(uint32_t{});
            } else {
            std::vector<uint64_t> bounds((size_t)nb * (nch + 1));
            CDB_HIP(hipMemcpyAsync(bounds.data(), d_bounds.p, bounds.size() * 8, hipMemcpyDeviceToHost, s));
            CDB_HIP(hipStreamSynchronize(s));
            // (the third entry buffer is a luxury: without it an odd number of passes costs a copy back)
            const bool third = !packed && avail > (double)maxb * (2.0 * (keyb + lowb) + 2 * sizeof(V)) * 1.15;
            const double scratch = packed ? (double)maxb * recb
                                          : (double)maxb * (keyb + lowb + (third ? 2 : 1) * sizeof(V)) +
                                                (third ? (double)(n / 32) * (sizeof(I) + 16 + 2 * sizeof(V) + 1) : 0.0);
            uint64_t gcap = (uint64_t)std::max(0.0, (avail - scratch) * 0.85 / recb);
            if (ix.bucket_group_limit) gcap = std::min<uint64_t>(gcap, ix.bucket_group_limit);
            gcap = std::max<uint64_t>(std::min<uint64_t>(gcap, n), maxb);
            DevBuf k32g, lowg, k32t, lowt, ET, EX, elog, elot, d_bh, d_items;
            k32g.alloc(gcap * keyb);
            if (auxb) lowg.alloc(gcap * auxb);
            k32t.alloc(maxb * keyb);
            if (auxb) lowt.alloc(maxb * auxb);
            if (packed) {
                elog.alloc(gcap * sizeof(uint32_t));
                elot.alloc(maxb * sizeof(uint32_t));
            } else {
                ET.alloc(maxb * sizeof(V));
                if (third) EX.alloc(maxb * sizeof(V));  // third entry buffer: an odd number of passes still ends in place (radix_sort.h)
            }
            std::vector<uint64_t> bh;
            std::vector<BucketItem> items;
            auto run_group = [&](auto wtag, auto ktag, uint32_t b0, uint32_t b1) {  // buckets [b0, b1)
                using W = decltype(wtag);
                using K = decltype(ktag);
                constexpr bool HAS_W = !std::is_same<W, NoVal>::value;
                using FW = typename std::conditional<HAS_W, W, uint8_t>::type;
                const uint64_t gstart = bstart[b0];
                const uint32_t gb = b1 - b0;
                d_bh.ensure((size_t)gb * 8 * 256 * sizeof(uint64_t));
                CDB_HIP(hipMemsetAsync(d_bh.p, 0, (size_t)gb * 8 * 256 * sizeof(uint64_t), s));
                items.clear();
                for (uint32_t x = 0; x < nch; ++x)      // text chunk major: the group's buckets share the cached text
                    for (uint32_t b = b0; b < b1; ++b) {
                        const uint64_t lo = bounds[(size_t)b * (nch + 1) + x], hi = bounds[(size_t)b * (nch + 1) + x + 1];
                        for (uint64_t o = lo; o < hi; o += BR_ITEM)
                            items.push_back(BucketItem{(unsigned long long)o, (uint32_t)std::min<uint64_t>(BR_ITEM, hi - o), b});
                    }
                st.gather_items += items.size();
                if (!items.empty()) {
                    d_items.ensure(items.size() * sizeof(BucketItem));
                    CDB_HIP(hipMemcpyAsync(d_items.p, items.data(), items.size() * sizeof(BucketItem), hipMemcpyHostToDevice, s));
                    int t = ix.prof.begin(s);
                    hipLaunchKernelGGL((sa_bucket_records_kernel<V, W, K>), dim3((unsigned)items.size()), dim3(256), 0, s,
                                       (const V*)E.as<V>(), (const BucketItem*)d_items.as<BucketItem>(), text, n, doc_start,
                                       (const uint16_t*)d_symmap.as<uint16_t>(), (int)ix.bits, ix.mask, nsym, bbase, blow, bpass,
                                       gstart, b0, k32g.as<K>(), lowg.as<W>(), d_bh.as<unsigned long long>());
                    ix.prof.end(t, "sa_bucket_records", (bstart[b1] - gstart) * ((uint64_t)nsym + keyb + lowb + sizeof(V)), s);
                }
                // (the per-bucket digit histograms stay on the device: the sorts below are queued without a single host
                //  round trip; this one synchronisation per GROUP only protects `items`, rebuilt for the next group)
                CDB_HIP(hipStreamSynchronize(s));
                for (uint32_t b = b0; b < b1; ++b) {
                    const uint64_t start = bstart[b], cnt = bstart[b + 1] - start;
                    V* eb = E.as<V>() + start;
                    K* kb = k32g.as<K>() + (start - gstart);
                    FW* lb = HAS_W ? lowg.as<FW>() + (start - gstart) : (FW*)nullptr;
                    int r = 0;
                    if (bpass > 0 && cnt > 1) {
                        const unsigned long long* hb = d_bh.as<unsigned long long>() + (size_t)(b - b0) * 8 * 256;
                        ix.rws.value_spare = EX.p;  // (null without the third buffer)
                        if constexpr (HAS_W)
                            r = radix_sort_split<V, W>(s, ix.rws, ix.prof, kb, k32t.as<uint32_t>(), eb, ET.as<V>(), lb, lowt.as<W>(), cnt,
                                                       bbits - blow, &ss, ix.sort_variant, 8, (const uint64_t*)nullptr,
                                                       (const TextGen*)nullptr, 0, hb);
                        else
                            r = radix_sort<K, V>(s, ix.rws, ix.prof, kb, k32t.as<K>(), eb, ET.as<V>(), cnt, 0, bbits, &ss,
                                                 ix.sort_variant, 8, (const uint64_t*)nullptr, (const TextGen*)nullptr, hb);
                        if (ix.rws.value_result == 1) CDB_HIP(hipMemcpyAsync(eb, ET.p, cnt * sizeof(V), hipMemcpyDeviceToDevice, s));
                    }
                    if constexpr (sizeof(K) == 8)
                        hipLaunchKernelGGL(sa_initflags_kernel, dim3((unsigned)ceil_div(cnt, 1024)), dim3(256), 0, s,
                                           (const uint64_t*)(r ? k32t.as<uint64_t>() : (uint64_t*)kb), cnt, bbase, bmagic,
                                           flags.as<uint8_t>() + start, (start & 3) == 0);
                    else
                        hipLaunchKernelGGL(sa_initflags32_kernel<FW>, dim3((unsigned)ceil_div(cnt, 1024)), dim3(256), 0, s,
                                           (const uint32_t*)(r ? k32t.as<uint32_t>() : (uint32_t*)kb),
                                           HAS_W ? (const FW*)(r ? lowt.as<FW>() : lb) : (const FW*)nullptr, blow, cnt, bbase, bmagic,
                                           flags.as<uint8_t>() + start, (start & 3) == 0);
                }
            };
            auto run_group_packed = [&](auto wtag, uint32_t b0, uint32_t b1) {  // buckets [b0, b1), packed entries
                using W = decltype(wtag);
                if constexpr (sizeof(V) == 8) {
                    const uint64_t gstart = bstart[b0];
                    const uint32_t gb = b1 - b0;
                    d_bh.ensure((size_t)gb * 8 * 256 * sizeof(uint64_t));
                    CDB_HIP(hipMemsetAsync(d_bh.p, 0, (size_t)gb * 8 * 256 * sizeof(uint64_t), s));
                    items.clear();
                    for (uint32_t x = 0; x < nch; ++x)      // text chunk major: the group's buckets share the cached text
                        for (uint32_t b = b0; b < b1; ++b) {
                            const uint64_t lo = bounds[(size_t)b * (nch + 1) + x], hi = bounds[(size_t)b * (nch + 1) + x + 1];
                            for (uint64_t o = lo; o < hi; o += BR_ITEM)
                                items.push_back(BucketItem{(unsigned long long)o, (uint32_t)std::min<uint64_t>(BR_ITEM, hi - o), b});
                        }
                    st.gather_items += items.size();
                    if (!items.empty()) {
                        d_items.ensure(items.size() * sizeof(BucketItem));
                        CDB_HIP(hipMemcpyAsync(d_items.p, items.data(), items.size() * sizeof(BucketItem), hipMemcpyHostToDevice, s));
                        int t = ix.prof.begin(s);
                        hipLaunchKernelGGL((sa_bucket_records_packed_kernel<W>), dim3((unsigned)items.size()), dim3(256), 0, s,
                                           (const uint64_t*)E.as<uint64_t>(), (const BucketItem*)d_items.as<BucketItem>(), text, n,
                                           doc_start, (const uint16_t*)d_symmap.as<uint16_t>(), (int)ix.bits, ix.mask, nsym, bbase,
                                           blow, bpass, gstart, b0, k32g.as<uint32_t>(), lowg.as<W>(), elog.as<uint32_t>(),
                                           d_bh.as<unsigned long long>());
                        ix.prof.end(t, "sa_bucket_records", (bstart[b1] - gstart) * ((uint64_t)nsym + recb + sizeof(V)), s);
                    }
                    CDB_HIP(hipStreamSynchronize(s));  // (protects `items`, rebuilt for the next group)
                    for (uint32_t b = b0; b < b1; ++b) {
                        const uint64_t start = bstart[b], cnt = bstart[b + 1] - start;
                        uint32_t* kb = k32g.as<uint32_t>() + (start - gstart);
                        uint32_t* vb = elog.as<uint32_t>() + (start - gstart);
                        W* lb = lowg.as<W>() + (start - gstart);
                        int r = 0, vr = 0;
                        if (bpass > 0 && cnt > 1) {
                            const unsigned long long* hb = d_bh.as<unsigned long long>() + (size_t)(b - b0) * 8 * 256;
                            ix.rws.value_spare = nullptr;
                            r = radix_sort_split<uint32_t, W>(s, ix.rws, ix.prof, kb, k32t.as<uint32_t>(), vb, elot.as<uint32_t>(), lb,
                                                              lowt.as<W>(), cnt, bbits - blow, &ss, ix.sort_variant, 8,
                                                              (const uint64_t*)nullptr, (const TextGen*)nullptr, 0, hb, lowb);
                            vr = ix.rws.value_result;
                        }
                        const W* ls = r ? lowt.as<W>() : lb;
                        int t = ix.prof.begin(s);
                        hipLaunchKernelGGL(sa_assemble_entries_kernel<W>, dim3((unsigned)ceil_div(cnt, 256)), dim3(256), 0, s,
                                           (const uint32_t*)(vr ? elot.as<uint32_t>() : vb), ls, blow, cnt, E.as<uint64_t>() + start);
                        ix.prof.end(t, "sa_assemble_entries", cnt * (4 + sizeof(W) + 8), s);
                        hipLaunchKernelGGL(sa_initflags32_kernel<W>, dim3((unsigned)ceil_div(cnt, 1024)), dim3(256), 0, s,
                                           (const uint32_t*)(r ? k32t.as<uint32_t>() : kb), lowb ? ls : (const W*)nullptr, blow, cnt,
                                           bbase, bmagic, flags.as<uint8_t>() + start, (start & 3) == 0);
                    }
                }
            };
            for (uint32_t b0 = 0; b0 < nb;) {
                uint32_t b1 = b0 + 1;
                while (b1 < nb && bstart[b1 + 1] - bstart[b0] <= gcap) ++b1;
                if (packed && lowb == 0) run_group_packed(uint8_t{}, b0, b1);
                else if (packed && lowb == 1) run_group_packed(uint16_t{}, b0, b1);
                else if (packed) run_group_packed(uint32_t{}, b0, b1);
                else if (bwide) run_group(NoVal{}, uint64_t{}, b0, b1);
                else if (blow == 0) run_group(NoVal{}, uint32_t{}, b0, b1);
                else if (blow == 8) run_group(uint8_t{}, uint32_t{}, b0, b1);
                else run_group(uint16_t{}, uint32_t{}, b0, b1);
                st.bucket_groups++;
                b0 = b1;
            }
            }  // (per-bucket launches)
        } else {
        KT[0].alloc(maxb * sizeof(uint64_t));
        if (nsym > 1 && maxb > 1) {
            KT[1].alloc(maxb * sizeof(uint64_t));
            ET.alloc(maxb * sizeof(V));
        }
        uint64_t start = 0;
        for (int k = 0; k < sigma; ++k) {
            const uint64_t cnt = h_first[border[k]];
            if (!cnt) continue;
            V* eb = E.as<V>() + start;
            int t = ix.prof.begin(s);
            hipLaunchKernelGGL((sa_bucket_keys_kernel<V>), dim3((unsigned)ceil_div(cnt, 256)), dim3(256), 0, s, (const V*)eb, cnt,
                               text, n, doc_start, (const uint16_t*)d_symmap.as<uint16_t>(), (int)ix.bits, ix.mask, nsym, symbits,
                               KT[0].as<uint64_t>());
            ix.prof.end(t, "sa_bucket_keys", cnt * ((uint64_t)nsym + 8 + sizeof(V)), s);
            int r = 0;
            if (nsym > 1 && cnt > 1) {
                r = radix_sort<uint64_t, V>(s, ix.rws, ix.prof, KT[0].as<uint64_t>(), KT[1].as<uint64_t>(), eb, ET.as<V>(), cnt, 0,
                                            top_shift, &ss, ix.sort_variant, dbits);
                if (r == 1) CDB_HIP(hipMemcpyAsync(eb, ET.p, cnt * sizeof(V), hipMemcpyDeviceToDevice, s));
            }
            hipLaunchKernelGGL(sa_initflags_kernel, dim3((unsigned)ceil_div(cnt, 1024)), dim3(256), 0, s,
                               (const uint64_t*)KT[r].as<uint64_t>(), cnt, kbase, kmagic, flags.as<uint8_t>() + start,
                               (start & 3) == 0);
            start += cnt;
        }
        }
        CDB_HIP(hipStreamSynchronize(s));
        st.bucketed = 1;
        sa_buf = std::move(E);
    }
    bool tile_sums_ready = flags_by_sort;  // the raw tile sums of the first compaction are in scan_partials already
    st.flags_in_last_pass = flags_by_sort ? 1 : 0;
    if (!big && !flags_by_sort) {  // (the bucket-wise sort writes the flags itself)
        int t = ix.prof.begin(s);
        if (layout == WIDE)
            hipLaunchKernelGGL(sa_initflags_kernel, dim3((unsigned)ceil_div(n, 1024)), dim3(256), 0, s,
                               (const uint64_t*)sorted_keys.as<uint64_t>(), n, kbase, kmagic, flags.as<uint8_t>(), true);
        else {
            // (the flag kernel leaves the per-tile counts of unresolved entries for the first compaction: no second sweep
            //  over the flags for them)
            const uint64_t nbt = ceil_div(n, (uint64_t)SC_TILE);
            ix.scan_partials.ensure(scan_partials_slots(nbt) * sizeof(U2));
            CDB_HIP(hipMemsetAsync(ix.scan_partials.p, 0, nbt * sizeof(U2), s));
            tile_sums_ready = true;
            if (layout == SPLIT2)
                hipLaunchKernelGGL(sa_initflags32_kernel<uint16_t>, dim3((unsigned)ceil_div(n, 1024)), dim3(256), 0, s,
                                   (const uint32_t*)sorted_k32.as<uint32_t>(), (const uint16_t*)sorted_low.as<uint16_t>(), low_bits,
                                   n, kbase, kmagic, flags.as<uint8_t>(), true, ix.scan_partials.as<U2>());
            else
                hipLaunchKernelGGL(sa_initflags32_kernel<uint8_t>, dim3((unsigned)ceil_div(n, 1024)), dim3(256), 0, s,
                                   (const uint32_t*)sorted_k32.as<uint32_t>(),
                                   layout == SPLIT ? (const uint8_t*)sorted_low.as<uint8_t>() : (const uint8_t*)nullptr, low_bits,
                                   n, kbase, kmagic, flags.as<uint8_t>(), true, ix.scan_partials.as<U2>());
        }
        ix.prof.end(t, "sa_initflags", n * (layout == WIDE ? 9 : 5 + low_bytes), s);
    }
    // (a starved pass of the initial sort left garbage entries: the refinement below would index the text with them —
    //  the error surfaces HERE, before anything dereferences an entry; build_suffix_array redoes the build in plain order)
    radix_check_error(s, ix.rws);
    ta = now_ms();
    // The sorted keys stay valid for the finished array: refinement only permutes entries inside groups
    // of equal keys.  They let a search probe decide on ONE load (query.hip) — kept when affordable.
    ix.drop_keys();
    if (ix.keep_keys && !big && n * 8 <= (16ull << 30)) {
        ix.d_keys = std::move(sorted_keys);
        ix.d_keys32 = std::move(sorted_k32);
        ix.d_keylow = std::move(sorted_low);
        ix.key_low_bits = low_bits;
        ix.key_low_bytes = low_bytes;
        ix.key_nsym = nsym;
        ix.key_base = kbase;
    } else {
        sorted_keys.release();
        sorted_k32.release();
        sorted_low.release();
    }
    st.free_ms += now_ms() - ta;

    // ---- 4. refinement rounds
    // slot order of the array as the bucket-wise build laid it out -> plain unsigned order (ranks of prefix doubling)
    DevBuf d_order;
    SlotOrder order;
    if (!folded_roots.empty()) {
        struct Blk { unsigned long long a, len; int code, cls; };  // cls: 0 = end of document, 1 = bytes < 0x80, 2 = bytes >= 0x80
        std::vector<Blk> blks;
        std::vector<int> code_of_root;  // first-symbol code of every root bucket, in array order
        {
            for (int b = 128; b < 256; ++b) if (h_map[b]) code_of_root.push_back(h_map[b]);
            for (int b = 0; b < 128; ++b) if (h_map[b]) code_of_root.push_back(h_map[b]);
        }
        for (size_t b = 0; b < folded_roots.size(); ++b) {
            const CompatBucket r = folded_roots[b];
            if (!depth1.empty() && depth1[b].done) {  // written as [end][high][low]
                blks.push_back(Blk{r.lo, depth1[b].nend, code_of_root[b], 0});
                blks.push_back(Blk{r.lo + depth1[b].nend, depth1[b].nhigh, code_of_root[b], 2});
                blks.push_back(Blk{r.lo + depth1[b].nend + depth1[b].nhigh, depth1[b].nlow, code_of_root[b], 1});
            } else {
                blks.push_back(Blk{r.lo, r.hi - r.lo, code_of_root[b], 0});
            }
        }
        // unsigned order: by first-symbol code, inside a bucket [end][low][high]
        std::vector<size_t> by_u(blks.size());
        for (size_t i = 0; i < by_u.size(); ++i) by_u[i] = i;
        std::sort(by_u.begin(), by_u.end(), [&](size_t x, size_t y) {
            return blks[x].code != blks[y].code ? blks[x].code < blks[y].code : blks[x].cls < blks[y].cls;
        });
        std::vector<unsigned long long> tab(2 * blks.size());
        unsigned long long at = 0;
        for (size_t i : by_u) {
            tab[blks.size() + i] = at;
            at += blks[i].len;
        }
        for (size_t i = 0; i < blks.size(); ++i) tab[i] = blks[i].a;
        d_order.alloc(tab.size() * sizeof(unsigned long long));
        CDB_HIP(hipMemcpyAsync(d_order.p, tab.data(), tab.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        CDB_HIP(hipStreamSynchronize(s));
        order.a_start = d_order.as<unsigned long long>();
        order.u_start = d_order.as<unsigned long long>() + blks.size();
        order.nseg = (uint32_t)blks.size();
    }
    DevBuf U, skey[2], sval[2], nh, rank, hcov;
    uint32_t h_acc = 0;  // symbols the text-extension rounds so far added behind every group's own depth (variable-length keys)
    uint64_t h = refine_depth0 ? refine_depth0 : (uint64_t)nsym;
    bool isa = false;
    uint64_t cap = 0;
    DevBuf d_open;  // entries still unresolved after the last round (saves a full flag scan to learn "none")
    d_open.alloc(sizeof(uint64_t));
    DevBuf U2b, lf;       // list-based rounds: the other list of positions, the flag bytes of the current list in list order
    bool have_list = false;  // the last round was a text-extension round: lf / U / sval[list_rs] describe its list
    uint64_t m_list = 0;
    int list_rs = 0;
    DevBuf d_gs_state;  // sa_group_sort_kernel: [0] it gave up (groups too long for it), [1] members walked by its long walks
    d_gs_state.alloc(2 * sizeof(unsigned long long));
    bool group_sort_on = ix.group_sort;
    // (generic over the array's storage: SaRW<V> = plain entries, Sa40RW = packed 5-byte entries, index_impl.h)
    auto refine = [&](auto sa) {
    using SAW = decltype(sa);
    for (;;) {
        if (st.rounds > 0) {
            uint64_t still = 0;
            CDB_HIP(hipMemcpyAsync(&still, d_open.p, sizeof(still), hipMemcpyDeviceToHost, s));
            CDB_HIP(hipStreamSynchronize(s));
            if (still == 0) break;
        }
        CDB_HIP(hipMemsetAsync(d_open.p, 0, sizeof(uint64_t), s));
        auto flag_tile_sums = [&]() {  // raw tile sums of the flag array into the scan's partials (wave-autonomous sweep)
            const uint64_t nb = ceil_div(n, SC_TILE);
            ix.scan_partials.ensure(scan_partials_slots(nb) * sizeof(U2));
            int t = ix.prof.begin(s);
            hipLaunchKernelGGL(sa_flag_count_kernel, dim3((unsigned)ceil_div(nb, (uint64_t)(4 * FC_TILES_PER_WAVE))), dim3(256), 0, s,
                               (const uint8_t*)flags.as<uint8_t>(), n, nb, ix.scan_partials.as<U2>());
            ix.prof.end(t, "sa_flag_count", n, s);
        };
        bool from_list = have_list && ix.list_rounds;  // (this round's unresolved entries are a subset of the previous round's list)
        have_list = false;
        if (from_list) {
            const uint64_t nb = ceil_div(m_list, SC_TILE);
            ix.scan_partials.ensure(scan_partials_slots(nb) * sizeof(U2));
            int t = ix.prof.begin(s);
            hipLaunchKernelGGL(sa_flag_count_kernel, dim3((unsigned)ceil_div(nb, (uint64_t)(4 * FC_TILES_PER_WAVE))), dim3(256), 0, s,
                               (const uint8_t*)lf.as<uint8_t>(), m_list, nb, ix.scan_partials.as<U2>());
            ix.prof.end(t, "sa_flag_count", m_list, s);
        } else if (!tile_sums_ready) {
            flag_tile_sums();
        }
        U2 tot = scan_totals_from_partials<U2>(s, ix.scan_partials, from_list ? m_list : n, OpAdd{}, U2{0, 0});
        tile_sums_ready = false;
        uint64_t m = tot.a, G = tot.b;
        if (st.rounds == 0) st.unresolved_initial = m;
        st.unresolved_max = std::max(st.unresolved_max, m);
        if (m == 0) break;
        // (the per-entry kernels of a round address one thread per unresolved entry: a launch holds < 2^32 of them)
        if (m >= (1ull << 32) - 4096) throw Error("too many unresolved suffixes for one refinement round (internal limit: 2^32)");
        const int gbits = bit_width64(G - 1);
        int nsym2 = std::min((64 - gbits) / symbits, KG_LOOK);
        if (!isa) {
            // Re-keying m suffixes from the text costs ~ m x (one random line + a sort); the inverse array of prefix
            // doubling costs n random writes before its first round.  Text extension therefore gets up to four
            // rounds while few suffixes are open, and two rounds even for a larger share (e.g. Zipf text, 5 % open
            // after the initial sort, resolved by 2 x 5 more symbols); only then the inverse array is built.
            const bool use_text = !ix.force_doubling && nsym2 >= 1 &&
                                  ((m <= n / 32 && st.ext_rounds < 4) || (m <= n / 8 && st.ext_rounds < 2));
            if (!use_text) {
                rank.alloc((n + D + 1) * sizeof(R));
                CDB_HIP(hipMemsetAsync(rank.p, 0, (n + D + 1) * sizeof(R), s));
                AllHeadIn ain{flags.as<uint8_t>()};
                (void)scan_totals<uint64_t>(s, ix.scan_partials, ain, n, OpMax{}, (uint64_t)0);
                IsaOut<SAW, R> iout{sa, doc_start, (int)ix.bits, ix.mask, rank.as<R>(), order};
                int t = ix.prof.begin(s);
                scan_apply<uint64_t>(s, ix.scan_partials, ain, n, OpMax{}, (uint64_t)0, iout);
                ix.prof.end(t, "sa_isa_init", n * (1 + sizeof(V) + sizeof(R)), s);
                isa = true;
                st.isa_built = 1;
                // the partials buffer now belongs to the max-scan: redo the compaction totals (from the array: the doubling rounds
                // do not keep a list)
                from_list = false;
                flag_tile_sums();
                (void)scan_totals_from_partials<U2>(s, ix.scan_partials, n, OpAdd{}, U2{0, 0});
            }
        }
        const int kbits = isa ? bit_width64(n) : nsym2 * symbits;
        if (kbits + gbits > 64) throw Error("refinement key does not fit 64 bits (internal limit)");
        if (m > cap) {
            cap = m;
            U.alloc(m * sizeof(I));
            skey[0].alloc(m * 8);
            skey[1].alloc(m * 8);
            sval[0].alloc(m * sizeof(V));
            sval[1].alloc(m * sizeof(V));
            nh.alloc(m);
            lf.alloc(m + 16);
            if (vl_kb1) hcov.alloc(m);
        }
        {
            int t = ix.prof.begin(s);
            if (from_list) {
                if (list_rs == 0) {  // the sorted list must not be the compaction's target
                    std::swap(skey[0], skey[1]);
                    std::swap(sval[0], sval[1]);
                    list_rs = 1;
                }
                U2b.ensure(cap * sizeof(I));
                CompactListOut<V, I> co{(const I*)U.as<I>(), (const V*)sval[1].as<V>(), U2b.as<I>(), skey[0].as<uint64_t>(), sval[0].as<V>(), kbits};
                hipLaunchKernelGGL((sa_flag_compact_kernel<decltype(co)>), dim3((unsigned)ceil_div(ceil_div(m_list, SC_TILE), (uint64_t)(4 * FC_TILES_PER_WAVE))),
                                   dim3(256), 0, s, (const uint8_t*)lf.as<uint8_t>(), m_list, (uint64_t)ceil_div(m_list, SC_TILE),
                                   (const U2*)ix.scan_partials.as<U2>(), co);
                std::swap(U, U2b);
                st.list_rounds++;
            } else {
            CompactOut<SAW, I> co{sa, U.as<I>(), skey[0].as<uint64_t>(), sval[0].as<V>(), kbits};
            hipLaunchKernelGGL((sa_flag_compact_kernel<decltype(co)>), dim3((unsigned)ceil_div(ceil_div(n, SC_TILE), (uint64_t)(4 * FC_TILES_PER_WAVE))),
                               dim3(256), 0, s, (const uint8_t*)flags.as<uint8_t>(), n, (uint64_t)ceil_div(n, SC_TILE),
                               (const U2*)ix.scan_partials.as<U2>(), co);
            }
            if (isa)
                hipLaunchKernelGGL((sa_round_keys_kernel<V, R, true>), dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, s,
                                   (const V*)sval[0].as<V>(), m, doc_start, text, (const uint16_t*)d_symmap.as<uint16_t>(),
                                   (const R*)rank.as<R>(), (int)ix.bits, ix.mask, h, nsym2, symbits, skey[0].as<uint64_t>(), n);
            else
                hipLaunchKernelGGL((sa_round_keys_kernel<V, R, false>), dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, s,
                                   (const V*)sval[0].as<V>(), m, doc_start, text, (const uint16_t*)d_symmap.as<uint16_t>(),
                                   (const R*)nullptr, (int)ix.bits, ix.mask, h, nsym2, symbits, skey[0].as<uint64_t>(), n,
                                   vl_kb1 ? (const uint8_t*)d_vl_bytelen.as<uint8_t>() : nullptr, vl_kb1, h_acc, vl_kb1 ? hcov.as<uint8_t>() : nullptr);
            ix.prof.end(t, "sa_compact", n + m * (sizeof(I) + 8 + 2 * sizeof(V)), s);
        }
        int rs = -1;
        if (group_sort_on) {  // one pass inside the groups (sa_group_sort_kernel); groups too long for it: the general sort, from now on
            CDB_HIP(hipMemsetAsync(d_gs_state.p, 0, 2 * sizeof(unsigned long long), s));
            int t = ix.prof.begin(s);
            hipLaunchKernelGGL((sa_group_sort_kernel<V>), dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, s, (const uint64_t*)skey[0].as<uint64_t>(),
                               (const V*)sval[0].as<V>(), m, kbits, (uint32_t)ix.group_sort_cap, (unsigned long long)(16 * m + (1u << 20)),
                               skey[1].as<uint64_t>(), sval[1].as<V>(), d_gs_state.as<unsigned long long>());
            ix.prof.end(t, "sa_group_sort", m * 2 * (8 + sizeof(V)), s);
            unsigned long long gave_up = 0;
            CDB_HIP(hipMemcpyAsync(&gave_up, d_gs_state.p, sizeof(gave_up), hipMemcpyDeviceToHost, s));
            CDB_HIP(hipStreamSynchronize(s));
            if (gave_up) {
                group_sort_on = false;
                st.group_sort_fallbacks++;
            } else {
                rs = 1;
                st.group_sorts++;
            }
        }
        if (rs < 0) {
            rs = radix_sort<uint64_t, V>(s, ix.rws, ix.prof, skey[0].as<uint64_t>(), skey[1].as<uint64_t>(),
                                         sval[0].as<V>(), sval[1].as<V>(), m, 0, kbits + gbits, &ss, ix.sort_variant);
            radix_check_error(s, ix.rws);  // (sa_update scatters through the sorted values: never after a failed sort)
        }
        hipLaunchKernelGGL((sa_newhead_kernel<I>), dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, s,
                           (const uint64_t*)skey[rs].as<uint64_t>(), (const I*)U.as<I>(),
                           (const uint8_t*)flags.as<uint8_t>(), m, nh.as<uint8_t>());
        const uint64_t hnew = isa ? 2 * h : h + (uint64_t)nsym2;
        {
            int t = ix.prof.begin(s);
            if (isa) {
                HeadIn<I> hin{nh.as<uint8_t>(), U.as<I>()};
                (void)scan_totals<uint64_t>(s, ix.scan_partials, hin, m, OpMax{}, (uint64_t)0);
                UpdateOut<SAW, I, R> uo{sval[rs].as<V>(), U.as<I>(), nh.as<uint8_t>(), m, doc_start, (int)ix.bits,
                                      ix.mask, hnew, sa, flags.as<uint8_t>(), rank.as<R>(), d_open.as<unsigned long long>(), order};
                scan_apply<uint64_t>(s, ix.scan_partials, hin, m, OpMax{}, (uint64_t)0, uo);
                st.dbl_rounds++;
            } else {
                hipLaunchKernelGGL((sa_update_kernel<SAW, I>), dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, s,
                                   (const V*)sval[rs].as<V>(), (const I*)U.as<I>(), (const uint8_t*)nh.as<uint8_t>(), m,
                                   doc_start, (int)ix.bits, ix.mask, hnew, sa, flags.as<uint8_t>(), d_open.as<unsigned long long>(),
                                   vl_kb1 ? (const uint8_t*)hcov.as<uint8_t>() : nullptr, h_acc + (uint32_t)nsym2, lf.as<uint8_t>());
                h_acc += (uint32_t)nsym2;
                st.ext_rounds++;
                have_list = true;
                m_list = m;
                list_rs = rs;
            }
            ix.prof.end(t, "sa_update", m * (2 * sizeof(V) + sizeof(I) + 2 + (isa ? sizeof(R) : 0)), s);
        }
        h = hnew;
        st.rounds++;
        if (st.rounds > 80) throw Error("suffix-array refinement did not converge (internal error)");
    }
    };
    if constexpr (sizeof(V) == 8) {
        if (packed_out) refine(Sa40RW{sa_buf.as<uint32_t>(), sa_hi_buf.as<uint8_t>()});
        else refine(SaRW<V>{sa_buf.as<V>()});
    } else {
        refine(SaRW<V>{sa_buf.as<V>()});
    }
    if (ix.debug_fail_build) throw Error("debug: build failure requested (test hook)");
    if (ix.debug_starve_group == 1 && !ix.rws.plain_order) throw Error("radix sort look-back timed out (test hook)");
    st.final_depth = h;
    st.sort_passes = ss.passes_run;
    st.sort_passes_skipped = ss.passes_skipped;
    radix_check_error(s, ix.rws);
    CDB_HIP(hipStreamSynchronize(s));
    ix.sa_sorted = !(ix.reference_compat && high_bytes);
    ix.pivot_levels = 0;  // the pivot table belongs to the previous suffix array
    if (ix.key_nsym) {  // the code table the kept keys were built with
        ix.d_symmap_q = std::move(d_symmap);
        std::memcpy(ix.h_symmap_q, h_map, sizeof(h_map));
    }
    if (ix.reference_compat && high_bytes) {
        // buckets whose two byte blocks were swapped by the sort enter the walk as their two blocks (each sorted by the
        // second symbol and with all children on one side of 0x80: no rotation there, their big children are walked)
        if (!folded_roots.empty() && !depth1.empty()) {
            std::vector<CompatBucket> nodes;
            for (size_t b = 0; b < folded_roots.size(); ++b) {
                const CompatBucket r = folded_roots[b];
                if (depth1[b].done) {
                    const unsigned long long hs = r.lo + depth1[b].nend, ls = hs + depth1[b].nhigh;
                    // (every slot must belong to some entry of the list: the out-of-place form of the walk COPIES the
                    //  array range by range — the fuzz-sized test caught the block of one-symbol suffixes missing here)
                    if (depth1[b].nend) nodes.push_back(CompatBucket{r.lo, hs});
                    nodes.push_back(CompatBucket{hs, ls});
                    nodes.push_back(CompatBucket{ls, r.hi});
                } else {
                    nodes.push_back(r);
                }
            }
            folded_roots.swap(nodes);
        }
        const std::vector<CompatBucket>* roots = folded_roots.empty() ? nullptr : &folded_roots;
        auto reorder = [&](auto tag) {
            using T = decltype(tag);
            if (!apply_reference_order_oop<T>(ix, sa_buf, &sa_hi_buf, roots)) {
                apply_reference_order<T>(ix, sa_buf, &sa_hi_buf, roots);  // in place when no second array fits
                ix.drop_keys();                                           // ... which moves the entries away from their keys
            }
        };
        if (packed_out) reorder(Packed40{});
        else reorder(V{});
    }
    ix.d_sa = std::move(sa_buf);
    ix.d_sa_hi = std::move(sa_hi_buf);
    ix.sa_packed = packed_out;
}

}  // namespace

void build_suffix_array(Index& ix) {
    const double t0 = now_ms();
    proof_stop(ix);           // (the order proof of the previous array reads what this build replaces)
    if (!ix.proof_in_repair) ix.proof.state.store(0);  // (whatever was proved, it was the previous array)
    query_resident_stop(ix);  // (a resident query workgroup reads the arrays this build replaces)
    struct GroupScope {  // the build's sorts may use the XCD-aware tile order: a starved pass is redone below
        RadixWorkspace& ws;
        explicit GroupScope(RadixWorkspace& w) : ws(w) { ws.allow_group = true; }
        ~GroupScope() { ws.allow_group = false; }
    } gscope(ix.rws);
    if (!ix.debug_starve_group && rs_group_order_starved(ix.device)) ix.rws.plain_order = true;  // (learnt by an earlier handle)
    auto run = [&]() {
        const bool wide = ix.size + ix.ndocs + 2 >= (1ull << 32) || ix.force_big_path;
        if (wide && ix.width != 8 && !ix.force_big_path) throw Error("internal: a corpus >= 2^32 bytes must have 8-byte entries");
        if (!wide) {
            if (ix.width == 4) build_typed<uint32_t, uint32_t, uint32_t>(ix, false);
            else build_typed<uint64_t, uint32_t, uint32_t>(ix, false);
        } else {
            if (ix.width == 4) build_typed<uint32_t, uint64_t, uint64_t>(ix, true);  // only reachable through force_big_path
            else build_typed<uint64_t, uint64_t, uint64_t>(ix, true);
        }
    };
    ix.vl_off_once = false;
    auto run_dense_after_vl = [&]() {
        try {
            run();
        } catch (const RetryWithDenseKeys&) {
            ix.dense_key_retries += 1;  // (visible: a column that pays two build prologues on every build is a regression)
            (void)hipStreamSynchronize(ix.stream);
            ix.prof.resolve();
            ix.release_sa();
            ix.drop_keys();
            ix.vl_off_once = true;
            run();
        }
    };
    try {
        try {
            run_dense_after_vl();
        } catch (const Error& e) {
            // A pass in XCD-aware tile order needs a few dozen workgroups resident at once (radix_sort.h: RS_GROUP);
            // other kernels on the device can starve it, which ends in the (bounded) look-back timeout.  The plain
            // ticket order needs one resident workgroup: rebuild with it, and keep it for this handle.
            if (ix.rws.plain_order || std::strstr(e.what(), "look-back timed out") == nullptr) throw;
            (void)hipStreamSynchronize(ix.stream);
            ix.prof.resolve();
            ix.release_sa();
            ix.drop_keys();
            ix.rws.plain_order = true;
            if (!ix.debug_starve_group) rs_group_order_disable(ix.device);  // (the test hook leaves the device alone)
            ix.group_fallbacks += 1;
            run_dense_after_vl();
        }
    } catch (...) {
        // (the scratch buffers went back to the block cache while the stack unwound; they carry an event of this
        //  stream, so no other stream gets them before the kernels queued here have finished — common.h: DevPool)
        (void)hipStreamSynchronize(ix.stream);
        ix.prof.resolve();
        ix.release_sa();
        ix.drop_keys();
        ix.width = 0;  // back to "never built": queries answer {} instead of touching a half-built array
        ix.size = 0;
        throw;
    }
    CDB_HIP(hipStreamSynchronize(ix.stream));
    ix.prof.resolve();
    // 8-byte entries below 2^40 are stored packed (index_impl.h: Sa40).  The fused bucket-wise build writes that form itself;
    // the other paths with 8-byte entries (partition + gather, columns below 2^32 bytes with many documents) pack here
    auto pack_now = [&]() {
        if (sa_packable(ix) && !ix.sa_packed) {
            int t = ix.prof.begin(ix.stream);
            sa_pack_inplace(ix);
            ix.prof.end(t, "sa_pack", ix.size * 13, ix.stream);
            ix.prof.resolve();
        }
    };
    pack_now();
    // Spot check of the finished array (verify.hip).  The stable ranking of the passes rests on observed LDS behaviour
    // (radix_sort.h: one-atomic ranking, self-tested per device); should a build ever come out wrong, this process
    // switches the device to the ballot ranking, rebuilds once, and fails loudly if that does not help either.
    if (ix.self_check && ix.size >= 2) {
        uint64_t sc[2] = {0, 0};
        auto check = [&]() {
            // The sample: n / 4096 random adjacent pairs, at least 2^15 and at most 2^21.  It notices a ranking that went wrong —
            // that scatters inversions over the whole array — and nothing smaller: with b bad pairs among n the 2^21 samples miss
            // them all with probability exp(-2^21 b / n) (round 5's wrong sweep records, 1 295 inversions among 8.6 x 10^9 pairs:
            // 73 % missed; 95 % detection needs ~12 000 bad pairs at that size).  It is a smoke alarm, not a proof.
            // self_check = 2: EVERY adjacent pair before the build returns (1.2-1.5 x the build).  self_check = 3 (default): the
            // sample here and every pair AFTER the build has returned, on a helper thread (verify.hip: proof_start).
            const double tc = now_ms();
            const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(1u << 15, ix.size >> 12), 1u << 21);
            // self_check = 3 (default): the sample here, the proof behind the publish (below)
            const uint32_t samples = ix.self_check == 2 ? 0u : (uint32_t)std::min<uint64_t>(want, ix.size - 1);
            spot_check_suffix_array(ix, samples, sc);
            ix.self_check_pairs = samples ? samples : ix.size - 1;
            ix.self_check_ms = now_ms() - tc;
            if (ix.debug_fail_self_check && ix.self_check_fallbacks == 0) sc[0] += 1;
            return sc[0] == 0 && sc[1] == 0;
        };
        if (!check()) {
            const bool was_atomic = rs_atomic_rank_ok(ix.stream);
            if (was_atomic && !ix.debug_fail_self_check) rs_atomic_rank_disable(ix.device);  // (the test hook leaves the device alone)
            ix.self_check_fallbacks += 1;
            ix.release_sa();  // (the failed array and its keys go first: the rebuild needs their memory on large corpora)
            ix.drop_keys();
            try {
                run_dense_after_vl();
                CDB_HIP(hipStreamSynchronize(ix.stream));
                ix.prof.resolve();
                pack_now();
            } catch (...) {
                (void)hipStreamSynchronize(ix.stream);
                ix.prof.resolve();
                ix.release_sa();
                ix.drop_keys();
                ix.width = 0;
                ix.size = 0;
                throw;
            }
            if (!check()) {
                ix.release_sa();
                ix.drop_keys();
                ix.width = 0;
                ix.size = 0;
                throw Error("suffix array self-check failed: " + std::to_string(sc[0]) + " pairs out of order, " + std::to_string(sc[1]) +
                            " invalid entries among the sampled pairs (internal error)");
            }
        }
    }
    ix.bstats.build_ms = now_ms() - t0;
    if ((ix.self_check >= 3 || ix.premap_generation) && !ix.proof_in_repair) {
        // every adjacent pair against the text, AFTER the caller has its index: a helper thread on a low-priority stream
        // (verify.hip: proof_start); cdb_get_stat("order_proved") goes 0 -> 1, damage is repaired under ix.mu
        if (ix.debug_damage_after_build) debug_swap_entries(ix, ix.debug_damage_after_build);
        ix.proof.of_loaded_file = false;
        proof_start(ix);
    }
    if (getenv("CDB_BUILD_TRACE"))
        std::fprintf(stderr, "[build] n=%llu: %.1f ms (allocation %.1f ms, release %.1f ms, self check %.2f ms), host upload before it %.1f ms\n",
                     (unsigned long long)ix.size, ix.bstats.build_ms, ix.bstats.alloc_ms, ix.bstats.free_ms, ix.self_check_ms, ix.host_upload_ms);
}

}  // namespace cdb
