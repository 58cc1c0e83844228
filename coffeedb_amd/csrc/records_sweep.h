// records_sweep.h — bucket records of ONE GROUP of first-symbol buckets straight from the text (round 4).
//
// The bucket-wise build (sa_build.hip, >= 2^32 suffixes) whose records do not fit the device all at once used to PARTITION the
// entries by first symbol (one generated pass over the text, 5-8 B written per suffix) and then, group of buckets by group,
// GATHER every bucket's records through those entries (entries re-read, the text re-read in bucket order, records written):
// 16 GiB of UTF-8: 82 + 129 ms of 692.  The look-back-free form of the generated passes (radix_sort.h: TextGen::tile_base)
// makes both unnecessary: with the per-tile byte counts scanned over the tiles, every tile knows where its suffixes of every
// bucket go, so a sweep over the text can write the records of ANY subset of the buckets in place — one sweep per group, each
// reading the text once (1 B per suffix) and writing only its group's records.
//
// A sweep must be cheap for the suffixes it does NOT keep (two thirds of them with three groups), so the kernel ranks first and
// generates afterwards: phase A looks at one staged symbol per position (bucket slot -> LDS counter -> rank), phase B walks the
// tile's KEPT positions in output order, evaluates their records (nsym - 1 <= 10 codes behind the
// bucket symbol as a number in base B <= 255, pairs by v_dot4_u32_u8, Horner in base B^2) and writes them lane-consecutively —
// no record ever crosses the LDS, only a 16-bit position does (40 KB of LDS: four workgroups per CU).
//
// reference: the records are the sort keys of the radix nodes, src/index.cpp:96-126 (257-way buckets on character(), src/index.h:66-73),
// restated as dense numbers per first-symbol bucket; DESIGN.md §4.2.
#pragma once
#include "radix_sort.h"

namespace cdb {

// Timing-only ablations of the sweep kernel (WRONG records) exist only in a library compiled with -DRS_SWEEP_ABL=<bits> and loaded
// through CDB_LIB_PATH by the measuring script — never in the product build, and never chosen at run time.
#ifndef RS_SWEEP_ABL
#define RS_SWEEP_ABL 0
#endif
#ifndef RS_SWEEP_UNALIGNED
#define RS_SWEEP_UNALIGNED 0   // (measured: unaligned 8-byte LDS reads of the code windows, see rs_sweep_msd_kernel)
#endif
// A wave-uniform switch, pinned to an SGPR AT THE POINT OF USE.  Round 5 met a wrong-code shape of hipcc (ROCm 7.2): booleans
// derived from wave-uniform kernel arguments (2 q < nsym - 1, part_m > 1, ...) were carried across basic blocks as LANE MASKS,
// re-materialised inside a divergent loop under the loop's current exec and tested again BEHIND the loop by lanes that had left it
// earlier — which read zeros (see the note in rs_sweep_records_kernel).  The volatile asm makes every call a fresh scalar value
// the optimiser can neither hoist nor merge with another copy of the same body, so the compares derived from it are computed
// (s_cmp, from the SGPR) inside the copy that uses them, under an exec that covers all of its lanes.  No instruction is emitted.
__device__ __forceinline__ int rs_uniform(int x) {
    int y = __builtin_amdgcn_readfirstlane(x);
    asm volatile("" : "+s"(y));
    return y;
}
// inclusive prefix sum over the 64 lanes of a wavefront by DPP adds (row shifts inside 16 lanes, then the row totals broadcast to
// the rows behind them): seven vector instructions, no LDS — __shfl_up costs a ds_bpermute and three vector instructions per step
__device__ __forceinline__ uint32_t rs_wave_incl_scan(uint32_t x) {
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, false);  // row_shr:3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);  // row_shr:4, lanes 4-15 of every row
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);  // row_shr:8, lanes 8-15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
    return v;
}

constexpr int RS_SWEEP_TILE = RS_GEN8_TILE;  // 512 threads x 16 positions: two workgroups per CU
constexpr uint32_t RS_SWEEP_DOCS = 768;      // document starts of a tile kept in LDS (more: binary searches in global memory)

// gen: text, doc_start, bits, base, nsym, rec_low_bits, padded, tile_doc (per RS_SWEEP_TILE), tile_base ([tiles][256]: array-wide
// output slot of the tile's first suffix of every bucket slot); codeslot[byte] = symbol code | bucket slot << 8.
// Keeps the suffixes whose bucket slot is in [g0, g1); record r of the group lands at index tile_base - gstart.
// Round 5 (profiles/r05a_sq_counters.txt: 105 vector and 17 LDS instructions per position, vector units and LDS ~78 % busy): the
// document of a kept position comes from a per-tile table instead of searches — per document a 64-bit entry bias
// ((tile base - start) << bits) + document and the start of the NEXT document relative to the tile, per 32 positions the bit map
// of document starts with the number of starts in front of it: document = count + popcount(bits & mask), entry = bias +
// (position << bits), symbols left = next start - position — no 64-bit compares, no dependent table walk; the per-slot tile
// starts are folded into the waves' counters, the scans are DPP adds.  Tiles with more documents than the table holds, with
// empty documents, and the ragged last tile take the generic path (searches in global memory).
template <typename W>
__global__ __launch_bounds__(512, 8) void rs_sweep_records_kernel(TextGen gen, const uint16_t* __restrict__ codeslot, uint64_t n, uint32_t tiles,
                                                                  uint32_t g0, uint32_t g1, uint64_t gstart, uint32_t* __restrict__ kout,
                                                                  uint32_t* __restrict__ vout, W* __restrict__ wout) {
    constexpr int NT = 512, IPT = 16, TILE = RS_SWEEP_TILE, NW = NT / 64;
    static_assert(NT * IPT == TILE, "tile shape");
    constexpr int abl = RS_SWEEP_ABL;  // (0 in every product build; timing-only ablations are a COMPILE-time choice, see the top of the file)
    constexpr uint32_t TEXTB = ((TILE + RS_GEN_LOOK + 15) / 16) * 16;
    constexpr uint32_t NBLK = TILE / 32, DOCS = 184;  // (40 KB of LDS in all: four workgroups per CU)
    __shared__ __attribute__((aligned(16))) uint8_t s_text[TEXTB];
    __shared__ uint16_t s_cs[256];
    __shared__ __attribute__((aligned(16))) uint32_t s_doc[4 * DOCS];          // document d: entry bias (64 bit), start of d + 1 - tile base, -
    __shared__ uint8_t s_slotc[256];  // symbol code -> bucket slot (phase B has a record's slot back from its first symbol)
    __shared__ uint32_t s_whist[NW][256];
    __shared__ __attribute__((aligned(8))) uint32_t s_blk[2 * (NBLK + 1)];     // block of 32 positions: start bits, 16 x (starts in front of it)
    __shared__ uint64_t s_gbase[256];
    __shared__ uint32_t s_wsum[8];
    __shared__ uint32_t s_flag;
    __shared__ __attribute__((aligned(16))) uint16_t s_idx[TILE];
    uint8_t* const s_dig = reinterpret_cast<uint8_t*>(s_idx);  // (its first half, until the ranking is done: the staged bucket slots)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware static tile map (no look-back, so no order between tiles is needed for progress): workgroup b runs on XCD b % 8,
    // which takes the tile groups x, x + 8, ... of RS_GROUP consecutive tiles each — neighbouring output runs meet in one L2
    const uint32_t x = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint64_t tile = (uint64_t)((slot / RS_GROUP) * 8u + x) * RS_GROUP + slot % RS_GROUP;
    if (tile >= (uint64_t)tiles) return;
    const uint64_t base = tile * TILE;
    const uint32_t valid = (uint32_t)((n - base) < (uint64_t)TILE ? (n - base) : (uint64_t)TILE);

    // ---- everything the tile needs from global memory is requested up front, in ONE round trip where possible
    const uint64_t ga = base + (uint64_t)tid * 16, gb = base + (uint64_t)TILE + (uint64_t)tid * 16;
    const bool has_b = (uint32_t)tid * 16 < TEXTB - (uint32_t)TILE;
    const bool oka = gen.padded ? (ga < n + RS_GEN_LOOK) : (ga + 16 <= n);
    const bool okb = has_b && (gen.padded ? (gb < n + RS_GEN_LOOK) : (gb + 16 <= n));
    uint4 ta = make_uint4(0, 0, 0, 0), tb = make_uint4(0, 0, 0, 0);
    if (oka) ta = *reinterpret_cast<const uint4*>(gen.text + ga);
    if (okb) tb = *reinterpret_cast<const uint4*>(gen.text + gb);
    uint64_t my_base = 0;
    if ((uint32_t)tid >= g0 && (uint32_t)tid < g1) my_base = (uint64_t)gen.tile_base[tile * 256 + (uint64_t)tid];
    if (tid < 256) {  // (a byte outside the group: slot 0xFF — alphabets of <= 254 symbols, rs_sweep_records_ok)
        const uint32_t e = codeslot[tid];
        s_cs[tid] = (uint16_t)(((e >> 8) >= g0 && (e >> 8) < g1) ? e : (e | 0xFF00u));
        s_slotc[tid] = gen.slotmap[tid];
    }
    const uint64_t dlo = gen.tile_doc[tile], dhi = gen.tile_doc[tile + 1];
    for (int i = tid; i < NW * 256; i += NT) (&s_whist[0][0])[i] = 0;
    if ((uint32_t)tid <= NBLK) s_blk[2 * tid] = 0;
    if (tid == 0) s_flag = 0;
    const uint32_t ndl = (uint32_t)(dhi - dlo);
    const bool docs_in_lds = dhi - dlo + 2 <= (uint64_t)DOCS;
    uint64_t dreg0 = 0;
    if (docs_in_lds && (uint32_t)tid < ndl + 2) dreg0 = gen.doc_start[dlo + tid];
    __syncthreads();
    if (docs_in_lds && (uint32_t)tid < ndl + 2) {
        const uint32_t d = (uint32_t)tid;
        const int64_t diff = (int64_t)(dreg0 - base);
        const int32_t rel = diff < -(1ll << 30) ? -(1 << 30) : (diff > (1ll << 30) ? (1 << 30) : (int32_t)diff);
        const uint64_t eb = ((base - dreg0) << gen.bits) + dlo + d;
        s_doc[4 * d] = (uint32_t)eb;
        s_doc[4 * d + 1] = (uint32_t)(eb >> 32);
        if (d >= 1) s_doc[4 * d - 2] = (uint32_t)rel;  // (the start of d beside the bias of d - 1: one 16-byte read in phase B)
        if (d >= 1 && rel <= (int32_t)TILE) {
            const uint32_t bit = 1u << ((uint32_t)rel & 31u);
            if (atomicOr(&s_blk[2 * ((uint32_t)rel >> 5)], bit) & bit) s_flag = 1;  // two starts on one position: an empty document
        }
    }
    // ---- staging: one table lookup per byte gives the symbol code (kept for phase B) and the bucket slot; both go to the LDS as
    // the 16-byte vectors the thread loaded
    auto fetch = [&](uint64_t g, bool ok, uint4 w, uint32_t* c) {
        c[0] = w.x; c[1] = w.y; c[2] = w.z; c[3] = w.w;
        if (!ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c[q] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) c[q] |= (uint32_t)((g + 4 * q + b < n) ? gen.text[g + 4 * q + b] : (uint8_t)0) << (8 * b);
            }
        }
    };
    {
        uint32_t c[4];
        fetch(ga, oka, ta, c);
        uint32_t e[IPT];
#pragma unroll
        for (int k = 0; k < IPT; ++k) e[k] = s_cs[(c[k >> 2] >> (8 * (k & 3))) & 0xFFu];
        uint32_t codes[4], slots[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            codes[q] = (e[4 * q] & 0xFFu) | ((e[4 * q + 1] & 0xFFu) << 8) | ((e[4 * q + 2] & 0xFFu) << 16) | ((e[4 * q + 3] & 0xFFu) << 24);
            slots[q] = (e[4 * q] >> 8) | ((e[4 * q + 1] >> 8) << 8) | ((e[4 * q + 2] >> 8) << 16) | ((e[4 * q + 3] >> 8) << 24);
        }
        if (valid < (uint32_t)TILE) {  // (uniform: the last tile) positions behind the text are kept by nobody
#pragma unroll
            for (int k = 0; k < IPT; ++k)
                if ((uint32_t)tid * 16 + k >= valid) slots[k >> 2] |= 0xFFu << (8 * (k & 3));
        }
        *reinterpret_cast<uint4*>(&s_text[(uint32_t)tid * 16]) = make_uint4(codes[0], codes[1], codes[2], codes[3]);
        *reinterpret_cast<uint4*>(&s_dig[(uint32_t)tid * 16]) = make_uint4(slots[0], slots[1], slots[2], slots[3]);
        if (has_b) {  // the look-ahead behind the tile: codes only
            fetch(gb, okb, tb, c);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                c[q] = (uint32_t)(s_cs[c[q] & 0xFF] & 0xFFu) | ((uint32_t)(s_cs[(c[q] >> 8) & 0xFF] & 0xFFu) << 8) |
                       ((uint32_t)(s_cs[(c[q] >> 16) & 0xFF] & 0xFFu) << 16) | ((uint32_t)(s_cs[c[q] >> 24] & 0xFFu) << 24);
            *reinterpret_cast<uint4*>(&s_text[(uint32_t)TILE + (uint32_t)tid * 16]) = make_uint4(c[0], c[1], c[2], c[3]);
        }
    }
    __syncthreads();
    if (abl & 2) return;  // (timing only: staging alone)
    // ---- phase A: rank the kept positions, lane-striped (position = wave chunk + j * 64 + lane): one returning LDS atomic on the
    // wave's counter of the slot per kept position.  The atomics of a wave instruction are served in lane order
    // (rs_atomic_rank_ok), so ranks ascend with the position: every bucket's records stay in TEXT order, which the stable passes
    // behind them hand on to suffixes with equal keys.  No dependent chain between positions: 16 reads, then 16 atomics.
    constexpr int WCHUNK = IPT * 64;
    const uint32_t wbase = wave * WCHUNK + lane;
    uint32_t info[IPT];  // rank | slot << 16; slot 0xFF = not kept
    {
        uint32_t sl[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) sl[j] = s_dig[wbase + j * 64];
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            uint32_t inf = 0xFFu << 16;  // (scalars inside the branch: element writes under divergent control turn the array into a 16-wide tuple)
            if (sl[j] != 0xFFu) {
                uint32_t col = sl[j];
                if (abl & 16) col = (col + (uint32_t)lane) & 0xFFu;  // (timing only: no two lanes on one counter)
                inf = atomicAdd(&s_whist[wave][col], 1u) | (sl[j] << 16);
            }
            info[j] = inf;
        }
    }
    __syncthreads();
    if (abl & 32) return;  // (timing only: staging + ranking)
    // ---- per-slot totals of the tile (waves 0-3) and, meanwhile (waves 4-7), the document starts in front of every block
    uint32_t wc[NW], cnt = 0, incl = 0;
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            wc[w] = s_whist[w][tid];
            cnt += wc[w];
        }
        incl = cnt;
    } else {
        incl = cnt = (uint32_t)__builtin_popcount(s_blk[2 * (tid - 256)]);
    }
    incl = rs_wave_incl_scan(incl);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    if (abl & 64) return;  // (timing only: ... + first half of the scan)
    const uint32_t kept = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    {
        uint32_t wpre = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (w < (wave & 3)) wpre += s_wsum[(wave & 4) + w];
        const uint32_t excl = wpre + incl - cnt;
        if (tid < 256) {  // every wave's counter of the slot becomes its first index in the tile's sorted order
            uint32_t run = excl;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                s_whist[w][tid] = run;
                run += wc[w];
            }
            s_gbase[tid] = my_base - gstart - (uint64_t)excl;
        } else {
            s_blk[2 * (tid - 256) + 1] = excl * 16u;
            if (tid == 511) s_blk[2 * NBLK + 1] = (excl + cnt) * 16u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const uint32_t sl = info[k] >> 16;
        if (sl != 0xFFu) s_idx[s_whist[wave][sl] + (info[k] & 0xFFFFu)] = (uint16_t)(wbase + k * 64);  // (the slot is not carried along)
    }
    __syncthreads();

    if (abl & 1) return;  // (timing only: no records)
    // ---- phase B: the kept positions in output order: record = (key >> low bits, entry low word, key low bits | entry high bits)
    const uint32_t* s_words = reinterpret_cast<const uint32_t*>(s_text);
    const uint32_t B = gen.base, B2 = B * B;
    const uint32_t wlo = B | (1u << 8), whi = (B << 16) | (1u << 24);  // dot4 weights: bytes 0,1 / bytes 2,3
    const int ns1_arg = gen.nsym - 1, part_arg = gen.part_m > 1 ? 1 : 0;
    const uint64_t lmask = (1ull << gen.rec_low_bits) - 1ull;
    auto emit = [&](uint32_t p, uint32_t li, uint64_t e64, uint32_t rem1) {  // rem1: key symbols left in the document
        // (the body's wave-uniform switches, scalar and private to THIS inlined copy of it: rs_uniform above)
        const int ns1 = rs_uniform(ns1_arg), part_on = rs_uniform(part_arg);
        const int nsx = ns1 + part_on;  // symbols the key looks at (incl. the quantised one in its leftover bits)
        const uint32_t sl = s_slotc[s_text[li]];
        const uint32_t l1 = li + 1u, wi = l1 >> 2, sel = l1 & 3u;
        const uint32_t w0 = s_words[wi], w1 = s_words[wi + 1], w2 = s_words[wi + 2];
        uint32_t x0 = __builtin_amdgcn_alignbyte(w1, w0, sel);  // codes of li + 1 .. li + 4
        uint32_t x1 = __builtin_amdgcn_alignbyte(w2, w1, sel);  // codes of li + 5 .. li + 8
        uint32_t x2 = 0;
        if (nsx > 8) x2 = __builtin_amdgcn_alignbyte(s_words[wi + 3], w2, sel);  // (uniform) codes of li + 9 .. li + 12
        if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(rem1 < (uint32_t)nsx) != 0)) != 0, 0)) {
            if (rem1 < (uint32_t)nsx) {  // the symbols behind the document end count as 0
                x0 = rem1 >= 4u ? x0 : (rem1 == 0u ? 0u : (x0 & ((1u << (8u * rem1)) - 1u)));
                x1 = rem1 >= 8u ? x1 : (rem1 <= 4u ? 0u : (x1 & ((1u << (8u * (rem1 - 4u))) - 1u)));
                x2 = rem1 >= 12u ? x2 : (rem1 <= 8u ? 0u : (x2 & ((1u << (8u * (rem1 - 8u))) - 1u)));
            }
        }
        uint64_t acc = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            if (2 * q < ns1) {  // (uniform)
                const uint32_t xw = q < 2 ? x0 : (q < 4 ? x1 : x2);
                uint32_t pv, mult;
                if (2 * q + 1 < ns1) {
                    pv = __builtin_amdgcn_udot4(xw, (q & 1) ? whi : wlo, 0u, false);
                    mult = B2;
                } else {  // an odd number of symbols: the last one stands alone
                    pv = (q & 1) ? ((xw >> 16) & 0xFFu) : (xw & 0xFFu);
                    mult = B;
                }
                if (q == 0) acc = pv;
                else if (q == 1) acc = (uint64_t)(uint32_t)acc * mult + pv;  // (below 2^16 x 2^16: one v_mad_u64_u32)
                else acc = acc * (uint64_t)mult + pv;
            }
        }
        if (part_on) {  // (uniform) the leftover key bits: the next symbol, quantised (TextGen::part_m)
            const uint32_t xq = ns1 < 4 ? x0 : (ns1 < 8 ? x1 : x2);
            const uint32_t cq = (xq >> (8 * (ns1 & 3))) & 0xFFu;
            acc = acc * (uint64_t)gen.part_m + (uint64_t)((cq * gen.part_r) >> gen.part_s);
        }
        const uint64_t dst = s_gbase[sl] + (uint64_t)p;
        if (abl & 8) {  // (timing only: no stores)
            if (acc == 0x123456789ull && e64 == 77) kout[dst] = 1;
            return;
        }
        kout[dst] = (uint32_t)(acc >> gen.rec_low_bits);
        vout[dst] = (uint32_t)e64;
        wout[dst] = (W)((acc & lmask) | ((e64 >> 32) << gen.rec_low_bits));
    };
    if (docs_in_lds && s_flag == 0 && valid == (uint32_t)TILE && !(abl & 4)) {
        const int bits = gen.bits;
        // (two records per trip pay when the whole tile is kept — one group: their loads are in flight together; a sweep that keeps
        //  a part of its positions takes one.  A first version paired records whenever half the tile was kept
        //      for (; p + NT < kept; p += 2 * NT) { rec(p); rec(p + NT); }   if (p < kept) rec(p);
        //  and came out with WRONG KEYS for the tail's lanes whenever other lanes of their wavefront ran one more pair
        //  (kept mod 2 NT > NT).  Cause, read off the ISA: hipcc (ROCm 7.2) keeps emit()'s wave-uniform switches (2 q < ns1,
        //  part_m > 1, ...) as lane masks and re-materialises them INSIDE the loop with vector compares under the current exec
        //  ("v_cndmask_b32 v, 0, 1, s[c]; v_cmp_ne_u32_e64 s[m], 1, v": bits only for the lanes still looping); the tail behind
        //  the loop tests the same s[m] for lanes that had left the loop earlier and reads zeros — every switch "on" for them.
        //  Nothing in the source is wrong; loops whose lanes leave together, or code that does not follow such a loop with a copy
        //  of its body, never show it.  tools/isa_lanemask_scan.py finds the shape in the assembly (the old build: 4 uses; every
        //  device object of the library is scanned by `make isa-scan` / tests/test_capi_cpu.py), and
        //  test_sweep_groups_that_keep_most_of_a_tile + the mid-size fuzz cover the values of `kept` that the all-kept and the
        //  one-third-kept configurations never produce.  Found at full size by the C3 test.)
        auto rec = [&](uint32_t p) {
            const uint32_t li = s_idx[p];
            const uint2 bk = *reinterpret_cast<const uint2*>(&s_blk[2 * (li >> 5)]);
            const uint32_t doff = bk.y + 16u * (uint32_t)__builtin_popcount(bk.x & ~(0xFFFFFFFEu << (li & 31u)));
            const uint4 dc = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(s_doc) + doff);  // bias lo, hi, next start
            emit(p, li, (((uint64_t)dc.y << 32) | dc.x) + ((uint64_t)li << bits), (uint32_t)((int32_t)dc.z - (int32_t)li - 1));
        };
        if (kept == (uint32_t)TILE) {  // (everything kept — one group: a fixed trip count, no lane leaves early)
            for (uint32_t p = tid; p < (uint32_t)TILE; p += 2 * NT) {
                rec(p);
                rec(p + NT);
            }
        } else {
            for (uint32_t p = tid; p < kept; p += NT) rec(p);
        }
    } else {
        // (the ragged last tile, tiles with more documents than the table holds or with empty ones: searches in global memory)
        for (uint32_t p = tid; p < kept; p += NT) {
            const uint32_t li = s_idx[p];
            const uint64_t pos = base + li;
            uint64_t dd = dlo, ds = 0, de = 1ull << 40;
            if (!(abl & 4)) {
                dd = rs_doc_upper(gen.doc_start, dlo, dhi, pos);
                ds = gen.doc_start[dd];
                de = gen.doc_start[dd + 1];
            }
            const uint64_t left = de - pos - 1ull;
            emit(p, li, ((pos - ds) << gen.bits) + dd, left < 64ull ? (uint32_t)left : 64u);
        }
    }
}

// ---- variable-length keys (round 5, vl_code.h) ------------------------------------------------------------------------------------
// The same sweep for keys that are the first B - 1 bits of the suffix's ALPHABETIC CODE STREAM behind its bucket symbol (+ one bit
// "the suffix continues behind the key").  Every position's key is a window into ONE bit stream, so the tile concatenates the
// code words of its symbols once (staging thread: 16 consecutive symbols -> two pieces of <= 56 bits, OR-ed into the LDS at the
// thread's bit offset, which a scan of the threads' bit counts provides) and keeps every position's 16-bit bit offset; phase B
// then reads 64 stream bits at the offset of its position: the first code word names the bucket slot and its own length
// (s_dec: the next 7 bits -> slot, length), the B - 1 bits behind it are the key.  Bits behind the end of the document are
// cleared: END is the all-zero word, so a short suffix is its code words, END, and zeros — in front of every continuation.
// The key's lowest bit says whether anything was cut off: keys with it clear are whole suffixes (equal keys = equal suffixes,
// final; the last pass tests "key mod 2 == 0" exactly like "key mod base == 0" of the dense coding).
// LDS: 52.5 KB — three workgroups per CU (the bit offsets are 16 KB).  Alphabets of <= 127 symbols (7-bit words, 128 slots).
struct VlTables {
    const uint16_t* sym = nullptr;  // [256] symbol code -> code word | length << 8 (code 0 = END)
    const uint16_t* dec = nullptr;  // [128] the next 7 stream bits -> bucket slot | length << 8 of the code word that starts there
    int key_bits = 0;               // B (whole key incl. the "continues" bit)
    int end_len = 0;                // length of END's (all-zero) word
};
constexpr uint32_t RS_VL_BITW = ((RS_GEN8_TILE + RS_GEN_LOOK) * 7 + 31) / 32 + 4;  // words of the tile's bit stream

template <typename W>
__global__ __launch_bounds__(512, 6) void rs_sweep_records_vl_kernel(TextGen gen, const uint16_t* __restrict__ codeslot, VlTables vl, uint64_t n,
                                                                     uint32_t tiles, uint32_t g0, uint32_t g1, uint64_t gstart,
                                                                     uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, W* __restrict__ wout) {
    constexpr int NT = 512, IPT = 16, TILE = RS_SWEEP_TILE, NW = NT / 64;
    constexpr uint32_t LOOK = RS_GEN_LOOK, NBLK = TILE / 32, DOCS = 256, SL = 128;
    static_assert(LOOK == 64, "four staging threads cover the look-ahead");
    __shared__ uint32_t s_bits[RS_VL_BITW];                                   // the code stream, most significant bit first
    __shared__ __attribute__((aligned(16))) uint16_t s_boff[TILE + LOOK + 8];  // bit offset of every position's code word (+ the end)
    __shared__ __attribute__((aligned(16))) uint16_t s_idx[TILE];
    __shared__ uint32_t s_whist[NW][SL];
    __shared__ uint16_t s_cs[256];
    __shared__ uint16_t s_vl[SL];
    __shared__ uint16_t s_dec[128];
    __shared__ __attribute__((aligned(16))) uint32_t s_doc[4 * DOCS + 4];      // document d: entry bias (64 bit), start of d + 1 - tile base, -
    __shared__ __attribute__((aligned(8))) uint32_t s_blk[2 * (NBLK + 1)];
    __shared__ uint64_t s_gbase[SL];
    __shared__ uint32_t s_wsum[16];
    __shared__ uint32_t s_flag;
    uint8_t* const s_dig = reinterpret_cast<uint8_t*>(s_idx);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t x = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint64_t tile = (uint64_t)((slot / RS_GROUP) * 8u + x) * RS_GROUP + slot % RS_GROUP;
    if (tile >= (uint64_t)tiles) return;
    const uint64_t base = tile * TILE;
    const uint32_t valid = (uint32_t)((n - base) < (uint64_t)TILE ? (n - base) : (uint64_t)TILE);

    const uint64_t ga = base + (uint64_t)tid * 16, gb = base + (uint64_t)TILE + (uint64_t)tid * 16;
    const bool has_b = (uint32_t)tid * 16 < LOOK;
    const bool oka = gen.padded ? (ga < n + LOOK) : (ga + 16 <= n);
    const bool okb = has_b && (gen.padded ? (gb < n + LOOK) : (gb + 16 <= n));
    uint4 ta = make_uint4(0, 0, 0, 0), tb = make_uint4(0, 0, 0, 0);
    if (oka) ta = *reinterpret_cast<const uint4*>(gen.text + ga);
    if (okb) tb = *reinterpret_cast<const uint4*>(gen.text + gb);
    uint64_t my_base = 0;
    if ((uint32_t)tid >= g0 && (uint32_t)tid < g1) my_base = (uint64_t)gen.tile_base[tile * 256 + (uint64_t)tid];
    if (tid < 256) {  // (a byte outside the group: slot 0xFF)
        const uint32_t e = codeslot[tid];
        s_cs[tid] = (uint16_t)(((e >> 8) >= g0 && (e >> 8) < g1) ? e : (e | 0xFF00u));
    }
    if (tid < (int)SL) {
        s_vl[tid] = vl.sym[tid];
        s_dec[tid] = vl.dec[tid];
    }
    const uint64_t dlo = gen.tile_doc[tile], dhi = gen.tile_doc[tile + 1];
    for (int i = tid; i < NW * (int)SL; i += NT) (&s_whist[0][0])[i] = 0;
    for (uint32_t i = tid; i < RS_VL_BITW; i += NT) s_bits[i] = 0;
    if ((uint32_t)tid <= NBLK) s_blk[2 * tid] = 0;
    if (tid == 0) s_flag = 0;
    const uint32_t ndl = (uint32_t)(dhi - dlo);
    const bool docs_in_lds = dhi - dlo + 2 <= (uint64_t)DOCS;
    uint64_t dreg0 = 0;
    if (docs_in_lds && (uint32_t)tid < ndl + 2) dreg0 = gen.doc_start[dlo + tid];
    __syncthreads();  // (the zeroes, the tables)
    if (docs_in_lds && (uint32_t)tid < ndl + 2) {
        const uint32_t d = (uint32_t)tid;
        const int64_t diff = (int64_t)(dreg0 - base);
        const int32_t rel = diff < -(1ll << 30) ? -(1 << 30) : (diff > (1ll << 30) ? (1 << 30) : (int32_t)diff);
        const uint64_t eb = ((base - dreg0) << gen.bits) + dlo + d;
        s_doc[4 * d] = (uint32_t)eb;
        s_doc[4 * d + 1] = (uint32_t)(eb >> 32);
        if (d >= 1) s_doc[4 * d - 2] = (uint32_t)rel;  // (the start of d beside the bias of d - 1: one 16-byte read in phase B)
        if (d >= 1 && rel <= (int32_t)TILE) {
            const uint32_t bit = 1u << ((uint32_t)rel & 31u);
            if (atomicOr(&s_blk[2 * ((uint32_t)rel >> 5)], bit) & bit) s_flag = 1;  // two starts on one position: an empty document
        }
    }
    // ---- staging, first half: bytes -> symbol code and bucket slot (one lookup), symbol code -> code word (a second one); the
    // thread's 16 code words as two left-aligned pieces of <= 56 bits and its bit count
    auto fetch = [&](uint64_t g, bool ok, uint4 w, uint32_t* c) {
        c[0] = w.x; c[1] = w.y; c[2] = w.z; c[3] = w.w;
        if (!ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c[q] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) c[q] |= (uint32_t)((g + 4 * q + b < n) ? gen.text[g + 4 * q + b] : (uint8_t)0) << (8 * b);
            }
        }
    };
    // four code words -> one word of <= 28 bits (right-aligned) and its length; lens4: the four lengths as nibbles
    auto pack4 = [&](const uint32_t* v, uint32_t& word, uint32_t& wlen, uint32_t& lens4) {
        const uint32_t l0 = v[0] >> 8, l1 = v[1] >> 8, l2 = v[2] >> 8, l3 = v[3] >> 8;
        word = ((((((v[0] & 0xFFu) << l1) | (v[1] & 0xFFu)) << l2) | (v[2] & 0xFFu)) << l3) | (v[3] & 0xFFu);
        wlen = l0 + l1 + l2 + l3;
        lens4 = l0 | (l1 << 4) | (l2 << 8) | (l3 << 12);
    };
    uint64_t pieceA[2] = {0, 0}, pieceB[2] = {0, 0};  // [0]: symbols 0-7, [1]: symbols 8-15 (A: the tile, B: the look-ahead)
    uint32_t plenA[2] = {0, 0}, plenB[2] = {0, 0}, lensA[2] = {0, 0}, lensB[2] = {0, 0};
    auto stage = [&](const uint32_t* c, uint64_t* piece, uint32_t* plen, uint32_t* lens, bool with_slots) {
        uint32_t e[IPT], v[IPT];
#pragma unroll
        for (int k = 0; k < IPT; ++k) e[k] = s_cs[(c[k >> 2] >> (8 * (k & 3))) & 0xFFu];
#pragma unroll
        for (int k = 0; k < IPT; ++k) v[k] = s_vl[e[k] & 0x7Fu];
        if (with_slots) {
            uint32_t slots[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) slots[q] = (e[4 * q] >> 8) | ((e[4 * q + 1] >> 8) << 8) | ((e[4 * q + 2] >> 8) << 16) | ((e[4 * q + 3] >> 8) << 24);
            if (valid < (uint32_t)TILE) {  // (uniform: the last tile) positions behind the text are kept by nobody
#pragma unroll
                for (int k = 0; k < IPT; ++k)
                    if ((uint32_t)tid * 16 + k >= valid) slots[k >> 2] |= 0xFFu << (8 * (k & 3));
            }
            *reinterpret_cast<uint4*>(&s_dig[(uint32_t)tid * 16]) = make_uint4(slots[0], slots[1], slots[2], slots[3]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t w0, n0, q0, w1, n1, q1;
            pack4(v + 8 * h, w0, n0, q0);
            pack4(v + 8 * h + 4, w1, n1, q1);
            plen[h] = n0 + n1;                                                   // <= 56
            piece[h] = (((uint64_t)w0 << n1) | (uint64_t)w1) << (64u - plen[h]);  // left-aligned (plen >= 8)
            lens[h] = q0 | (q1 << 16);
        }
    };
    {
        uint32_t c[4];
        fetch(ga, oka, ta, c);
        stage(c, pieceA, plenA, lensA, true);
        if (has_b) {
            fetch(gb, okb, tb, c);
            stage(c, pieceB, plenB, lensB, false);
        }
    }
    const uint32_t totA = plenA[0] + plenA[1], totB = has_b ? plenB[0] + plenB[1] : 0u;
    const uint32_t inclA = rs_wave_incl_scan(totA), inclB = rs_wave_incl_scan(totB);  // (look-ahead: lanes 0-3 of wave 0)
    if (lane == 63) s_wsum[wave] = inclA;
    __syncthreads();
    // ---- staging, second half: every position's bit offset, the pieces into the stream
    {
        uint32_t wpre = 0, all = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint32_t t = s_wsum[w];
            if (w < wave) wpre += t;
            all += t;
        }
        auto emit_bits = [&](uint32_t at, const uint64_t* piece, const uint32_t* plen, const uint32_t* lens, uint32_t pos0) {
            uint32_t o = at, packed[8];
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                if (k & 1) packed[k >> 1] |= o << 16; else packed[k >> 1] = o;
                o += (lens[k >> 3] >> (4 * (k & 7))) & 0xFu;
            }
            *reinterpret_cast<uint4*>(&s_boff[pos0]) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
            *reinterpret_cast<uint4*>(&s_boff[pos0 + 8]) = make_uint4(packed[4], packed[5], packed[6], packed[7]);
            uint32_t bo = at;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t wi = bo >> 5, sh = bo & 31u;
                const uint32_t xh = (uint32_t)(piece[h] >> 32), xl = (uint32_t)piece[h];
                atomicOr(&s_bits[wi], __builtin_amdgcn_alignbit(0u, xh, sh));
                atomicOr(&s_bits[wi + 1], __builtin_amdgcn_alignbit(xh, xl, sh));
                atomicOr(&s_bits[wi + 2], __builtin_amdgcn_alignbit(xl, 0u, sh));
                bo += plen[h];
            }
            return o;
        };
        (void)emit_bits(wpre + inclA - totA, pieceA, plenA, lensA, (uint32_t)tid * 16);
        if (has_b) {
            const uint32_t end = emit_bits(all + inclB - totB, pieceB, plenB, lensB, (uint32_t)TILE + (uint32_t)tid * 16);
            if ((uint32_t)tid * 16 + 16 == LOOK) s_boff[TILE + LOOK] = (uint16_t)end;
        }
    }
    __syncthreads();
    // ---- phase A: rank the kept positions, lane-striped (see rs_sweep_records_kernel)
    constexpr int WCHUNK = IPT * 64;
    const uint32_t wbase = wave * WCHUNK + lane;
    uint32_t info[IPT];  // rank | slot << 16; slot 0xFF = not kept
    {
        uint32_t sl[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) sl[j] = s_dig[wbase + j * 64];
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            uint32_t inf = 0xFFu << 16;
            if (sl[j] != 0xFFu) inf = atomicAdd(&s_whist[wave][sl[j]], 1u) | (sl[j] << 16);
            info[j] = inf;
        }
    }
    __syncthreads();
    // ---- per-slot totals (waves 0-1) and, meanwhile (waves 4-7), the document starts in front of every block of 32 positions
    uint32_t wc[NW], cnt = 0, incl = 0;
    if (tid < (int)SL) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            wc[w] = s_whist[w][tid];
            cnt += wc[w];
        }
        incl = cnt;
    } else if (tid >= 256) {
        incl = cnt = (uint32_t)__builtin_popcount(s_blk[2 * (tid - 256)]);
    }
    incl = rs_wave_incl_scan(incl);
    if (lane == 63) s_wsum[8 + wave] = incl;
    __syncthreads();
    const uint32_t kept = s_wsum[8] + s_wsum[9];
    {
        uint32_t wpre = 0;
        if (tid < (int)SL) {
            if (wave == 1) wpre = s_wsum[8];
            const uint32_t excl = wpre + incl - cnt;
            uint32_t run = excl;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                s_whist[w][tid] = run;
                run += wc[w];
            }
            s_gbase[tid] = my_base - gstart - (uint64_t)excl;
        } else if (tid >= 256) {
#pragma unroll
            for (int w = 4; w < NW; ++w)
                if (w < wave) wpre += s_wsum[8 + w];
            const uint32_t excl = wpre + incl - cnt;
            s_blk[2 * (tid - 256) + 1] = excl * 16u;
            if (tid == 511) s_blk[2 * NBLK + 1] = (excl + cnt) * 16u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const uint32_t sl = info[k] >> 16;
        if (sl != 0xFFu) s_idx[s_whist[wave][sl] + (info[k] & 0xFFFFu)] = (uint16_t)(wbase + k * 64);
    }
    __syncthreads();

    // ---- phase B: the kept positions in output order
    const int KB1 = vl.key_bits - 1;  // code bits of a key
    const uint64_t lmask = (1ull << gen.rec_low_bits) - 1ull;
    auto emit = [&](uint32_t p, uint32_t li, uint64_t e64, uint32_t endbit) {
        // 64 stream bits at the position's offset: its own code word (<= 7 bits: slot and length from the table), then the key
        const uint32_t o0 = s_boff[li];
        const uint32_t wi = o0 >> 5, sh = o0 & 31u;
        const uint32_t w0 = s_bits[wi], w1 = s_bits[wi + 1], w2 = s_bits[wi + 2];
        const uint32_t hi = __builtin_amdgcn_alignbit(w0, w1, 32u - sh), lo = __builtin_amdgcn_alignbit(w1, w2, 32u - sh);
        const uint64_t top = sh ? (((uint64_t)hi << 32) | lo) : (((uint64_t)w0 << 32) | w1);
        const uint32_t dv = s_dec[(uint32_t)(top >> 57)];
        const uint32_t sl = dv & 0xFFu, len = dv >> 8;
        const uint32_t avail = endbit - (o0 + len);  // code bits between the bucket symbol and the end of the document
        const uint32_t take = avail < (uint32_t)KB1 ? avail : (uint32_t)KB1;
        uint64_t kb = (top << len) >> (64 - KB1);                   // the next KB1 stream bits
        kb &= ~0ull << ((uint32_t)KB1 - take);                      // ... of which the document holds `take`: zeros behind its end
        const uint64_t key = (kb << 1) | (uint64_t)(avail + (uint32_t)vl.end_len > (uint32_t)KB1 ? 1u : 0u);
        const uint64_t dst = s_gbase[sl] + (uint64_t)p;
        kout[dst] = (uint32_t)(key >> gen.rec_low_bits);
        vout[dst] = (uint32_t)e64;
        wout[dst] = (W)((key & lmask) | ((e64 >> 32) << gen.rec_low_bits));
    };
    if (docs_in_lds && s_flag == 0 && valid == (uint32_t)TILE) {
        const int bits = gen.bits;
        for (uint32_t p = tid; p < kept; p += NT) {
            const uint32_t li = s_idx[p];
            const uint2 bk = *reinterpret_cast<const uint2*>(&s_blk[2 * (li >> 5)]);
            const uint32_t doff = bk.y + 16u * (uint32_t)__builtin_popcount(bk.x & ~(0xFFFFFFFEu << (li & 31u)));
            const uint4 dc = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(s_doc) + doff);  // bias lo, hi, next start
            const int32_t rn = (int32_t)dc.z;
            const uint32_t relend = rn < (int32_t)(TILE + LOOK) ? (uint32_t)rn : (uint32_t)(TILE + LOOK);
            emit(p, li, (((uint64_t)dc.y << 32) | dc.x) + ((uint64_t)li << bits), (uint32_t)s_boff[relend]);
        }
    } else {
        // (the ragged last tile, tiles with hundreds of documents or empty ones: searches in global memory)
        for (uint32_t p = tid; p < kept; p += NT) {
            const uint32_t li = s_idx[p];
            const uint64_t pos = base + li;
            const uint64_t dd = rs_doc_upper(gen.doc_start, dlo, dhi, pos);
            const uint64_t ds = gen.doc_start[dd], de = gen.doc_start[dd + 1];
            const uint64_t rel = de - base;
            const uint32_t relend = rel < (uint64_t)(TILE + LOOK) ? (uint32_t)rel : (uint32_t)(TILE + LOOK);
            emit(p, li, ((pos - ds) << gen.bits) + dd, (uint32_t)s_boff[relend]);
        }
    }
}

// The same two-phase form for the generated pass of the MSD-first sort below 2^32 (radix_sort.h: radix_sort_msd, pair form): digit
// = top digit of the 6-symbol key, a function of the first two symbols; records (u32 key - top * M, u32 entry).  Every position
// is kept.  gen: text, doc_start, symmap, bits, base, pair_span / pair_r / pair_s, padded, tile_doc, tile_base.
// The digit of a document's LAST position counts its second symbol as 0 — exactly what the per-tile counts the tile bases
// come from did (sa_build.hip: sa_tile_docend_fix_kernel).
//
// Round 5: SQ counters (profiles/r05a_sq_counters.txt) showed the round-4 form issue-bound, not latency-bound: 84 vector and 15 LDS
// instructions per position kept the vector units ~78 % and the LDS ~77 % busy (half of the LDS cycles bank conflicts).  Most of
// them were 64-bit document arithmetic in phase B (position -> document by three table reads and a 64-bit compare, offset, entry,
// symbols left: ~30 instructions per record).  Now the tile keeps, per document, a 32-bit start RELATIVE to the tile and an entry
// bias ((tile base - document start) << bits) + document, and per 32 positions the bit map of document starts with the number of
// starts in front of it: document = count + popcount(bits & mask) — exact for any number of documents in a block, no search —,
// entry = bias + (position << bits), symbols left = next start - position: 9 instructions and two LDS reads.  The code windows
// are ONE unaligned 8-byte LDS read (gfx950 reads the LDS at any byte address), the per-slot tile starts are folded into the
// waves' counters (one read less per position in the scatter), the top digits of the staging thread ignore document ends
// (the few ends are patched afterwards by the lanes that hold one), and the byte behind a thread's 16 comes from its neighbour
// lane instead of a global load.
__global__ __launch_bounds__(512, 8) void rs_sweep_msd_kernel(TextGen gen, uint64_t n, uint32_t tiles, uint32_t* __restrict__ kout,
                                                              uint32_t* __restrict__ vout) {
    constexpr int NT = 512, IPT = 16, TILE = RS_SWEEP_TILE, NW = NT / 64;
    constexpr uint32_t TEXTB = ((TILE + RS_GEN_LOOK + 15) / 16) * 16;
    constexpr uint32_t DOCS = 384;         // documents of a tile kept in the LDS (more: the generic path searches global memory)
    constexpr uint32_t NBLK = TILE / 32;
    __shared__ __attribute__((aligned(16))) uint8_t s_text[TEXTB];
    __shared__ uint8_t s_code[256];
    __shared__ __attribute__((aligned(8))) uint32_t s_doc[2 * DOCS + 2];      // document d of the tile: [2d] = entry bias, [2d + 1] = start of document d + 1 - tile base (clamped)
    __shared__ uint32_t s_whist[NW][256];
    __shared__ __attribute__((aligned(8))) uint32_t s_blk[2 * (NBLK + 1)];    // block b of 32 positions: [2b] = bit p: a document starts at 32 b + p, [2b + 1] = 8 x (starts in front of the block)
    __shared__ uint32_t s_gbase[256];  // (below 2^32 suffixes: output slots are 32-bit, differences modulo 2^32)
    __shared__ uint32_t s_wsum[8];
    __shared__ uint32_t s_flag;        // a document of the tile is empty (two starts on one position): the generic path
    __shared__ __attribute__((aligned(16))) uint16_t s_idx[TILE];
    uint8_t* const s_dig = reinterpret_cast<uint8_t*>(s_idx);  // (its first half, until the ranking is done: the staged top digits)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t x = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint64_t tile = (uint64_t)((slot / RS_GROUP) * 8u + x) * RS_GROUP + slot % RS_GROUP;
    if (tile >= (uint64_t)tiles) return;
    const uint64_t base = tile * TILE;
    const uint32_t valid = (uint32_t)((n - base) < (uint64_t)TILE ? (n - base) : (uint64_t)TILE);

    const uint64_t ga = base + (uint64_t)tid * 16, gb = base + (uint64_t)TILE + (uint64_t)tid * 16;
    const bool has_b = (uint32_t)tid * 16 < TEXTB - (uint32_t)TILE;
    const bool oka = gen.padded ? (ga < n + RS_GEN_LOOK) : (ga + 16 <= n);
    const bool okb = has_b && (gen.padded ? (gb < n + RS_GEN_LOOK) : (gb + 16 <= n));
    uint4 ta = make_uint4(0, 0, 0, 0), tb = make_uint4(0, 0, 0, 0);
    if (oka) ta = *reinterpret_cast<const uint4*>(gen.text + ga);
    if (okb) tb = *reinterpret_cast<const uint4*>(gen.text + gb);
    uint32_t tnext = 0;  // the byte behind the wave's 1024: second symbol of its last pair (the other lanes ask their neighbour)
    if (lane == 63 && ga + 16 < n) tnext = (uint32_t)gen.text[ga + 16];
    uint64_t my_base = 0;
    if (tid < 256) {
        my_base = (uint64_t)gen.tile_base[tile * 256 + (uint64_t)tid];
        s_code[tid] = (uint8_t)gen.symmap[tid];
    }
    const uint64_t dlo = gen.tile_doc[tile], dhi = gen.tile_doc[tile + 1];
    for (int i = tid; i < NW * 256; i += NT) (&s_whist[0][0])[i] = 0;
    if ((uint32_t)tid <= NBLK) s_blk[2 * tid] = 0;
    if (tid == 0) s_flag = 0;
    const uint32_t ndl = (uint32_t)(dhi - dlo);
    const bool docs_in_lds = dhi - dlo + 2 <= (uint64_t)DOCS;
    uint64_t dreg0 = 0, dreg1 = 0;
    if (docs_in_lds) {
        if ((uint32_t)tid < ndl + 2) dreg0 = gen.doc_start[dlo + tid];
        if ((uint32_t)tid + NT < ndl + 2) dreg1 = gen.doc_start[dlo + tid + NT];
    }
    __syncthreads();  // (the zeroes)
    if (docs_in_lds) {
        auto put = [&](uint32_t d, uint64_t ds) {
            const int64_t diff = (int64_t)(ds - base);
            const int32_t rel = diff < -(1ll << 30) ? -(1 << 30) : (diff > (1ll << 30) ? (1 << 30) : (int32_t)diff);
            s_doc[2 * d] = (uint32_t)(((base - ds) << gen.bits) + dlo + d);
            if (d >= 1) s_doc[2 * d - 1] = (uint32_t)rel;  // (beside the entry bias of the document in front: one aligned 8-byte read)
            // a document that starts at s > base ends its predecessor at s - 1; two starts on one position = an empty document
            if (d >= 1 && rel <= (int32_t)TILE) {
                const uint32_t bit = 1u << ((uint32_t)rel & 31u);
                if (atomicOr(&s_blk[2 * ((uint32_t)rel >> 5)], bit) & bit) s_flag = 1;
            }
        };
        if ((uint32_t)tid < ndl + 2) put((uint32_t)tid, dreg0);
        if ((uint32_t)tid + NT < ndl + 2) put((uint32_t)tid + NT, dreg1);
    }
    __syncthreads();

    // ---- staging: bytes -> codes (kept for phase B) and, thread-consecutively, the top digit of every position's first pair
    auto fetch = [&](uint64_t g, bool ok, uint4 w, uint32_t* c) {
        c[0] = w.x; c[1] = w.y; c[2] = w.z; c[3] = w.w;
        if (!ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c[q] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) c[q] |= (uint32_t)((g + 4 * q + b < n) ? gen.text[g + 4 * q + b] : (uint8_t)0) << (8 * b);
            }
        }
    };
    const uint32_t B = gen.base, B2 = B * B;
    {
        uint32_t c[4];
        fetch(ga, oka, ta, c);
        const uint32_t nb = __shfl_down(c[0] & 0xFFu, 1);  // the neighbour thread's first byte
        uint32_t e[IPT + 1];
#pragma unroll
        for (int k = 0; k < IPT; ++k) e[k] = s_code[(c[k >> 2] >> (8 * (k & 3))) & 0xFFu];
        e[IPT] = s_code[lane == 63 ? tnext : nb];
        uint32_t ends;  // bit k: position k of this thread is the last one of its document (position k + 1 starts one)
        if (docs_in_lds) {
            const uint32_t w0 = s_blk[2 * ((uint32_t)tid >> 1)], w1 = s_blk[2 * ((uint32_t)tid >> 1) + 2];
            ends = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (16u * ((uint32_t)tid & 1u) + 1u)) & 0xFFFFu;
        } else {  // (thousands of documents in the tile) the documents that start in (ga, ga + 16]
            ends = 0;
            const uint64_t d = rs_doc_upper(gen.doc_start, dlo, dhi, ga < n ? ga : n - 1);  // document of the thread's first position
            for (uint64_t q = d + 1; q <= gen.ndocs; ++q) {
                const uint64_t st = gen.doc_start[q];
                if (st > ga + 16) break;
                if (st > ga) ends |= 1u << (uint32_t)(st - 1 - ga);
            }
        }
        uint32_t codes[4], digs[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) codes[q] = e[4 * q] | (e[4 * q + 1] << 8) | (e[4 * q + 2] << 16) | (e[4 * q + 3] << 24);
#pragma unroll
        for (int k = 0; k < IPT; ++k)  // (document ends ignored here: patched below by the few lanes that hold one)
            digs[k >> 2] |= (__umul24(__umul24(e[k], B) + e[k + 1], gen.pair_r) >> gen.pair_s) << (8 * (k & 3));
        *reinterpret_cast<uint4*>(&s_text[(uint32_t)tid * 16]) = make_uint4(codes[0], codes[1], codes[2], codes[3]);
        *reinterpret_cast<uint4*>(&s_dig[(uint32_t)tid * 16]) = make_uint4(digs[0], digs[1], digs[2], digs[3]);
        while (ends) {  // a document's last position: its second symbol counts as 0
            const uint32_t k = (uint32_t)__builtin_ctz(ends);
            ends &= ends - 1u;
            const uint32_t cc = s_text[(uint32_t)tid * 16 + k];
            s_dig[(uint32_t)tid * 16 + k] = (uint8_t)(__umul24(__umul24(cc, B), gen.pair_r) >> gen.pair_s);
        }
        if (has_b) {  // the look-ahead behind the tile: codes only
            fetch(gb, okb, tb, c);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                c[q] = (uint32_t)s_code[c[q] & 0xFF] | ((uint32_t)s_code[(c[q] >> 8) & 0xFF] << 8) | ((uint32_t)s_code[(c[q] >> 16) & 0xFF] << 16) |
                       ((uint32_t)s_code[c[q] >> 24] << 24);
            *reinterpret_cast<uint4*>(&s_text[(uint32_t)TILE + (uint32_t)tid * 16]) = make_uint4(c[0], c[1], c[2], c[3]);
        }
    }
    __syncthreads();
    // ---- phase A: ranks, lane-striped (text order inside every bucket, see rs_sweep_records_kernel)
    constexpr int WCHUNK = IPT * 64;
    const uint32_t wbase = wave * WCHUNK + lane;
    uint32_t info[IPT];  // rank | digit << 16; ~0 behind the text
    {
        uint32_t sl[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) sl[j] = s_dig[wbase + j * 64];
        if (valid == (uint32_t)TILE) {
#pragma unroll
            for (int j = 0; j < IPT; ++j) info[j] = atomicAdd(&s_whist[wave][sl[j]], 1u) | (sl[j] << 16);
        } else {
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                uint32_t inf = ~0u;
                if (wbase + j * 64 < valid) inf = atomicAdd(&s_whist[wave][sl[j]], 1u) | (sl[j] << 16);
                info[j] = inf;
            }
        }
    }
    __syncthreads();
    // ---- per-digit totals of the tile (waves 0-3), and meanwhile (waves 4-7) the number of document starts in front of every block
    uint32_t wc[NW], cnt = 0, incl = 0;
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            wc[w] = s_whist[w][tid];
            cnt += wc[w];
        }
        incl = cnt;
    } else {
        incl = cnt = (uint32_t)__builtin_popcount(s_blk[2 * (tid - 256)]);
    }
    incl = rs_wave_incl_scan(incl);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    {
        uint32_t wpre = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (w < (wave & 3)) wpre += s_wsum[(wave & 4) + w];
        const uint32_t excl = wpre + incl - cnt;
        if (tid < 256) {  // every wave's counter of the digit becomes its first slot in the tile's sorted order
            uint32_t run = excl;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                s_whist[w][tid] = run;
                run += wc[w];
            }
            s_gbase[tid] = (uint32_t)my_base - excl;
        } else {
            s_blk[2 * (tid - 256) + 1] = excl * 8u;
            if (tid == 511) s_blk[2 * NBLK + 1] = (excl + cnt) * 8u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        if (info[k] != ~0u) {
            const uint32_t pos = s_whist[wave][info[k] >> 16] + (info[k] & 0xFFFFu);
            s_idx[pos] = (uint16_t)(wbase + k * 64);  // (the digit is not carried along: phase B has it back from the first pair)
        }
    }
    __syncthreads();

    // ---- phase B: output order; key - top M = ((a - top span) B^2 + m) B^2 + r with a, m, r = the three symbol pairs (one
    // v_dot4_u32_u8 each; products below 2^24, rs_pair_setup), entry = (offset << bits) | document
    const uint32_t wlo = B | (1u << 8), whi = (B << 16) | (1u << 24);
    const int nspan = -(int)gen.pair_span;
    auto emit = [&](uint32_t p, uint32_t li, uint32_t ent, uint32_t rem) {
#if RS_SWEEP_UNALIGNED
        uint2 xw;
        __builtin_memcpy(&xw, &s_text[li], 8);  // codes of li .. li + 7: one LDS read at a byte address
        uint32_t x0 = xw.x, x1 = xw.y;
#else
        const uint32_t* s_words = reinterpret_cast<const uint32_t*>(s_text);
        const uint32_t wi = li >> 2, sel = li & 3u;
        const uint32_t w0 = s_words[wi], w1 = s_words[wi + 1], w2 = s_words[wi + 2];
        uint32_t x0 = __builtin_amdgcn_alignbyte(w1, w0, sel);  // codes of li .. li + 3
        uint32_t x1 = __builtin_amdgcn_alignbyte(w2, w1, sel);  // codes of li + 4 .. li + 7
#endif
        if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(rem < 6u) != 0)) != 0, 0)) {  // (wave-uniform; a quarter of the waves at 1 KiB documents)
            if (rem < 6u) {  // the symbols behind the document end count as 0
                x0 = rem >= 4u ? x0 : (x0 & ((1u << (8u * rem)) - 1u));
                x1 = rem <= 4u ? 0u : (x1 & 0xFFu);
            }
        }
        const uint32_t a = __builtin_amdgcn_udot4(x0, wlo, 0u, false);
        const uint32_t m = __builtin_amdgcn_udot4(x0, whi, 0u, false);
        const uint32_t r = __builtin_amdgcn_udot4(x1, wlo, 0u, false);
        const uint32_t top = (uint32_t)__umul24(a, gen.pair_r) >> gen.pair_s;  // floor(a / span): the digit the position was ranked on
        const uint32_t a2 = (uint32_t)(__mul24((int)top, nspan) + (int)a);
        const uint32_t dst = s_gbase[top] + p;
        kout[dst] = (uint32_t)__umul24((uint32_t)__umul24(a2, B2) + m, B2) + r;
        vout[dst] = ent;
    };
    if (docs_in_lds && s_flag == 0 && valid == (uint32_t)TILE) {
        constexpr int U = 2;  // (two records per trip: their loads stage by stage in flight together)
        const int bits = gen.bits;
#pragma nounroll
        for (uint32_t p0 = tid; p0 < (uint32_t)TILE; p0 += U * NT) {
            uint32_t li[U];
            uint2 bk[U], dc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) li[u] = s_idx[p0 + u * NT];
#pragma unroll
            for (int u = 0; u < U; ++u) bk[u] = *reinterpret_cast<const uint2*>(&s_blk[2 * (li[u] >> 5)]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // document = starts in front of the block + starts inside it up to the position
                const uint32_t doff = bk[u].y + 8u * (uint32_t)__builtin_popcount(bk[u].x & ~(0xFFFFFFFEu << (li[u] & 31u)));
                dc[u] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(s_doc) + doff);  // entry bias, next start
            }
#pragma unroll
            for (int u = 0; u < U; ++u) emit(p0 + u * NT, li[u], dc[u].x + (li[u] << bits), dc[u].y - li[u]);
        }
    } else {
        // (the ragged last tile, tiles with hundreds of documents or empty ones: searches in global memory)
        for (uint32_t p = tid; p < valid; p += NT) {
            const uint32_t li = s_idx[p];
            const uint64_t pos = base + li;
            const uint64_t dd = rs_doc_upper(gen.doc_start, dlo, dhi, pos);
            const uint64_t ds = gen.doc_start[dd], de = gen.doc_start[dd + 1];
            const uint64_t left = de - pos;
            emit(p, li, (uint32_t)(((pos - ds) << gen.bits) + dd), left < 64ull ? (uint32_t)left : 64u);
        }
    }
}

// whether the sweep's arithmetic applies: codes are bytes weighted by B in a dot4 (B <= 255), the key's nsym - 1 symbols come
// from three code windows (<= 10 symbols: two or three 4-byte windows)
inline bool rs_sweep_records_ok(uint32_t base, int nsym) { return base <= 255u && nsym >= 2 && nsym - 1 <= 10; }
// constants of the quantiser floor(code * m / base) = (code * r) >> s for every code < base; false: none found (m stays 1)
template <typename G>
inline bool rs_part_setup(G& gen, uint32_t base, uint32_t m) {
    gen.part_m = 1;
    gen.part_r = gen.part_s = 0;
    if (m < 2 || base < 2 || base > 256) return false;
    for (uint32_t sh = 0; sh < 24; ++sh) {
        const uint64_t r = (((uint64_t)m << sh) + base - 1) / base;
        if (r >= (1u << 24)) break;
        bool ok = true;
        for (uint64_t c = 0; c < base && ok; ++c) ok = ((c * r) >> sh) == c * m / base;
        if (ok) {
            gen.part_m = m;
            gen.part_r = (uint32_t)r;
            gen.part_s = sh;
            return true;
        }
    }
    return false;
}

// One group: records of the buckets [g0, g1) into (k, v, w) at group-local indices, then the digit histograms of every bucket's
// passes from the records (d_hist_out: [nseg][8][256], zeroed here).  d_tile_doc: [tiles8 + 1] (rs_tiledoc_kernel over
// RS_SWEEP_TILE); gen_in.tile_base: array-wide tile bases of all bucket slots (rs_tile_bases).
template <typename W>
void radix_sweep_records(hipStream_t s, Profiler& prof, uint32_t* k, uint32_t* v, W* w, uint64_t n, const TextGen& gen_in, const uint16_t* d_codeslot,
                         uint32_t g0, uint32_t g1,
                         uint64_t gstart, uint64_t gelems, const uint32_t* d_tile_seg, const SegInfo* d_segs, uint32_t nseg, uint32_t seg_tiles,
                         int lead, int npass, unsigned long long* d_hist_out, SortStats* stats) {
    if (!gen_in.tile_base || !gen_in.tile_doc || !d_codeslot || !gen_in.slotmap) throw Error("radix_sweep_records: tile bases / documents / slots missing (internal)");
    if (!rs_sweep_records_ok(gen_in.base, gen_in.nsym)) throw Error("radix_sweep_records: key shape (internal)");
    const uint32_t tiles8 = (uint32_t)ceil_div(n, (uint64_t)RS_SWEEP_TILE);
    const uint32_t grid = (uint32_t)(ceil_div(tiles8, 8u * RS_GROUP) * 8u * RS_GROUP);
    CDB_HIP(hipMemsetAsync(d_hist_out, 0, (size_t)nseg * 8 * 256 * sizeof(uint64_t), s));
    int t = prof.begin(s);
    hipLaunchKernelGGL((rs_sweep_records_kernel<W>), dim3(grid), dim3(512), 0, s, gen_in, d_codeslot, n, tiles8, g0, g1, gstart, k, v, w);
    prof.end(t, (std::string("rs_sweep_records") + (sizeof(W) == 1 ? "_w8" : (sizeof(W) == 2 ? "_w16" : "_w32")) + "_t8192").c_str(),
             n + gelems * (8 + sizeof(W)), s);
    if (stats) stats->passes_run++;
    t = prof.begin(s);
    hipLaunchKernelGGL((rs_seg_hist_kernel<W>), dim3((unsigned)ceil_div(seg_tiles, (uint32_t)RS_MSD_HIST_TILES)), dim3(1024), 0, s, (const uint32_t*)k,
                       (const W*)w, lead, npass, d_tile_seg, d_segs, seg_tiles, d_hist_out);
    prof.end(t, "rs_seg_hist", gelems * (4 + (lead > 0 ? sizeof(W) : 0)), s);
    CDB_HIP(hipGetLastError());
}

// the same with variable-length keys (rs_sweep_records_vl_kernel): gen_in.base / nsym are not used, gen_in.rec_low_bits and the passes as above
template <typename W>
void radix_sweep_records_vl(hipStream_t s, Profiler& prof, uint32_t* k, uint32_t* v, W* w, uint64_t n, const TextGen& gen_in, const uint16_t* d_codeslot,
                            const VlTables& vl, uint32_t g0, uint32_t g1, uint64_t gstart, uint64_t gelems, const uint32_t* d_tile_seg,
                            const SegInfo* d_segs, uint32_t nseg, uint32_t seg_tiles, int lead, int npass, unsigned long long* d_hist_out,
                            SortStats* stats) {
    if (!gen_in.tile_base || !gen_in.tile_doc || !d_codeslot || !vl.sym || !vl.dec) throw Error("radix_sweep_records_vl: tables missing (internal)");
    if (vl.key_bits < 16 || vl.key_bits > 56) throw Error("radix_sweep_records_vl: key width (internal)");
    const uint32_t tiles8 = (uint32_t)ceil_div(n, (uint64_t)RS_SWEEP_TILE);
    const uint32_t grid = (uint32_t)(ceil_div(tiles8, 8u * RS_GROUP) * 8u * RS_GROUP);
    CDB_HIP(hipMemsetAsync(d_hist_out, 0, (size_t)nseg * 8 * 256 * sizeof(uint64_t), s));
    int t = prof.begin(s);
    hipLaunchKernelGGL((rs_sweep_records_vl_kernel<W>), dim3(grid), dim3(512), 0, s, gen_in, d_codeslot, vl, n, tiles8, g0, g1, gstart, k, v, w);
    prof.end(t, (std::string("rs_sweep_records_vl") + (sizeof(W) == 1 ? "_w8" : (sizeof(W) == 2 ? "_w16" : "_w32")) + "_t8192").c_str(),
             n + gelems * (8 + sizeof(W)), s);
    if (stats) stats->passes_run++;
    t = prof.begin(s);
    hipLaunchKernelGGL((rs_seg_hist_kernel<W>), dim3((unsigned)ceil_div(seg_tiles, (uint32_t)RS_MSD_HIST_TILES)), dim3(1024), 0, s, (const uint32_t*)k,
                       (const W*)w, lead, npass, d_tile_seg, d_segs, seg_tiles, d_hist_out);
    prof.end(t, "rs_seg_hist", gelems * (4 + (lead > 0 ? sizeof(W) : 0)), s);
    CDB_HIP(hipGetLastError());
}

}  // namespace cdb
