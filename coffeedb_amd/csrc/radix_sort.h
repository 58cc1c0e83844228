// radix_sort.h — stable LSD radix sort for gfx950 (digits of up to 8 bits, one read + one write of the
// data per pass: "onesweep" with chained-scan decoupled look-back).
//
// This is the bandwidth-dominant primitive of the suffix-array build (replaces the reference's
// multithreaded MSD radix + std::sort leaves, /root/reference/src/index.cpp:75-128) and of the
// hit->document grouping in the query path (replaces index.cpp:294-315).
//
// Per pass and per element the kernel moves sizeof(K)+sizeof(V) (+ sizeof(W) for an auxiliary low-digit array)
// bytes in and the same out; that is the "algorithmic bytes" figure used for the HBM roofline.
//
// Hardware mapping:
//   * 64-wide wavefronts: the rank of an element among the equal digits of its wave comes from ONE returning
//     LDS atomic on the wave's private counters (ATOMRANK; stable because same-address LDS atomics of one
//     instruction complete in lane order on gfx950 — verified per device by rs_lane_order_probe_kernel), or
//     from 8 ballots, one per digit bit, when that self-test fails;
//   * LDS: keys, then values, of one tile are staged in sorted-by-digit order so that the global
//     write-out has consecutive lanes writing consecutive addresses inside each digit run; the
//     production tile is 1024 threads x 16 keys (150 KB of the CU's 160 KB LDS, one workgroup per CU);
//   * 8 XCDs with non-coherent L2s: tiles exchange {epoch,state,count} words only through agent-scope
//     relaxed atomic loads/stores (one 8-byte granule is both flag and payload — MI355X guide §G16 R2),
//     tile ids come from an atomic ticket so a tile only ever waits on tiles that already started, and
//     every spin is bounded (a stuck look-back raises an error flag instead of hanging the GPU).
#pragma once
#include <cstdlib>

#include "common.h"

namespace cdb {

constexpr int RS_MAX_PASSES = 16;

constexpr uint64_t RS_VAL_MASK = (1ull << 54) - 1;
// bounded look-back spins.  XCD-ordered (grouped) passes reserve tiles for workgroups that have not started: a predecessor that has
// not answered after ~0.3-0.5 s of polling is not coming (starved pass; the build retries in plain ticket order), and 2^22 held a
// starved build for ~8 s per wait.  In plain ticket order a tile only waits for tiles that already RUN, so a long wait means the
// predecessor was descheduled (two processes on one GPU, CWSR preemption, a profiler): that pass keeps the patient bound — nothing
// retries it.
constexpr uint32_t RS_SPIN_LIMIT_GROUPED = 1u << 18;
constexpr uint32_t RS_SPIN_LIMIT_PLAIN = 1u << 22;
// XCD-aware tile order of the big-tile configurations: groups of RS_GROUP consecutive tiles go to one XCD (RsCfg::GROUP).
// Tiles are reserved for workgroups that have not started yet, so up to 7 * RS_GROUP resident workgroups can wait for
// one that is still to be dispatched: the pass needs more than that many resident at a time (it has 256 when it runs
// alone; concurrent kernels can take some away) — a pass that starves raises the look-back error flag (bounded
// spins), and the callers redo the sort in plain ticket order (variant 33), which needs one.
constexpr int RS_GROUP = 8;

struct NoVal {};

// keys per thread of the small 256-thread configurations (cases 1 and 4 of radix_sort)
template <typename K, typename V> struct RsTraits { static constexpr int IPT = 12; };
template <> struct RsTraits<uint64_t, uint32_t> { static constexpr int IPT = 15; };
template <> struct RsTraits<uint64_t, uint64_t> { static constexpr int IPT = 12; };

// ---------------------------------------------------------------------------------------------
// upfront histogram of every pass's digit (one read of the keys)
// ---------------------------------------------------------------------------------------------
template <typename K>
__global__ __launch_bounds__(256) void rs_hist_kernel(const K* __restrict__ keys, uint64_t n, int begin_bit,
                                                      int npass, int dbits, uint32_t last_mask,
                                                      unsigned long long* __restrict__ ghist) {
    __shared__ uint32_t sh[RS_MAX_PASSES * 256];
    for (int i = threadIdx.x; i < npass * 256; i += 256) sh[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const K k = keys[i];
#pragma unroll
        for (int p = 0; p < RS_MAX_PASSES; ++p) {
            if (p < npass) {
                uint32_t d = (uint32_t)(k >> (begin_bit + dbits * p)) & ((1u << dbits) - 1u);
                if (p == npass - 1) d &= last_mask;
                // (keys that arrive nearly sorted — the hit keys of a query batch come in pattern order — agree on their high
                //  digits: 64 same-address atomics serialise, one add of the lane count does not.  8 GiB Zipf batch: 0.92 -> ms below)
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                const uint64_t same = __builtin_amdgcn_ballot_w64(d == d0);
                const uint64_t act = __builtin_amdgcn_ballot_w64(true);
                if (same == act) {
                    if ((uint32_t)__builtin_ffsll((long long)act) - 1u == (threadIdx.x & 63u)) atomicAdd(&sh[p * 256 + d0], (uint32_t)__popcll(act));
                } else {
                    atomicAdd(&sh[p * 256 + d], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npass * 256; i += 256)
        if (sh[i]) atomicAdd(&ghist[i], (unsigned long long)sh[i]);
}

// exclusive scan of each pass's 256 counts -> first output slot of each digit
static __global__ __launch_bounds__(256) void rs_digit_start_kernel(const unsigned long long* __restrict__ ghist,
                                                             unsigned long long* __restrict__ gstart) {
    __shared__ unsigned long long s[256];
    const int p = blockIdx.x, t = threadIdx.x;
    const unsigned long long c = ghist[p * 256 + t];
    s[t] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        unsigned long long v = t >= off ? s[t - off] : 0;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    gstart[p * 256 + t] = s[t] - c;
}

__device__ __forceinline__ uint64_t rs_ld_status(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rs_st_status(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// one radix pass: rank inside the tile, look back for the global prefix, scatter
// ---------------------------------------------------------------------------------------------
// Kernel configuration.  IPT = keys per thread (tile = 256 * IPT); REUSE = keys and values share one
// LDS staging buffer (two write-out phases, smaller footprint -> more workgroups per CU); EARLYV =
// values are fetched together with the keys instead of after the look-back.
template <int IPT_, bool REUSE_, bool EARLYV_, int NT_ = 256, bool NONTEMP_ = false, int MINW_ = 1, int ABL_ = 0,
          int LB_ = 1, bool DMA_ = false, bool ATOMRANK_ = false, bool TICKET_ = true, int LOAD_ = 0, int GROUP_ = 0>
struct RsCfg {
    static constexpr int GROUP = GROUP_;      // > 0: XCD-aware tile order — workgroup b draws from the ticket counter of
                                              // class b % 8 (the XCD it runs on, as observed) and class x owns the tile
                                              // groups x, x + 8, ... of GROUP consecutive tiles each, so neighbouring
                                              // tiles (whose digit runs touch) are written through the same L2
    static constexpr int LOAD = LOAD_;        // plain load path: 0 = predicated loads, keys -> values -> auxiliary digits;
                                              // 1 = unpredicated, keys -> digits -> values (values in flight under the
                                              // ranking); 2 = unpredicated, keys -> values -> digits
    static constexpr bool TICKET = TICKET_;   // tile id from an atomic ticket (false: blockIdx.x — relies on in-order
                                              // dispatch of workgroups; every spin is bounded either way)
    static constexpr bool ATOMRANK = ATOMRANK_;  // rank inside the wave with one returning LDS atomic per element
                                              // instead of 8 ballots (relies on same-address LDS atomics of one
                                              // instruction completing in lane order — checked by a self-test)
    static constexpr bool DMA = DMA_;         // keys/values reach the registers through LDS by 16-byte global->LDS DMA
    static constexpr int LB = LB_;            // look-back window: predecessors polled per round trip
    static constexpr int ABL = ABL_;          // timing-only ablations (WRONG results): 1 = no look-back,
                                              // 2 = linear (unscattered) write-out, 8 = look-back depth counters
    static constexpr int MINW = MINW_;        // min waves per SIMD the register allocator must allow
    static constexpr int IPT = IPT_;
    static constexpr bool REUSE = REUSE_;
    static constexpr bool EARLYV = EARLYV_;
    static constexpr int NT = NT_;            // threads per workgroup (>= 256, multiple of 64)
    static constexpr bool NONTEMP = NONTEMP_;  // streaming (non-temporal) global loads/stores
};

template <bool NT_, typename T> __device__ __forceinline__ T rs_load(const T* p) {
    if constexpr (NT_) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT_, typename T> __device__ __forceinline__ void rs_store(T* p, T v) {
    if constexpr (NT_) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// Optional producer for the FIRST pass of the suffix-array sort: instead of reading (key, entry) pairs
// that a separate kernel would have to write first, the pass computes them from the text on the fly —
// key = the suffix's first `nsym` symbol codes (0 = end of document) as a number in base `base`, entry =
// (off << bits) | doc.  base = 2^symbits gives bit-aligned symbols; base = alphabet + 1 the densest key.
// Saves one write and one read of 8+w bytes per suffix plus the key read of the histogram kernel.
struct NoGen {};
struct TextGen {
    const uint8_t* text;
    const uint64_t* doc_start;
    const uint16_t* symmap;
    uint64_t ndocs;
    int bits;
    uint32_t base;
    int nsym;
    int low_bits = 0;  // split keys: the generated pass sorts on the key's low_bits lowest bits, which then travel
                       // as a separate byte per element; the remaining passes see key >> low_bits (32-bit keys)
    bool padded = false;  // text buffer has >= RS_GEN_LOOK + 16 readable bytes behind n
    bool first_only = false;  // entries-only partition by the first symbol: the key is just that symbol's digit
    int msd_shift = 0;  // > 0 (split records, MSD-first sort): the generated pass sorts on the key's TOP digit, key >> msd_shift,
                        // which is then implied by the bucket an element sits in — the records are (u32 key, entry) with no
                        // auxiliary byte at all, and the remaining passes sort every bucket on its own (radix_sort_msd)
    // Bucket-wise build, fused form (sa_build.hip): the generated pass writes the bucket RECORDS itself instead of partitioning
    // the entries for a later gather — sort digit = bucket slot of the suffix's first symbol (slotmap[code]), key = the
    // nsym - 1 symbols behind it as a dense number (k32 = key >> rec_low_bits), W = its low digits | entry bits 32.. above
    // them, value = entry bits 0..31.  16 Ki-key tiles only (thread-consecutive rolling keys); kernels of type TextGenRec.
    const uint8_t* slotmap = nullptr;  // [257] symbol code -> bucket slot
    int rec_low_bits = 0;
    // Leftover key bits (round 5, sweep form only): a key of nsym - 1 symbols in whole 8-bit passes leaves room for part_m >= 2 more
    // values below it — they hold the NEXT symbol quantised to part_m levels, floor(code * part_m / base) = (code * part_r) >>
    // part_s (monotone in the code, so the key stays order-preserving): key = dense * part_m + level.  part_m = 1: none.
    uint32_t part_m = 1, part_r = 0, part_s = 0;
    // MSD-first sort, pair form (6-symbol keys; kernels of type TextGenPair): top digit = (first two symbols as a number A) /
    // span, i.e. key / M with M = span * base^4 <= 2^32 — a function of two symbols, so its histogram comes from a pair count
    // of the text instead of a sweep that evaluates every key.  The generated pass works in 32-bit part arithmetic (G = three
    // symbols as a number, key = G(p) base^3 + G(p + 3)), lane-striped: every element straight from the staged codes.
    bool msd_pair = false;
    uint32_t pair_span = 0, pair_r = 0, pair_s = 0;  // floor(a / span) = (a * pair_r) >> pair_s for every a < base^2 (rs_pair_setup)
    const uint64_t* tile_doc = nullptr;  // [tiles + 1] document of each tile's first position (set by the driver)
    // Look-back-free form (round 4): tile_base[tile * 256 + d] = the output slot of the tile's first element of digit d, from a
    // counting pre-pass over the text + a scan over the tiles (the digit of a generated pass is a function of one or two symbols, so
    // its per-tile counts cost one cheap sweep).  The pass then needs no status words, no tile order and waits for nobody — which
    // is what lets two 8 Ki-key workgroups share a CU (with the chained scan, twice the tiles cost more than the overlap gained)
    const unsigned long long* tile_base = nullptr;
    uint8_t* vout_hi = nullptr;  // 8-byte values (first_only partition): write them packed — low words to (uint32_t*)vout, bits 32..39 here
};
// The generator's FORM is part of the kernel's type: the 16 Ki-tile pass is ~15 k instructions of straight-line code per
// form (everything is unrolled 16 x), and a kernel that carries all three runs 4 % slower than one that carries its own
// (tools/experiments/gen_bench.hip, -DRS_GEN_MODES: 5.55 vs 5.31 ms).  TextGen itself = the rolling-key form.
struct TextGenPair : TextGen {};  // MSD-first sort, pair form (msd_pair)
struct TextGenRec : TextGen {};   // bucket records (rec_mode), thread-consecutive rolling keys
constexpr int RS_GEN_LOOK = 64;
// timing-only ablations of the generated pass (tools/experiments/gen_bench.hip; WRONG results): 1 = no key arithmetic,
// 2 = no transposition through LDS, 4 = no text staging
#ifndef RS_GEN_ABL
#define RS_GEN_ABL 0
#endif
// ... and of the final pass of a segmented sort (WRONG results): 1 = no flag stores, 2 = no entry stores, 4 = no neighbour
// compares.  Compile-time (make HIPFLAGS+=-DRS_SEG_ABL=n): the product kernel carries no checks for them.
#ifndef RS_SEG_ABL
#define RS_SEG_ABL 0
#endif



// floor(x / d) for x < 2^24 as one multiply-high: with L = ceil(log2 d) and m = ceil(2^(24+L) / d) (< 2^25, error
// m d - 2^(24+L) < d <= 2^L, so x < 2^24 keeps the quotient exact), mul = m << 7 and sh = L + 7
struct RsDiv24 { uint32_t mul, sh; };
inline RsDiv24 rs_div24_make(uint32_t d) {
    uint32_t L = 0;
    while ((1ull << L) < d) ++L;
    const uint64_t m = ((1ull << (24 + L)) + d - 1) / d;
    return RsDiv24{(uint32_t)(m << 7), L + 7};
}
__device__ __forceinline__ uint32_t rs_div24(uint32_t x, uint32_t mul, uint32_t sh) { return __umulhi(x << 8, mul) >> sh; }

// Pair form of the generated pass (TextGenPair): constants for top = floor(a / span), a < base^2, as one 24-bit multiply and a
// shift, and the range conditions of its 24-bit products.  false = this (base, span) cannot take the pair form.
template <typename G>
inline bool rs_pair_setup(G& gen, uint32_t base, uint32_t span) {
    const uint64_t b2 = (uint64_t)base * base;
    if (base < 2 || base > 255 || span == 0 || span >= (1u << 24) || b2 >= (1u << 24)) return false;
    if ((uint64_t)span * b2 >= (1u << 24)) return false;              // (a - top span) B^2 + m stays below 2^24
    if ((uint64_t)span * b2 * b2 > (1ull << 32)) return false;         // key - top M < M <= 2^32
    for (uint32_t sh = 0; sh < 32; ++sh) {
        const uint64_t r = ((1ull << sh) + span - 1) / span;
        if (r >= (1u << 24) || (b2 - 1) * r >= (1ull << 32)) break;
        bool ok = true;
        for (uint64_t a = 0; a < b2 && ok; ++a) ok = ((a * r) >> sh) == a / span;
        if (ok) {
            gen.pair_span = span;
            gen.pair_r = (uint32_t)r;
            gen.pair_s = sh;
            return true;
        }
    }
    return false;
}

// Segmented passes (the bucket-wise build of corpora >= 2^32, sa_build.hip): ONE launch sorts every first-symbol
// bucket ("segment") of a bucket group on its own — a tile belongs to exactly one segment (tile_seg), takes its digit
// starts from that segment's table, and the look-back chain restarts at the segment's first tile.  Replaces one
// launch per bucket and pass (~1000 launches per 4 GiB build) by one launch per pass.
struct NoSeg {};
struct SegInfo {
    unsigned long long begin, end;  // element range of the segment inside the group's record buffers
    uint32_t tile_begin, top;       // its first tile; MSD-first sort: the top digit every key of the segment shares
    // final pass only: the sorted segment [0, a) [a, b) [b, len) is written as [0, a) [b, len) [a, b) — the reference's child
    // order inside a radix node (end of document, bytes 0x80..0xFF, bytes 0x00..0x7F; index.h:66-73).  a = b = 0: as sorted
    unsigned long long rot_a, rot_b;
};
// where slot r of a segment goes under that block swap
__device__ __forceinline__ unsigned long long rs_seg_rotated(const SegInfo& si, unsigned long long slot) {
    if (si.rot_b == 0) return slot;
    const unsigned long long r = slot - si.begin, len = si.end - si.begin;
    if (r < si.rot_a) return slot;
    return r < si.rot_b ? slot + (len - si.rot_b) : slot - (si.rot_b - si.rot_a);
}
struct SegArgs {
    const uint32_t* tile_seg = nullptr;  // [tiles] segment of every tile
    const SegInfo* segs = nullptr;
    uint32_t tiles = 0;                  // tiles of all segments together
    uint32_t start_stride = 0;           // digit starts of segment g at digit_start[g * start_stride + d]
};
// The LAST pass of a segmented sort of packed bucket records writes the finished suffix-array entries and their
// group flags instead of the records: entry = (aux >> hi_shift) << 32 | value, flag bit0 = first of a group of equal
// keys, bit1 = still unresolved (sa_build.hip: sa_flags_of).  A tile sees the neighbours of all its elements except
// across the ends of its per-digit runs, whose true neighbours sit in other tiles: those elements get provisional
// flags (head / tail assumed), every (tile, digit) run leaves an edge record, and rs_seg_edge_fix_kernel settles them.
struct SegEdge {
    unsigned long long first, last;  // full keys of the run's first and last element
    unsigned long long pos;          // output slot (inside the group) of the run's first element
    uint32_t cnt, pad;
};
struct SegFinalArgs : SegArgs {
    uint64_t* eout = nullptr;   // entries of the group (already offset to the group's first slot)
    uint32_t* elo = nullptr;    // ... or, packed storage (index_impl.h: Sa40): their low words and
    uint8_t* ehi = nullptr;     //     their bits 32..39 (elo != nullptr selects this form)
    uint8_t* flags = nullptr;   // ... and its flags
    SegEdge* edges = nullptr;   // [tiles][256]
    int hi_shift = 0;           // entry bits 32.. sit above this many bits of the auxiliary word
    int low_bits = 0;           // key = (k32 << low_bits) | (aux & (2^low_bits - 1))
    uint32_t kbase = 2;
    unsigned long long kmagic = 0;
};

// The last pass of an ordinary (single-segment) sort of split records can do the same and still write its records (the
// kept search keys): the suffix-array build below 2^32 then needs no flag kernel (5-6 B read + 1 B written per suffix).
// tile_sums (optional): per scan tile of `sums_tile` flags {unresolved entries, unresolved group heads} as pairs of u64 —
// what the first compaction of the refinement wants; counted here (unresolved entries are rare), corrected by the edge fix.
struct SegFinalKeepArgs : SegArgs {
    uint8_t* flags = nullptr;
    SegEdge* edges = nullptr;
    int low_bits = 0;
    uint32_t kbase = 2;
    unsigned long long kmagic = 0;
    unsigned long long* tile_sums = nullptr;
    uint32_t sums_tile = 1;
    // MSD-first sort (radix_sort_msd): the records have no auxiliary array (win == nullptr), the full key of an element is
    // (segment's top digit << msd_shift) | k32, and the pass writes the kept search keys in the layout of the LSD split sort —
    // (u32)(full >> 8) and the low byte — so that everything downstream of the sort is the same
    unsigned long long msd_m = 0;  // full key = top * msd_m + k32 (0: not an MSD-first sort)
};

// ... as a type of its own for the last pass of an MSD-first sort (no auxiliary input, full key = top * msd_m + k32): the
// auxiliary loads, their LDS traffic and the byte compares of the LSD form are not compiled into it
struct SegFinalKeepMsdArgs : SegFinalKeepArgs {};

// Key of the suffix at tile-local position li from symbol CODES staged in LDS (dword view, code of position i
// in byte i): Horner over the first nsym (<= 16) codes in base `base`, first symbol most significant; codes
// at or behind the end of the document (rem symbols left) count as 0.  Four codes at a time are combined
// with 32-bit operations (base <= 256, so four of them stay below 2^32).
__device__ __forceinline__ uint64_t rs_pack_key(const uint32_t* __restrict__ s_words, uint32_t li, int nsym, uint32_t base,
                                                uint32_t rem) {
    const uint32_t wi = li >> 2, sel = li & 3u;
    const uint32_t b2 = base * base, b3 = b2 * base;
    const uint32_t pw[5] = {1u, base, b2, b3, b3 * base};  // pw[4] wraps to 0 for base = 256: handled below
    uint64_t kk = 0;
    uint32_t lo = s_words[wi];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int q = 4 * w;
        if (q >= nsym) break;                           // uniform
        const int r = nsym - q < 4 ? nsym - q : 4;      // symbols taken from this window (uniform)
        const uint32_t hi = s_words[wi + w + 1];
        uint32_t x = __builtin_amdgcn_alignbyte(hi, lo, sel);  // codes of li + q .. li + q + 3
        lo = hi;
        if (rem < (uint32_t)(q + 4)) x = rem <= (uint32_t)q ? 0u : (x & ((1u << (8u * (rem - (uint32_t)q))) - 1u));
        uint32_t v = x & 0xFFu;
        if (r > 1) v = v * base + ((x >> 8) & 0xFFu);
        if (r > 2) v = v * base + ((x >> 16) & 0xFFu);
        if (r > 3) v = v * base + (x >> 24);
        kk = (r == 4 && pw[4] == 0u) ? ((kk << 32) | v) : kk * pw[r] + v;
    }
    return kk;
}

__device__ __forceinline__ uint64_t rs_doc_upper(const uint64_t* __restrict__ doc_start, uint64_t lo, uint64_t hi,
                                                 uint64_t p) {
    while (lo < hi) {  // largest d in [lo, hi] with doc_start[d] <= p
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (doc_start[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// document holding the first position of every tile (tile_doc[tiles] = the last document): turns the
// two ~log2(D)-step searches per tile into one parallel pre-pass
static __global__ __launch_bounds__(256) void rs_tiledoc_kernel(const uint64_t* __restrict__ doc_start, uint64_t ndocs,
                                                                uint64_t n, uint64_t tile_elems, uint64_t tiles,
                                                                uint64_t* __restrict__ tile_doc) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t > tiles) return;
    const uint64_t p = t * tile_elems < n ? t * tile_elems : n - 1;
    tile_doc[t] = rs_doc_upper(doc_start, 0, ndocs - 1, p);
}

// W (optional, u8 or u16) = auxiliary low digits of a split key that travel with the pair (TextGen::low_bits);
// aux_shift >= 0 makes this pass sort on (aux >> aux_shift) & dmask instead of a key digit (the leading
// passes of a split sort: the first, generated one, and for two low digits the one after it).
template <typename K, typename V, typename Cfg, typename Gen = NoGen, typename W = NoVal, typename Seg = NoSeg>
__global__ __launch_bounds__(Cfg::NT, Cfg::MINW) void rs_onesweep_kernel(
    const K* __restrict__ kin, K* __restrict__ kout, const V* __restrict__ vin, V* __restrict__ vout, uint64_t n,
    int shift, uint32_t dmask, const unsigned long long* __restrict__ digit_start, uint64_t* __restrict__ status,
    uint32_t* __restrict__ ticket, uint32_t epoch, uint32_t* __restrict__ err, Gen gen = Gen(),
    const W* __restrict__ win = nullptr, W* __restrict__ wout = nullptr, int aux_shift = -1, Seg seg = Seg()) {
    constexpr bool GEN = !std::is_same<Gen, NoGen>::value;
    constexpr bool SEG = !std::is_same<Seg, NoSeg>::value;
    constexpr bool FINAL = std::is_same<Seg, SegFinalArgs>::value;
    constexpr bool KEEPM = std::is_same<Seg, SegFinalKeepMsdArgs>::value;  // ... of an MSD-first sort (no auxiliary input)
    constexpr bool KEEP = std::is_same<Seg, SegFinalKeepArgs>::value || KEEPM;  // flags + edge records beside the ordinary record write-out
    constexpr bool FLAGS = FINAL || KEEP;
    static_assert(!SEG || (!GEN && !Cfg::DMA), "segmented passes: materialised records");
    static_assert(!FLAGS || Cfg::REUSE || KEEPM, "flag-writing passes: shared staging (the MSD-first last pass also runs with keys and values staged at once)");
    constexpr bool HAS_V = !std::is_same<V, NoVal>::value;
    constexpr bool HAS_W = !std::is_same<W, NoVal>::value;
    using WS = typename std::conditional<HAS_W, W, uint8_t>::type;
    static_assert(!HAS_W || KEEPM || (Cfg::REUSE && HAS_V && !Cfg::DMA), "the auxiliary byte needs the shared-staging configuration");
    constexpr int IPT = Cfg::IPT;
    constexpr bool REUSE = Cfg::REUSE && HAS_V;
    constexpr bool EARLYV = (Cfg::EARLYV || REUSE) && HAS_V;
    constexpr int NT = Cfg::NT;
    constexpr int NW = NT / 64;
    constexpr bool NTM = Cfg::NONTEMP;
    constexpr int TILE = NT * IPT;
    constexpr int WCHUNK = 64 * IPT;  // elements owned by one wave (contiguous => stable)
    using VS = typename std::conditional<HAS_V, V, uint32_t>::type;
    constexpr size_t STAGE_K = sizeof(K) * TILE;
    constexpr size_t STAGE_V = HAS_V ? sizeof(VS) * TILE : 0;
    constexpr size_t STAGE_BYTES = REUSE ? (STAGE_K > STAGE_V ? STAGE_K : STAGE_V) : STAGE_K + STAGE_V;

    // BLK: the generated pass of the big 32-bit-record configuration computes its keys thread-consecutively
    // (rolling, see below); text codes, code table and document table then live in their own LDS region
    // because the staging buffer carries the transposition
    constexpr bool BLK = GEN && sizeof(K) == 4 && sizeof(VS) == 4 && IPT == 16 && (NT == 1024 || NT == 512);
    constexpr uint32_t GEN_TEXTB = ((TILE + RS_GEN_LOOK + 15) / 16) * 16;
    constexpr uint32_t GEN_DOCS = 1024;
    __shared__ __attribute__((aligned(16))) unsigned char s_stage[STAGE_BYTES];
    constexpr uint32_t GEN_SLOTS = 320;  // code -> bucket slot (records mode), behind the document table
    __shared__ __attribute__((aligned(16))) unsigned char s_gen[BLK ? GEN_TEXTB + 512 + GEN_DOCS * 8 + GEN_SLOTS : 16];
    // W32G: a generated pass with 4-byte auxiliary words (bucket records with two or three low digits) has no room for them
    // beside text + records: they cross the tile through the staging buffer in a phase of their own, like keys and values
    constexpr bool W32G = GEN && HAS_W && sizeof(WS) == 4;
    __shared__ WS s_aux[(HAS_W && !W32G && !KEEPM) ? TILE : 1];
    __shared__ uint64_t s_gbase[256];
    __shared__ uint32_t s_whist[NW][256];
    __shared__ uint32_t s_tstart[256];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_tile;
    K* s_keys = reinterpret_cast<K*>(s_stage);
    VS* s_vals = reinterpret_cast<VS*>(s_stage + (REUSE ? 0 : STAGE_K));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (Cfg::GROUP > 0 && !Cfg::TICKET) {
        // static form of the XCD-aware order (experiment): no atomic round trip in front of the tile; relies on workgroups
        // being dispatched in blockIdx order inside an XCD (bounded spins catch a violation)
    } else if constexpr (Cfg::GROUP > 0) {
        if (tid == 0) {
            const uint32_t x = blockIdx.x & 7u;
            const uint32_t slot = atomicAdd(ticket + x, 1u);
            s_tile = ((slot / Cfg::GROUP) * 8u + x) * Cfg::GROUP + slot % Cfg::GROUP;
        }
    } else if constexpr (Cfg::TICKET) {
        if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    }
    // A pass whose look-back timed out leaves its output partly unwritten; a LATER pass over that output would see digit
    // counts its digit starts were not made for and scatter out of bounds (seen as GPU memory faults when two processes
    // shared one device and starved each other's XCD-ordered passes).  Once the error flag is up every tile only publishes
    // an (empty) inclusive prefix — nobody waits for it — and leaves; the host finds the flag and redoes the sort.
    __shared__ uint32_t s_bad;
    if (tid == 0) s_bad = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = tid; i < NW * 256; i += NT) (&s_whist[0][0])[i] = 0;
    __syncthreads();
    uint64_t tile;
    if constexpr (Cfg::GROUP > 0 && !Cfg::TICKET) {
        const uint32_t x = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        tile = (uint64_t)((slot / Cfg::GROUP) * 8u + x) * Cfg::GROUP + slot % Cfg::GROUP;
    } else {
        tile = (Cfg::TICKET || Cfg::GROUP > 0) ? (uint64_t)s_tile : (uint64_t)blockIdx.x;
    }
    uint64_t base = tile * TILE;
    uint64_t tile0 = 0;     // first tile of the look-back chain this tile belongs to
    uint64_t seg_n = n;     // end of the element range the tile may read
    uint32_t sg = 0;
    SegInfo si = {};
    uint64_t ktop = 0;  // MSD-first final pass: the segment's top digit, in place above the 32 key bits
    if constexpr (SEG) {
        if (tile >= (uint64_t)seg.tiles) return;  // (the grid is rounded up to whole tile groups)
        sg = seg.tile_seg ? seg.tile_seg[tile] : 0u;  // (no map: one segment)
        si = seg.segs[sg];
        tile0 = si.tile_begin;
        base = si.begin + (tile - tile0) * TILE;
        seg_n = si.end;
        if constexpr (KEEPM) ktop = (uint64_t)si.top * seg.msd_m;
    } else if constexpr (Cfg::GROUP > 0) {
        if (base >= n) return;  // (the grid is rounded up to whole groups; nobody looks back at a tile behind the input)
    }
    const uint32_t valid = (uint32_t)((seg_n - base) < (uint64_t)TILE ? (seg_n - base) : (uint64_t)TILE);
    if (s_bad) {
        if constexpr (GEN) {
            if (gen.tile_base) return;  // (no status words in the look-back-free form: nobody waits for this tile)
        }
        if (tid < 256) rs_st_status(status + tile * 256 + tid, ((uint64_t)epoch << 56) | (2ull << 54));
        return;
    }

    // ---- load keys (and values), wave-striped: lane-contiguous 512 B per load instruction
    K key[IPT];
    VS val[EARLYV ? IPT : 1];
    WS aux[HAS_W ? IPT : 1] = {};
    constexpr bool GM_PAIR = std::is_same<Gen, TextGenPair>::value;
    constexpr bool GM_REC = std::is_same<Gen, TextGenRec>::value;
    static_assert(!(GM_PAIR || GM_REC) || (BLK && HAS_W), "pair / records generators: 16 Ki tiles with an auxiliary word");
    constexpr bool recs = GM_REC;
    uint32_t gdig[recs ? IPT / 4 : 1] = {};  // records generators: the elements' sort digits, four per register
    auto digit_of = [&](K k, WS a) -> uint32_t {
        if constexpr (HAS_W) {
            if (aux_shift >= 0) return ((uint32_t)a >> aux_shift) & dmask;
        }
        return (uint32_t)(k >> shift) & dmask;
    };
    const uint32_t wbase = wave * WCHUNK + lane;
    if constexpr (GEN) {
        static_assert(!GEN || EARLYV, "generator pass needs early values");
        static_assert(!GEN || STAGE_BYTES >= (size_t)TILE + RS_GEN_LOOK + 2048, "staging buffer too small for the text tile");
        // stage the text of this tile (+ look-ahead) in the still unused LDS staging buffer
        unsigned char* const gbuf = BLK ? s_gen : s_stage;
        uint8_t* s_text = gbuf;
        uint16_t* s_map = reinterpret_cast<uint16_t*>(gbuf + GEN_TEXTB);
        // document boundaries of this tile, copied to LDS when they fit (binary searches then cost
        // LDS instead of L2 latency); s_docs[i] = doc_start[dlo + i]
        constexpr uint32_t DOC_OFF = GEN_TEXTB + 512;
        constexpr uint32_t DOC_CAP = BLK ? GEN_DOCS : (uint32_t)((STAGE_BYTES - DOC_OFF) / 8);
        uint64_t* s_docs = reinterpret_cast<uint64_t*>(gbuf + DOC_OFF);
        // (one workgroup per CU: every dependent global round trip at the head of a tile — ~1-2 us each — is dead time for
        //  the whole CU.  The text of a 16 Ki tile is ONE 16-byte load per thread: it is issued first, beside the tile's
        //  document range, and lands while the document table and the code map are staged)
        constexpr bool PRELOAD = BLK;  // (16 Ki tiles: TILE == 16 * NT; the look-ahead is a second load of the first threads)
        uint4 tw0 = make_uint4(0, 0, 0, 0), tw1 = make_uint4(0, 0, 0, 0);
        bool tw0_ok = false, tw1_ok = false;
        if constexpr (PRELOAD) {
            const uint64_t g0 = base + (uint64_t)tid * 16, g1 = base + (uint64_t)TILE + (uint64_t)tid * 16;
            tw0_ok = gen.padded ? (g0 < n + RS_GEN_LOOK) : (g0 + 16 <= n);
            tw1_ok = (uint32_t)tid * 16 < (uint32_t)RS_GEN_LOOK && (gen.padded ? (g1 < n + RS_GEN_LOOK) : (g1 + 16 <= n));
            if (tw0_ok) tw0 = *reinterpret_cast<const uint4*>(gen.text + g0);
            if (tw1_ok) tw1 = *reinterpret_cast<const uint4*>(gen.text + g1);
        }
        const uint64_t dlo = gen.tile_doc[tile], dhi = gen.tile_doc[tile + 1];
        for (int i = tid; i < 256; i += NT) s_map[i] = gen.symmap[i];
        if constexpr (recs) {
            if (tid < 257) (gbuf + GEN_TEXTB + 512 + GEN_DOCS * 8)[tid] = gen.slotmap[tid];
        }
        const bool docs_in_lds = dhi - dlo + 2 <= (uint64_t)DOC_CAP;
        if (docs_in_lds)
            for (uint32_t i = tid; i < (uint32_t)(dhi - dlo + 2); i += NT) s_docs[i] = gen.doc_start[dlo + i];
        __syncthreads();
        // bytes -> symbol codes on their way into LDS: one table lookup per text byte instead of one per
        // (suffix, symbol)
        for (uint32_t i = tid * 16; i < (uint32_t)TILE + RS_GEN_LOOK; i += NT * 16) {
            if (RS_GEN_ABL & 4) break;
            const uint64_t g = base + i;
            uint32_t x[4];
            if (PRELOAD && (i < (uint32_t)TILE ? tw0_ok : tw1_ok)) {
                const uint4 w = i < (uint32_t)TILE ? tw0 : tw1;
                x[0] = w.x; x[1] = w.y; x[2] = w.z; x[3] = w.w;
            } else if (!PRELOAD && (gen.padded ? (g < n + RS_GEN_LOOK) : (g + 16 <= n))) {
                const uint4 w = *reinterpret_cast<const uint4*>(gen.text + g);
                x[0] = w.x; x[1] = w.y; x[2] = w.z; x[3] = w.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x[q] = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        x[q] |= (uint32_t)((g + 4 * q + b < n) ? gen.text[g + 4 * q + b] : (uint8_t)0) << (8 * b);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                x[q] = (uint32_t)s_map[x[q] & 0xFF] | ((uint32_t)s_map[(x[q] >> 8) & 0xFF] << 8) |
                       ((uint32_t)s_map[(x[q] >> 16) & 0xFF] << 16) | ((uint32_t)s_map[x[q] >> 24] << 24);
            *reinterpret_cast<uint4*>(&s_text[i]) = make_uint4(x[0], x[1], x[2], x[3]);
        }
        __syncthreads();
        // Per-document state lives in registers and changes only when a thread's positions (ascending by 64)
        // leave the document: its end in tile-local 32-bit coordinates (clamped — only "fewer than nsym symbols
        // left" matters) and ebase = doc - (doc_start << bits), so that entry = (position << bits) + ebase.
        uint64_t d = dlo;
        uint32_t dend_l = 0;  // tile-local end of the current document (0 = not looked up yet)
        uint64_t ebase = 0;
        const int nsym = gen.nsym;
        const uint32_t* s_words = reinterpret_cast<const uint32_t*>(s_text);
        if constexpr (GM_PAIR) {
            // Lane-striped generation of the pair form: a key is two 3-symbol parts of the six codes at its position — cheap
            // enough (four multiply-adds) to be evaluated per element straight from the staged codes, in the order the ranking
            // wants them: no transposition of keys, digits and entries through the staging buffer, no barriers.
            // Arithmetic (round 4): the six codes at a position are three PAIRS a = c0 B + c1, m = c2 B + c3, r = c4 B + c5 — one
            // v_dot4_u32_u8 each on the byte-aligned code windows — and with top = floor(a / span), M = span B^4:
            //     key - top M = ((a - top span) B^2 + m) B^2 + r,
            // all products of operands below 2^24 (v_mul_u32_u24 / v_mad_u32_u24, full rate; the 3-symbol parts of round 3
            // needed two 32-bit multiplies and a multiply-high per suffix, quarter rate each).  floor(a / span) = (a R) >> s
            // with (R, s) checked on the host for every a < B^2 (rs_pair_setup).
            const uint32_t B = gen.base, B2 = B * B;
            const uint32_t wlo = B | (1u << 8), whi = (B << 16) | (1u << 24);  // dot4 weights: bytes 0,1 / bytes 2,3
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const uint32_t li = wbase + j * 64;
                key[j] = (K)~(K)0;
                val[j] = VS(0);
                aux[j] = WS(0);
                if (li < valid) {
                    if (li >= dend_l) {  // (also the first element: dend_l = 0)
                        const uint64_t p = base + li;
                        uint64_t ds, de;
                        if (docs_in_lds) {
                            d = dlo + rs_doc_upper(s_docs, d - dlo, dhi - dlo, p);
                            ds = s_docs[d - dlo];
                            de = s_docs[d - dlo + 1];
                        } else {
                            d = rs_doc_upper(gen.doc_start, d, dhi, p);
                            ds = gen.doc_start[d];
                            de = gen.doc_start[d + 1];
                        }
                        const uint64_t rel = de - base;
                        dend_l = rel < (1ull << 30) ? (uint32_t)rel : (1u << 30);
                        ebase = d - (ds << gen.bits);
                    }
                    const uint32_t wi = li >> 2, sel = li & 3u;
                    const uint32_t w0 = s_words[wi], w1 = s_words[wi + 1], w2 = s_words[wi + 2];
                    uint32_t x0 = __builtin_amdgcn_alignbyte(w1, w0, sel);  // codes of li .. li + 3
                    uint32_t x1 = __builtin_amdgcn_alignbyte(w2, w1, sel);  // codes of li + 4 .. li + 7
                    const uint32_t rem = dend_l - li;  // symbols left in the document (>= 1)
                    if (rem < 6u) {  // (rare) the symbols behind the document end count as 0
                        x0 = rem >= 4u ? x0 : (x0 & ((1u << (8u * rem)) - 1u));
                        x1 = rem <= 4u ? 0u : (x1 & 0xFFu);
                    }
                    const uint32_t a = __builtin_amdgcn_udot4(x0, wlo, 0u, false);
                    const uint32_t m = __builtin_amdgcn_udot4(x0, whi, 0u, false);
                    const uint32_t r = __builtin_amdgcn_udot4(x1, wlo, 0u, false);
                    const uint32_t top = __umul24(a, gen.pair_r) >> gen.pair_s;
                    const uint32_t a2 = a - __umul24(top, gen.pair_span);
                    key[j] = (K)(__umul24(__umul24(a2, B2) + m, B2) + r);  // key - top * M (< M <= 2^32)
                    aux[j] = (WS)top;
                    val[j] = (VS)((((uint32_t)base + li) << gen.bits) + (uint32_t)ebase);
                }
            }
        } else if constexpr (BLK) {
            // Thread-consecutive generation: a thread owns IPT consecutive positions, so inside a document
            //   key(q + 1) = (key(q) - code(q) * base^(nsym-1)) * base + code(q + nsym)   [0 behind the document end]
            // and only the first position of a thread or of a document pays for a full Horner evaluation; the
            // document state changes at most a few times per thread.  The records then cross to the
            // lane-consecutive order of the ranking through the (XOR-swizzled, conflict-free) staging buffer.
            uint32_t* kt = reinterpret_cast<uint32_t*>(s_stage);
            auto swz = [](uint32_t q) -> uint32_t { return (q & ~15u) | ((q & 15u) ^ ((q >> 6) & 15u)); };
            const uint32_t q0 = (uint32_t)tid * IPT;
            uint64_t top = 1;  // weight of the symbol that leaves the window
            for (int q = 1; q < nsym; ++q) top *= gen.base;
            uint32_t ent[IPT];
            uint32_t auxc[W32G ? IPT : 1] = {};  // (W32G) the auxiliary words of the thread's consecutive positions
            uint64_t kk = 0;
            if constexpr (GM_REC) {  // bucket records: key = the nsym - 1 symbols BEHIND the first one, 40-bit entries
                const int ns1 = nsym - 1;
                uint64_t top1 = 1;  // weight of the symbol that leaves the (shifted) window
                for (int q = 1; q < ns1; ++q) top1 *= gen.base;
                const uint64_t lmask = (1ull << gen.rec_low_bits) - 1ull;
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const uint32_t q = q0 + j;
                    ent[j] = 0;
                    if (q < valid) {
                        bool fresh = j == 0;
                        if (j == 0 || q >= dend_l) {
                            const uint64_t p = base + q;
                            uint64_t ds, de;
                            if (docs_in_lds) {
                                d = dlo + rs_doc_upper(s_docs, d - dlo, dhi - dlo, p);
                                ds = s_docs[d - dlo];
                                de = s_docs[d - dlo + 1];
                            } else {
                                d = rs_doc_upper(gen.doc_start, d, dhi, p);
                                ds = gen.doc_start[d];
                                de = gen.doc_start[d + 1];
                            }
                            const uint64_t rel = de - base;
                            dend_l = rel < (1ull << 30) ? (uint32_t)rel : (1u << 30);
                            ebase = d - (ds << gen.bits);
                            fresh = true;
                        }
                        if (fresh) {
                            kk = rs_pack_key(s_words, q + 1u, ns1, gen.base, dend_l - q - 1u);
                        } else {
                            const uint64_t cin = q + (uint32_t)nsym - 1u < dend_l ? (uint64_t)s_text[q + nsym - 1] : 0ull;
                            kk = (kk - (uint64_t)s_text[q] * top1) * gen.base + cin;
                        }
                        const uint64_t e64 = ((base + q) << gen.bits) + ebase;
                        kt[swz(q)] = (uint32_t)(kk >> gen.rec_low_bits);
                        if constexpr (W32G) auxc[j] = (uint32_t)((kk & lmask) | ((e64 >> 32) << gen.rec_low_bits));
                        else if constexpr (HAS_W) s_aux[swz(q)] = (WS)((kk & lmask) | ((e64 >> 32) << gen.rec_low_bits));
                        ent[j] = (uint32_t)e64;
                    }
                }
            } else {
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const uint32_t q = q0 + j;
                ent[j] = 0;
                if (q < valid) {
                    bool fresh = j == 0;
                    if (j == 0 || q >= dend_l) {
                        const uint64_t p = base + q;
                        uint64_t ds, de;
                        if (docs_in_lds) {
                            d = dlo + rs_doc_upper(s_docs, d - dlo, dhi - dlo, p);
                            ds = s_docs[d - dlo];
                            de = s_docs[d - dlo + 1];
                        } else {
                            d = rs_doc_upper(gen.doc_start, d, dhi, p);
                            ds = gen.doc_start[d];
                            de = gen.doc_start[d + 1];
                        }
                        const uint64_t rel = de - base;
                        dend_l = rel < (1ull << 30) ? (uint32_t)rel : (1u << 30);
                        ebase = d - (ds << gen.bits);
                        fresh = true;
                    }
                    if (fresh) {
                        kk = rs_pack_key(s_words, q, nsym, gen.base, dend_l - q);
                    } else {
                        const uint64_t cin = q + (uint32_t)nsym <= dend_l ? (uint64_t)s_text[q + nsym - 1] : 0ull;
                        kk = (kk - (uint64_t)s_text[q - 1] * top) * gen.base + cin;
                    }
                    const uint64_t kx = kk;
                    if (gen.msd_shift) {  // (uniform)
                        kt[swz(q)] = (uint32_t)kx;
                        if constexpr (HAS_W && !W32G) s_aux[swz(q)] = (WS)(kx >> gen.msd_shift);
                    } else {
                        kt[swz(q)] = (uint32_t)(kx >> gen.low_bits);
                        if constexpr (HAS_W && !W32G) s_aux[swz(q)] = (WS)(kx & ((1ull << gen.low_bits) - 1ull));
                    }
                    ent[j] = (((uint32_t)base + q) << gen.bits) + (uint32_t)ebase;
                }
            }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const uint32_t li = wbase + j * 64;
                key[j] = li < valid ? (K)kt[swz(li)] : (K)~(K)0;
                if constexpr (HAS_W && !W32G) aux[j] = li < valid ? s_aux[swz(li)] : WS(0);
            }
            __syncthreads();
            if constexpr (W32G) {
#pragma unroll
                for (int j = 0; j < IPT; ++j) kt[swz(q0 + j)] = auxc[j];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const uint32_t li = wbase + j * 64;
                    aux[j] = li < valid ? (WS)kt[swz(li)] : WS(0);
                }
                __syncthreads();
            }
            if (RS_GEN_ABL & 2) {
#pragma unroll
                for (int j = 0; j < IPT; ++j) val[j] = (VS)ent[j];
            } else {
#pragma unroll
            for (int j = 0; j < IPT; ++j) kt[swz(q0 + j)] = ent[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const uint32_t li = wbase + j * 64;
                val[j] = li < valid ? (VS)kt[swz(li)] : VS(0);
            }
            }
            if constexpr (recs) {  // sort digits of the thread's (lane-striped) elements: bucket slot of the first symbol
                const uint8_t* s_slot = gbuf + GEN_TEXTB + 512 + GEN_DOCS * 8;
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const uint32_t li = wbase + j * 64;
                    gdig[j >> 2] |= (uint32_t)s_slot[s_text[li < valid ? li : 0u]] << (8 * (j & 3));
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t li = wbase + j * 64;
            key[j] = (K)~(K)0;
            val[j] = VS(0);
            if constexpr (HAS_W) aux[j] = WS(0);
            if (li < valid) {
                if (li >= dend_l) {  // (also the first element: dend_l = 0)
                    const uint64_t p = base + li;
                    uint64_t ds, de;
                    if (docs_in_lds) {
                        d = dlo + rs_doc_upper(s_docs, d - dlo, dhi - dlo, p);
                        ds = s_docs[d - dlo];
                        de = s_docs[d - dlo + 1];
                    } else {
                        d = rs_doc_upper(gen.doc_start, d, dhi, p);
                        ds = gen.doc_start[d];
                        de = gen.doc_start[d + 1];
                    }
                    const uint64_t rel = de - base;
                    dend_l = rel < (1ull << 30) ? (uint32_t)rel : (1u << 30);
                    ebase = d - (ds << gen.bits);
                }
                uint64_t kk = gen.first_only ? ((uint64_t)s_text[li] << shift)
                                             : rs_pack_key(s_words, li, nsym, gen.base, dend_l - li);
                if constexpr (HAS_W) {
                    if (gen.msd_shift) {  // (uniform)
                        aux[j] = (WS)(kk >> gen.msd_shift);
                        key[j] = (K)kk;
                    } else {
                        aux[j] = (WS)(kk & ((1ull << gen.low_bits) - 1ull));
                        key[j] = (K)(kk >> gen.low_bits);
                    }
                } else {
                    key[j] = (K)kk;
                }
                if constexpr (sizeof(VS) == 4) val[j] = (VS)((((uint32_t)base + li) << gen.bits) + (uint32_t)ebase);
                else val[j] = (VS)(((base + li) << gen.bits) + ebase);
            }
        }
        }
        // the staging buffer is reused for the sorted keys below (the pair generator never touched it: its codes, code table and
        // document table live in s_gen, which nothing writes before the next barrier)
        if constexpr (!GM_PAIR) __syncthreads();
    } else if constexpr (Cfg::DMA && HAS_V && EARLYV && sizeof(K) == 8 && (IPT % 4) == 0) {
        // Keys and values travel global -> LDS by 16-byte-per-lane DMA (global_load_lds: full-rate 1 KiB
        // per wave instruction, no staging registers) into this wave's slice of the still unused
        // staging buffer, and are read back striped.  The value DMA overlaps the ranking below.
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        unsigned char* wdst = s_stage + (size_t)wave * WCHUNK * sizeof(K);
        // clamp to the last 16-byte vector that holds a real element (device blocks are padded to 256 B)
        const uint64_t last_pair = (n - 1) & ~1ull;
#pragma unroll
        for (int i = 0; i < IPT / 2; ++i) {
            uint64_t e = base + (uint64_t)wave * WCHUNK + i * 128 + 2 * lane;
            e = e < last_pair ? e : last_pair;
            __builtin_amdgcn_global_load_lds((gptr_t)(kin + e), (lptr_t)(wdst + i * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t li = wbase + j * 64;
            const K kk = reinterpret_cast<const K*>(wdst)[j * 64 + lane];
            key[j] = li < valid ? kk : (K)~(K)0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slice is free again: reuse it for the values
        constexpr int VPL = 16 / (int)sizeof(VS);             // values per lane per DMA
        const uint64_t last_vec = (n - 1) & ~(uint64_t)(VPL - 1);
#pragma unroll
        for (int i = 0; i < IPT / VPL; ++i) {
            uint64_t e = base + (uint64_t)wave * WCHUNK + i * (64 * VPL) + VPL * lane;
            e = e < last_vec ? e : last_vec;
            __builtin_amdgcn_global_load_lds((gptr_t)(vin + e), (lptr_t)(wdst + i * 1024), 16, 0, 0);
        }
    } else if constexpr (Cfg::LOAD == 0) {
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t li = wbase + j * 64;
            key[j] = li < valid ? rs_load<NTM>(kin + base + li) : (K)~(K)0;
        }
        if constexpr (EARLYV) {
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const uint32_t li = wbase + j * 64;
                val[j] = li < valid ? rs_load<NTM>(vin + base + li) : VS(0);
            }
        }
        if constexpr (HAS_W) {
            if constexpr (!KEEPM) {
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const uint32_t li = wbase + j * 64;
                    aux[j] = li < valid ? rs_load<NTM>(win + base + li) : WS(0);
                }
            }
        }
    } else {
        // Unpredicated loads (slots behind the end of the input re-read its last element and are masked afterwards):
        // straight-line code lets the compiler count outstanding loads exactly, so the ranking starts as soon as what
        // it needs has landed (loads return in issue order).
        const uint32_t lastv = valid - 1;  // (valid >= 1: the grid has ceil(n / TILE) tiles)
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t li = wbase + j * 64;
            key[j] = rs_load<NTM>(kin + base + (li < valid ? li : lastv));
        }
        if constexpr (HAS_W && Cfg::LOAD == 1) {
            if constexpr (!KEEPM) {  // (the final pass of an MSD-first sort has no auxiliary input)
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const uint32_t li = wbase + j * 64;
                    aux[j] = rs_load<NTM>(win + base + (li < valid ? li : lastv));
                }
            }
        }
        if constexpr (EARLYV) {
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const uint32_t li = wbase + j * 64;
                val[j] = rs_load<NTM>(vin + base + (li < valid ? li : lastv));
            }
        }
        if constexpr (HAS_W && Cfg::LOAD != 1) {
            if constexpr (!KEEPM) {
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const uint32_t li = wbase + j * 64;
                    aux[j] = rs_load<NTM>(win + base + (li < valid ? li : lastv));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < IPT; ++j)
            if (wbase + j * 64 >= valid) key[j] = (K)~(K)0;
    }

    // ---- rank inside the wave: lanes with the same digit find each other with 8 ballots, or (ATOMRANK) one
    // returning atomic on the wave's private counter hands every element its rank directly
    uint32_t rank[IPT];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    if constexpr (Cfg::ATOMRANK) {
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t li = wbase + j * 64;
            uint32_t d = digit_of(key[j], aux[HAS_W ? j : 0]);
            if constexpr (recs) d = (gdig[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            d = li < valid ? d : 255u;
            rank[j] = atomicAdd(&s_whist[wave][d], 1u);
        }
    } else {
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const uint32_t li = wbase + j * 64;
        // out-of-range slots take digit 255: they have the largest indices of the tile, so they end
        // up behind every real element and are simply not written out.
        uint32_t d = digit_of(key[j], aux[HAS_W ? j : 0]);
        if constexpr (recs) d = (gdig[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        d = li < valid ? d : 255u;
        uint64_t m = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t below = __popcll(m & lt_mask);
        uint32_t old = 0;
        if (below == 0) {  // lowest lane of the group owns the counter update
            old = s_whist[wave][d];
            s_whist[wave][d] = old + __popcll(m);
        }
        old = __shfl(old, __ffsll((unsigned long long)m) - 1);
        rank[j] = old + below;
    }
    }
    if constexpr (!GEN && Cfg::DMA && HAS_V && EARLYV && sizeof(K) == 8 && (IPT % 4) == 0) {
        // the value DMA issued before the ranking has had the whole loop to land
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const VS* wv = reinterpret_cast<const VS*>(s_stage + (size_t)wave * WCHUNK * sizeof(K));
#pragma unroll
        for (int j = 0; j < IPT; ++j) val[j] = wv[j * 64 + lane];
    }
    __syncthreads();

    // ---- per-digit totals of the tile, exclusive prefix across waves and across digits
    // (threads 0..255 own one digit each; wider workgroups leave the other waves idle here)
    const int d = tid;
    uint32_t cnt = 0, incl = 0;
    uint64_t real = 0;
    const uint64_t tag = (uint64_t)epoch << 56;
    uint64_t* my = status + tile * 256 + (d & 255);
    bool prebased = false;  // (generated passes with counted tile bases: no status words, no look-back)
    uint64_t pre = 0;
    if constexpr (GEN) {
        prebased = gen.tile_base != nullptr;
        if (prebased && tid < 256) pre = (uint64_t)gen.tile_base[tile * 256 + (uint64_t)d];
    }
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint32_t t = s_whist[w][d];
            s_whist[w][d] = cnt;
            cnt += t;
        }
        // real (in-range) count of this digit: padding only ever sits in digit 255
        real = d == 255 ? (uint64_t)cnt - (uint64_t)(TILE - valid) : (uint64_t)cnt;
        // publish the aggregate as early as possible: successors can already add it up
        if (!prebased) rs_st_status(my, tag | ((tile == tile0 ? 2ull : 1ull) << 54) | real);
        // exclusive scan of cnt over the 256 digits: wave scan, then across the 4 digit-owning waves
        incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) s_wsum[wave] = incl;
    }
    __syncthreads();
    uint32_t tstart = 0;
    if (tid < 256) {
        uint32_t wpre = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (w < wave) wpre += s_wsum[w];
        tstart = wpre + incl - cnt;
        s_tstart[d] = tstart;
    }

    __syncthreads();

    // ---- place keys in LDS in sorted-by-digit order
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const uint32_t li = wbase + j * 64;
        uint32_t dd = digit_of(key[j], aux[HAS_W ? j : 0]);
        if constexpr (recs) dd = (gdig[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        dd = li < valid ? dd : 255u;
        const uint32_t pos = s_tstart[dd] + s_whist[wave][dd] + rank[j];
        rank[j] = pos;
        s_keys[pos] = key[j];
        if constexpr (HAS_W && !W32G && !KEEPM) s_aux[pos] = aux[j];
        if constexpr (recs) s_gen[pos] = (unsigned char)dd;  // (the text codes are dead: their LDS carries the digits to the write-out)
    }
    if constexpr (HAS_V && !REUSE) {
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t li = wbase + j * 64;
            if constexpr (EARLYV) {
                s_vals[rank[j]] = val[j];
            } else {
                if (li < valid) s_vals[rank[j]] = rs_load<NTM>(vin + base + li);
            }
        }
    }
    // (the keys now live in LDS: their registers are free during the look-back below)

    // ---- chained scan: look back over the predecessors, then publish the inclusive prefix.
    // A tile becomes ready every pass_time / tiles (~0.1 us at 1 Gi keys) while one agent-scope load
    // costs ~1 us on this part (it has to leave the XCD's L2), so a one-at-a-time walk falls behind and
    // the walk gets ever longer.  LB predecessors are therefore fetched per round trip and consumed
    // nearest-first up to the first inclusive prefix.
    uint64_t excl = 0;
    if (tid < 256 && tile != tile0 && !(Cfg::ABL & 1) && !prebased) {
        constexpr int LB = Cfg::LB;
        int64_t p = (int64_t)tile - 1;
        uint32_t spins = 0;
        bool done = false;
        while (!done) {
            uint64_t sw[LB];
#pragma unroll
            for (int k = 0; k < LB; ++k)
                sw[k] = p - k >= (int64_t)tile0 ? rs_ld_status(status + (uint64_t)(p - k) * 256 + d) : 0ull;
            int used = 0;
            bool open = true;
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                const uint32_t st = (sw[k] >> 56) == (uint64_t)epoch ? ((uint32_t)(sw[k] >> 54) & 3u) : 0u;
                const bool take = open && st != 0;
                excl += take ? (sw[k] & RS_VAL_MASK) : 0ull;
                used += take ? 1 : 0;
                done = done || (take && st == 2);
                open = take && st != 2;
            }
            if ((Cfg::ABL & 8) && d == 0) {  // measurement only: look-back depth / round trips
                atomicAdd(err + 1, (uint32_t)used);
                atomicAdd(err + 2, 1u);
            }
            p -= used;  // the chain's first tile always publishes an inclusive prefix, so p never underflows
            if (used == 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (Cfg::GROUP > 0 ? RS_SPIN_LIMIT_GROUPED : RS_SPIN_LIMIT_PLAIN)) {
                    atomicExch(err, 1u);
                    break;
                }
            } else {
                spins = 0;
            }
        }
        rs_st_status(my, tag | (2ull << 54) | (excl + real));
    }
    uint64_t dstart = 0;
    if (tid < 256) {
        if constexpr (SEG) dstart = (uint64_t)digit_start[(size_t)sg * seg.start_stride + d];
        else dstart = (uint64_t)digit_start[d];
        s_gbase[d] = (Cfg::ABL & 2) ? base : (prebased ? pre : dstart + excl) - (uint64_t)tstart;
    }
    __syncthreads();
    if constexpr (FLAGS) {
        if (tid < 256) {  // edge record of this tile's run of digit d (the sorted keys sit in LDS until the next barrier)
            SegEdge e;
            e.first = e.last = 0;
            e.pos = dstart + excl;
            e.cnt = (uint32_t)real;
            e.pad = 0;
            if (real) {
                const uint64_t lmask = (1ull << seg.low_bits) - 1ull;
                const uint32_t a = tstart, b = tstart + (uint32_t)real - 1u;
                if constexpr (KEEPM) {
                    e.first = ktop + (uint64_t)s_keys[a];
                    e.last = ktop + (uint64_t)s_keys[b];
                } else {
                    e.first = ((uint64_t)s_keys[a] << seg.low_bits) | ((uint64_t)s_aux[HAS_W ? a : 0] & lmask);
                    e.last = ((uint64_t)s_keys[b] << seg.low_bits) | ((uint64_t)s_aux[HAS_W ? b : 0] & lmask);
                }
            }
            seg.edges[tile * 256 + d] = e;
        }
    }

    // ---- coalesced write-out: consecutive lanes -> consecutive slots of one digit run
    if constexpr (!REUSE) {
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t i = j * NT + tid;
            if (i < valid) {
                const K k = s_keys[i];
                const uint32_t dd = (uint32_t)(k >> shift) & dmask;
                const uint64_t dst = s_gbase[dd] + i;
                if constexpr (KEEPM) {
                    // last pass of an MSD-first sort with keys and values staged at once: flags, kept search keys (full key >> 8 and
                    // its low byte) and entries in ONE write-out phase — the logic of the shared-staging branch below
                    bool head = true, tail = true;
                    if (i > 0) head = s_keys[i - 1] != k;
                    if (i + 1 < valid) tail = s_keys[i + 1] != k;
                    const uint64_t kf = ktop + (uint64_t)k;
                    uint32_t f = head ? 1u : 0u;
                    if (!(head && tail)) {
                        const bool exhausted = seg.kmagic ? (kf - __umul64hi(kf, seg.kmagic) * seg.kbase) == 0
                                                          : (kf & (uint64_t)(seg.kbase - 1u)) == 0;
                        if (!exhausted) f |= 2u;
                    }
                    seg.flags[dst] = (uint8_t)f;
                    if ((f & 2u) && seg.tile_sums) {
                        atomicAdd(seg.tile_sums + 2 * (dst / seg.sums_tile), 1ull);
                        if (f & 1u) atomicAdd(seg.tile_sums + 2 * (dst / seg.sums_tile) + 1, 1ull);
                    }
                    rs_store<NTM>(kout + dst, (K)(kf >> 8));
                    if constexpr (HAS_W) rs_store<NTM>(wout + dst, (W)(kf & 0xFFu));
                    rs_store<NTM>(vout + dst, (V)s_vals[i]);
                    continue;
                }
                if (!GEN || kout) rs_store<NTM>(kout + dst, k);  // generated pass: kout may be null (entries only)
                if constexpr (HAS_V) rs_store<NTM>(vout + dst, (V)s_vals[i]);
            }
        }
    } else {
        uint8_t dig[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t i = j * NT + tid;
            dig[j] = 0;
            if (i < valid) {
                const K k = s_keys[i];
                uint32_t dd;
                if constexpr (KEEPM) dd = (uint32_t)(k >> shift) & dmask;  // (no auxiliary word: the digit is a key digit)
                else dd = digit_of(k, s_aux[HAS_W ? i : 0]);
                if constexpr (recs) dd = (uint32_t)s_gen[i];
                dig[j] = (uint8_t)dd;
                if constexpr (FLAGS) {
                    // group flags from the neighbours in the sorted tile (equal keys have equal digits, so the ends of
                    // a digit run come out as head / tail: provisional there, settled by rs_seg_edge_fix_kernel)
                    // (32-bit compares; the "key ends inside the document" test — a 64-bit division by the alphabet size —
                    //  only for the rare element that has an equal neighbour: the pass stays memory-bound)
                    const uint32_t lmask = seg.low_bits >= 32 ? 0xFFFFFFFFu : (1u << seg.low_bits) - 1u;
                    uint32_t ac = 0;
                    if constexpr (!KEEPM) ac = (uint32_t)s_aux[HAS_W ? i : 0] & lmask;
                    bool head = true, tail = true;
                    bool cmp = true;
                    if constexpr (FINAL) cmp = !(RS_SEG_ABL & 4);
                    if constexpr (KEEPM) {
                        if (i > 0) head = s_keys[i - 1] != k;
                        if (i + 1 < valid) tail = s_keys[i + 1] != k;
                    } else if (cmp) {
                        if (i > 0) head = s_keys[i - 1] != k || ((uint32_t)s_aux[HAS_W ? i - 1 : 0] & lmask) != ac;
                        if (i + 1 < valid) tail = s_keys[i + 1] != k || ((uint32_t)s_aux[HAS_W ? i + 1 : 0] & lmask) != ac;
                    }
                    uint32_t f = head ? 1u : 0u;
                    if (!(head && tail)) {
                        const uint64_t kc = KEEPM ? ktop + (uint64_t)k : (((uint64_t)k << seg.low_bits) | (uint64_t)ac);
                        const bool exhausted = seg.kmagic ? (kc - __umul64hi(kc, seg.kmagic) * seg.kbase) == 0
                                                          : (kc & (uint64_t)(seg.kbase - 1u)) == 0;
                        if (!exhausted) f |= 2u;
                    }
                    if constexpr (FINAL) {
                        if (!(RS_SEG_ABL & 1)) seg.flags[rs_seg_rotated(si, s_gbase[dd] + i)] = (uint8_t)f;
                        continue;
                    } else {
                        const uint64_t slot = s_gbase[dd] + i;
                        seg.flags[slot] = (uint8_t)f;
                        if ((f & 2u) && seg.tile_sums) {  // (rare: a fraction of a percent of the suffixes stays unresolved)
                            atomicAdd(seg.tile_sums + 2 * (slot / seg.sums_tile), 1ull);
                            if (f & 1u) atomicAdd(seg.tile_sums + 2 * (slot / seg.sums_tile) + 1, 1ull);
                        }
                    }
                }
                if constexpr (KEEPM) {  // kept search keys in the layout of the LSD split sort: full key >> 8, low byte
                    const uint64_t kf = ktop + (uint64_t)k;
                    rs_store<NTM>(kout + s_gbase[dd] + i, (K)(kf >> 8));
                    if constexpr (HAS_W) rs_store<NTM>(wout + s_gbase[dd] + i, (W)(kf & 0xFFu));
                    continue;
                }
                if (!GEN || kout) rs_store<NTM>(kout + s_gbase[dd] + i, k);
                if constexpr (HAS_W && !W32G) {
                    if (!GEN || wout) rs_store<NTM>(wout + s_gbase[dd] + i, (W)s_aux[i]);
                }
            }
        }
        __syncthreads();
        if constexpr (W32G) {  // the auxiliary words' own phase through the staging buffer
#pragma unroll
            for (int j = 0; j < IPT; ++j) s_vals[rank[j]] = (VS)aux[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const uint32_t i = j * NT + tid;
                if (i < valid) rs_store<NTM>(wout + s_gbase[dig[j]] + i, (W)s_vals[i]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < IPT; ++j) s_vals[rank[j]] = val[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t i = j * NT + tid;
            if constexpr (FINAL) {
                if (i < valid && !(RS_SEG_ABL & 2)) {
                    const unsigned long long slot = rs_seg_rotated(si, s_gbase[dig[j]] + i);
                    const uint32_t hi32 = (uint32_t)s_aux[HAS_W ? i : 0] >> seg.hi_shift;
                    if (seg.elo) {  // (uniform) 5 bytes per entry instead of 8
                        seg.elo[slot] = (uint32_t)s_vals[i];
                        seg.ehi[slot] = (uint8_t)hi32;
                    } else {
                        seg.eout[slot] = ((uint64_t)hi32 << 32) | (uint64_t)s_vals[i];
                    }
                }
            } else {
                if constexpr (GEN && sizeof(VS) == 8) {
                    if (gen.vout_hi) {  // (uniform) entries-only partition of 8-byte entries below 2^40: written packed (Sa40)
                        if (i < valid) {
                            const uint64_t dst = s_gbase[dig[j]] + i;
                            const uint64_t e = (uint64_t)s_vals[i];
                            reinterpret_cast<uint32_t*>(vout)[dst] = (uint32_t)e;
                            gen.vout_hi[dst] = (uint8_t)(e >> 32);
                        }
                        continue;
                    }
                }
                if (i < valid) rs_store<NTM>(vout + s_gbase[dig[j]] + i, (V)s_vals[i]);
            }
        }
    }
}

// Settles the provisional flags at the ends of the per-digit runs of a segmented final pass: for every run (tile t,
// digit d) the run in front of it in the output is the nearest earlier tile of the same segment that holds digit d
// — its last element sits right in front of this run's first.  Equal keys there: the first element is no group head,
// and both elements belong to an unresolved group unless the key ends inside the document.
static __global__ __launch_bounds__(256) void rs_seg_edge_fix_kernel(const SegEdge* __restrict__ edges, const uint32_t* __restrict__ tile_seg,
                                                                     const SegInfo* __restrict__ segs, uint32_t tiles,
                                                                     uint8_t* __restrict__ flags, uint32_t kbase, unsigned long long kmagic,
                                                                     unsigned long long* __restrict__ tile_sums = nullptr,
                                                                     uint32_t sums_tile = 1, const uint32_t* __restrict__ err = nullptr) {
    const uint32_t t = blockIdx.x, d = threadIdx.x;
    if (t >= tiles) return;
    if (err && *err) return;  // (a failed pass left edge records unwritten: their stale slots must not be touched)
    const SegEdge e = edges[(size_t)t * 256 + d];
    if (!e.cnt) return;
    const SegInfo si = segs[tile_seg ? tile_seg[t] : 0u];
    const uint32_t t0 = si.tile_begin;
    if (t == t0) return;
    // largest t' in [t0, t) whose run starts in front of this one (runs of tiles without the digit share its start)
    uint32_t lo = t0, hi = t;  // invariant: answer (if any) in [lo, hi)
    if (edges[(size_t)(t - 1) * 256 + d].pos < e.pos) {
        lo = t - 1;
    } else {
        if (edges[(size_t)t0 * 256 + d].pos >= e.pos) return;  // first run of this digit in the segment
        hi = t - 1;
        while (hi - lo > 1) {  // pos[lo] < e.pos <= pos[hi]
            const uint32_t mid = lo + (hi - lo) / 2;
            if (edges[(size_t)mid * 256 + d].pos < e.pos) lo = mid; else hi = mid;
        }
    }
    const SegEdge pe = edges[(size_t)lo * 256 + d];
    if (pe.last != e.first) return;
    const bool exhausted = kmagic ? (e.first - __umul64hi(e.first, kmagic) * kbase) == 0 : (e.first & (uint64_t)(kbase - 1u)) == 0;
    // one atomic per change; the old value each one returns says exactly what that change did to the tile sums
    auto change = [&](unsigned long long slot, uint32_t clear, uint32_t set) {
        // (the word that holds the byte: `flags` points at the group's first flag, which need not be 4-byte aligned;
        //  the flag array itself is a 256-byte-aligned device block, so the word lies inside it)
        const uintptr_t a = reinterpret_cast<uintptr_t>(flags + slot);
        unsigned int* w = reinterpret_cast<unsigned int*>(a & ~(uintptr_t)3);
        const uint32_t sh = 8u * (uint32_t)(a & 3u);
        uint32_t before, after;
        if (clear) {
            before = (atomicAnd(w, ~(clear << sh)) >> sh) & 0xFFu;
            after = before & ~clear;
        } else {
            before = (atomicOr(w, set << sh) >> sh) & 0xFFu;
            after = before | set;
        }
        if (tile_sums && before != after) {
            const long long da = (long long)((after >> 1) & 1u) - (long long)((before >> 1) & 1u);
            const long long db = (long long)((after >> 1) & after & 1u) - (long long)((before >> 1) & before & 1u);
            unsigned long long* ts = tile_sums + 2 * (slot / sums_tile);
            if (da) atomicAdd(ts, (unsigned long long)da);
            if (db) atomicAdd(ts + 1, (unsigned long long)db);
        }
    };
    const unsigned long long sb = rs_seg_rotated(si, e.pos), sa_ = rs_seg_rotated(si, e.pos - 1);
    change(sb, 1u, 0u);                    // first of this run: not a head
    if (!exhausted) {
        change(sb, 0u, 2u);                // ... and part of an unresolved group, like
        change(sa_, 0u, 2u);               // the last of the run in front: not a tail
    }
}

// ---------------------------------------------------------------------------------------------
// self-test behind the ATOMRANK configurations
// ---------------------------------------------------------------------------------------------
// The one-atomic ranking is only a stable rank if same-address LDS atomics issued by ONE wave instruction
// complete in ascending lane order.  gfx950 does that (the LDS resolves a conflict lowest lane first), but
// it is observed behaviour, not an ISA guarantee — so every process checks it once per device against the
// ballot ranking on conflict patterns from "all 64 lanes on one counter" to "256 counters", and the sorts
// fall back to the ballot configurations if a single return value differs.
static __global__ __launch_bounds__(256) void rs_lane_order_probe_kernel(uint32_t* __restrict__ bad) {
    __shared__ uint32_t cnt[4][256];
    __shared__ uint32_t ref[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * 256; i += 256) {
        (&cnt[0][0])[i] = 0;
        (&ref[0][0])[i] = 0;
    }
    __syncthreads();
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t wrong = 0;
    for (uint32_t r = 0; r < 64; ++r) {
        uint32_t h = (blockIdx.x * 64u + r) * 256u + (uint32_t)tid;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        const uint32_t mode = blockIdx.x & 3u;
        const uint32_t d = mode == 0 ? 7u : mode == 1 ? (h & 1u) * 64u : mode == 2 ? (h & 15u) * 4u : (h & 255u);
        const uint32_t got = atomicAdd(&cnt[wave][d], 1u);
        uint64_t m = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t below = __popcll(m & lt_mask);
        uint32_t old = 0;
        if (below == 0) {
            old = ref[wave][d];
            ref[wave][d] = old + __popcll(m);
        }
        old = __shfl(old, __ffsll((unsigned long long)m) - 1);
        wrong += got != old + below ? 1u : 0u;
    }
    if (wrong) atomicAdd(bad, wrong);
}

// true when the device passed the probe (cached per device; CDB_RANK=ballot forces the ballot ranking)
struct RsRankCache {
    std::mutex mu;
    std::map<int, bool> ok;
    std::map<int, bool> starved;  // a pass in XCD-aware tile order starved on this device: the process keeps plain tickets there
    static RsRankCache& get() {
        static RsRankCache c;
        return c;
    }
};
// the builds' spot check failed with the one-atomic ranking in use: this process ranks with ballots on that device
inline void rs_atomic_rank_disable(int dev) {
    RsRankCache& c = RsRankCache::get();
    std::lock_guard<std::mutex> g(c.mu);
    c.ok[dev] = false;
}
// An XCD-ordered pass starved on `dev` (other kernels — another process sharing the GPU — held the CUs its reserved tiles
// needed): every later build of this process on that device takes plain ticket order from the start instead of paying the
// look-back timeout again per handle (database.cpp builds a fresh index object per rebuild).
inline void rs_group_order_disable(int dev) {
    RsRankCache& c = RsRankCache::get();
    std::lock_guard<std::mutex> g(c.mu);
    c.starved[dev] = true;
}
inline bool rs_group_order_starved(int dev) {
    RsRankCache& c = RsRankCache::get();
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.starved.find(dev);
    return it != c.starved.end() && it->second;
}
inline bool rs_atomic_rank_ok(hipStream_t s) {
    std::mutex& mu = RsRankCache::get().mu;
    std::map<int, bool>& cache = RsRankCache::get().ok;
    int dev = 0;
    CDB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    bool ok = false;
    const char* env = std::getenv("CDB_RANK");
    if (!(env && std::string(env) == "ballot")) {
        DevBuf d_bad;
        d_bad.alloc(sizeof(uint32_t));
        CDB_HIP(hipMemsetAsync(d_bad.p, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(rs_lane_order_probe_kernel, dim3(512), dim3(256), 0, s, d_bad.as<uint32_t>());
        uint32_t bad = 1;
        CDB_HIP(hipMemcpyAsync(&bad, d_bad.p, sizeof(bad), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
        ok = bad == 0;
    }
    cache[dev] = ok;
    return ok;
}

// ---------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------
struct RadixWorkspace {
    DevBuf hist;     // [2][RS_MAX_PASSES][256] u64 : counts, then digit starts
    DevBuf status;   // [tiles][256] u64
    DevBuf tickets;  // [256] u32 tickets (indexed by epoch) + [4] u32 error flag / counters + [256][8] per-class tickets
    DevBuf tile_doc; // [tiles + 1] u64, generated first pass only
    // optional third value buffer for the NEXT sort (cleared by it): with it an odd number of passes still ends
    // with the values in buffer 0 (v0 -> spare -> v1 -> spare ... -> v0) instead of needing a copy back;
    // value_result says where the values ended up (0 = v0, 1 = v1)
    void* value_spare = nullptr;
    int value_result = 0;
    DevBuf seg1;               // one SegInfo: the whole array as a single segment (flags written by the last pass)
    SegInfo h_seg1 = {};
    bool keep_applied = false;  // the last pass of the previous sort wrote the group flags (SegFinalKeepArgs)
    uint32_t epoch = 0;
    uint64_t min_tile = 0;
    // XCD-aware tile order (RS_GROUP) is used only where the caller can redo the sort after a starved pass: the
    // suffix-array build sets allow_group for its own sorts and falls back to plain_order after a look-back timeout
    bool allow_group = false;
    bool plain_order = false;
    bool debug_poison = false;  // test hook: the next sort finds the error flag already up — what the passes BEHIND a starved
                                // pass see: they must leave their (stale) buffers alone, and the host must notice

    void prepare(uint64_t n, int tile, hipStream_t s) {
        const uint64_t tiles = ceil_div(n, (uint64_t)tile);
        if (!hist.p) hist.alloc(2 * RS_MAX_PASSES * 256 * sizeof(uint64_t));
        if (!tickets.p) {
            tickets.alloc((260 + 256 * 8) * sizeof(uint32_t));
            CDB_HIP(hipMemsetAsync(tickets.p, 0, tickets.bytes, s));
        }
        if (debug_poison) {
            CDB_HIP(hipMemsetAsync(err_ptr(), 0xFF, sizeof(uint32_t), s));
            debug_poison = false;
        }
        const size_t need = (size_t)tiles * 256 * sizeof(uint64_t);
        if (need > status.bytes) {
            status.alloc(need + need / 4);
            CDB_HIP(hipMemsetAsync(status.p, 0, status.bytes, s));
            CDB_HIP(hipMemsetAsync(tickets.p, 0, 256 * sizeof(uint32_t), s));
            CDB_HIP(hipMemsetAsync(tickets.as<uint32_t>() + 260, 0, 256 * 8 * sizeof(uint32_t), s));
            epoch = 0;
        }
    }
    uint32_t next_epoch(hipStream_t s) {
        if (epoch == 255) {  // tags wrap: forget every published word
            CDB_HIP(hipMemsetAsync(status.p, 0, status.bytes, s));
            CDB_HIP(hipMemsetAsync(tickets.p, 0, 256 * sizeof(uint32_t), s));
            CDB_HIP(hipMemsetAsync(tickets.as<uint32_t>() + 260, 0, 256 * 8 * sizeof(uint32_t), s));
            epoch = 0;
        }
        return ++epoch;
    }
    uint32_t* ticket_ptr(uint32_t e) { return tickets.as<uint32_t>() + (e & 255u); }
    uint32_t* xticket_ptr(uint32_t e) { return tickets.as<uint32_t>() + 260 + (e & 255u) * 8; }
    uint32_t* err_ptr() { return tickets.as<uint32_t>() + 256; }
    void release() { hist.release(); status.release(); tickets.release(); tile_doc.release(); seg1.release(); epoch = 0; }
};

template <typename K, typename V> inline const char* rs_kernel_name();
template <> inline const char* rs_kernel_name<uint64_t, uint32_t>() { return "rs_onesweep_k64_v32"; }
template <> inline const char* rs_kernel_name<uint64_t, uint64_t>() { return "rs_onesweep_k64_v64"; }
template <> inline const char* rs_kernel_name<uint64_t, NoVal>() { return "rs_onesweep_k64"; }
template <> inline const char* rs_kernel_name<uint32_t, uint32_t>() { return "rs_onesweep_k32_v32"; }
template <> inline const char* rs_kernel_name<uint32_t, uint64_t>() { return "rs_onesweep_k32_v64"; }
template <> inline const char* rs_kernel_name<uint32_t, NoVal>() { return "rs_onesweep_k32"; }

struct SortStats {
    int passes_run = 0, passes_skipped = 0;
};

// kernel configurations (radix_sort's `variant`) that have a generated first pass
inline bool rs_variant_has_gen(int v) {
    return v == 0 || v == 21 || v == 26 || v == 1 || v == 31 || v == 33 || v == 36 || v == 32;
}

struct SortPlan {
    int npass, begin_bit;
    uint32_t last_mask;
    std::vector<uint64_t> h_hist;
};

// One sort.  `h_hist_in` (optional, host, [npass][256]) supplies the per-pass digit histograms when the
// caller can derive them more cheaply than by reading the keys; `gen` (optional) makes the first pass
// produce its (key, value) input on the fly (buffers 0 are then never read).
template <typename K, typename V, typename Cfg, typename Gen = NoGen, typename W = NoVal>
int radix_sort_cfg(hipStream_t s, RadixWorkspace& ws, Profiler& prof, K* k0, K* k1, V* v0, V* v1, uint64_t n,
                   int begin_bit, int end_bit, SortStats* stats, int dbits, const uint64_t* h_hist_in = nullptr,
                   const Gen* gen = nullptr, W* w0 = nullptr, W* w1 = nullptr, int lead_in = 0,
                   const unsigned long long* d_hist_in = nullptr, const SegFinalKeepArgs* keep = nullptr) {
    constexpr int IPT = Cfg::IPT;
    constexpr int TILE = Cfg::NT * IPT;
    constexpr bool HAS_V = !std::is_same<V, NoVal>::value;
    constexpr bool HAS_W = !std::is_same<W, NoVal>::value;
    constexpr bool GEN = !std::is_same<Gen, NoGen>::value;
    // split keys: the first `lead` passes sort on the auxiliary low digits (a generated first pass produces
    // them: lead = low_bits / dbits; materialised records say how many digits their auxiliary array holds:
    // lead_in); the passes over the key bits [begin_bit, end_bit) follow.  h_hist_in has `lead` leading rows.
    if (dbits < 1 || dbits > 8) dbits = 8;
    int lead = 0;
    if constexpr (GEN && HAS_W) lead = gen->low_bits / dbits;
    else if constexpr (HAS_W) lead = lead_in;
    if (HAS_W && !GEN && !h_hist_in && !d_hist_in) throw Error("radix_sort: split records need caller-supplied histograms (internal)");
    const int LEAD = lead;
    if (n == 0 || end_bit < begin_bit || (!LEAD && end_bit == begin_bit)) return 0;
    const int nbits = end_bit - begin_bit;
    const int kpass = (int)ceil_div(nbits, dbits);  // passes over the key proper
    const int npass = kpass + LEAD;
    if (npass > RS_MAX_PASSES) throw Error("radix_sort: too many passes requested");
    const int last_bits = kpass ? nbits - dbits * (kpass - 1) : dbits;
    const uint32_t last_mask = (1u << last_bits) - 1u;
    ws.prepare(n, TILE, s);
    ws.keep_applied = false;

    unsigned long long* d_hist = ws.hist.as<unsigned long long>();
    unsigned long long* d_start = d_hist + RS_MAX_PASSES * 256;
    std::vector<uint64_t> h_hist((size_t)npass * 256);
    if (d_hist_in) {
        // histograms already on the device ([npass][256], lowest digit first): no host round trip at all — the
        // bucket-wise build queues hundreds of sorts back to back (no pass is skipped: the host never sees the counts)
        CDB_HIP(hipMemcpyAsync(d_hist, d_hist_in, (size_t)npass * 256 * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(rs_digit_start_kernel, dim3(npass), dim3(256), 0, s, d_hist, d_start);
    } else if (h_hist_in) {
        std::copy(h_hist_in, h_hist_in + h_hist.size(), h_hist.begin());
        CDB_HIP(hipMemcpyAsync(d_hist, h_hist.data(), h_hist.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(rs_digit_start_kernel, dim3(npass), dim3(256), 0, s, d_hist, d_start);
        CDB_HIP(hipStreamSynchronize(s));  // h_hist is pageable host memory
    } else {
        if (GEN) throw Error("radix_sort: a generated first pass needs caller-supplied histograms (internal)");
        CDB_HIP(hipMemsetAsync(d_hist, 0, RS_MAX_PASSES * 256 * sizeof(uint64_t), s));
        const int grid = (int)std::min<uint64_t>(ceil_div(n, 256 * 16), 256 * 8);
        int t = prof.begin(s);
        hipLaunchKernelGGL(rs_hist_kernel<K>, dim3(grid), dim3(256), 0, s, (const K*)k0, n, begin_bit, npass,
                           dbits, last_mask, d_hist);
        prof.end(t, "rs_hist", n * sizeof(K), s);
        hipLaunchKernelGGL(rs_digit_start_kernel, dim3(npass), dim3(256), 0, s, d_hist, d_start);
        CDB_HIP(hipMemcpyAsync(h_hist.data(), d_hist, h_hist.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        CDB_HIP(hipStreamSynchronize(s));
    }

    K* kb[2] = {k0, k1};
    W* wb[2] = {w0, w1};
    int cur = 0;
    bool materialised = !GEN;  // with a generator the input exists only after the first executed pass
    const uint32_t tiles = (uint32_t)ceil_div(n, (uint64_t)TILE);
    const uint32_t grid_tiles = Cfg::GROUP > 0 ? (uint32_t)(ceil_div(tiles, 8u * Cfg::GROUP) * 8u * Cfg::GROUP) : tiles;
    const size_t pair_bytes = sizeof(K) + (HAS_V ? sizeof(V) : 0) + (HAS_W ? sizeof(W) : 0);
    // which passes run (a constant digit is skipped; the leading pass of a generated sort always runs: it
    // produces the records) and through which value buffers
    std::vector<int> run;
    {
        bool mat = materialised;
        for (int p = 0; p < npass; ++p) {
            bool trivial = false;
            for (int d = 0; d < 256; ++d)
                if (h_hist[(size_t)p * 256 + d] == n) trivial = true;
            if (trivial && (mat || p + 1 < npass) && !(GEN && LEAD > 0 && p == 0)) {
                if (stats) stats->passes_skipped++;
                continue;
            }
            run.push_back(p);
            mat = true;
        }
    }
    V* spare = static_cast<V*>(ws.value_spare);
    ws.value_spare = nullptr;
    const size_t k = run.size();
    std::vector<V*> vseq(k + 1);
    for (size_t i = 0; i <= k; ++i) vseq[i] = (i & 1) ? v1 : v0;
    if (HAS_V && !GEN && spare && (k & 1) && k >= 3) {
        for (size_t i = 1; i < k; ++i) vseq[i] = (i & 1) ? spare : v1;
        vseq[k] = v0;
    }
    ws.value_result = vseq[k] == v0 ? 0 : 1;
    for (size_t ri = 0; ri < k; ++ri) {
        const int p = run[ri];
        V* const vin_p = vseq[ri];
        V* const vout_p = vseq[ri + 1];
        const uint32_t e = ws.next_epoch(s);
        const uint32_t dmask = p == npass - 1 && kpass ? last_mask : ((1u << dbits) - 1u);
        const int shift = begin_bit + dbits * (p - LEAD);  // (unused by the leading passes of a split sort)
        const int aux_shift = p < LEAD ? dbits * p : -1;
        int t = prof.begin(s);
        if (!materialised) {
            if constexpr (GEN) {
                Gen g2 = *gen;
                ws.tile_doc.ensure(((size_t)tiles + 1) * sizeof(uint64_t));
                hipLaunchKernelGGL(rs_tiledoc_kernel, dim3((unsigned)ceil_div((uint64_t)tiles + 1, 256)), dim3(256), 0, s,
                                   g2.doc_start, g2.ndocs, n, (uint64_t)TILE, (uint64_t)tiles, ws.tile_doc.as<uint64_t>());
                g2.tile_doc = ws.tile_doc.as<uint64_t>();
                hipLaunchKernelGGL((rs_onesweep_kernel<K, V, Cfg, Gen, W>), dim3(grid_tiles), dim3(Cfg::NT), 0, s,
                                   (const K*)nullptr, kb[cur ^ 1], (const V*)nullptr, vout_p, n, shift, dmask,
                                   (const unsigned long long*)(d_start + p * 256), ws.status.as<uint64_t>(),
                                   Cfg::GROUP > 0 ? ws.xticket_ptr(e) : ws.ticket_ptr(e), e, ws.err_ptr(), g2, (const W*)nullptr, wb[cur ^ 1], aux_shift);
            }
            prof.end(t, (std::string("rs_onesweep_textgen") + (HAS_W ? "_split" : "") + "_t" + std::to_string(TILE)).c_str(),
                     n * (1 + (kb[cur ^ 1] ? sizeof(K) : 0) + (HAS_V ? sizeof(V) : 0) + (HAS_W ? sizeof(W) : 0)), s);
            materialised = true;
        } else {
            // the last pass can write the group flags beside its records (16 Ki-tile one-atomic configurations, split records)
            constexpr bool CAN_KEEP = HAS_W && HAS_V && sizeof(K) == 4 && sizeof(V) == 4 && Cfg::NT == 1024 && Cfg::ATOMRANK && Cfg::REUSE && !Cfg::DMA;
            bool kept = false;
            if constexpr (CAN_KEEP) {
                if (keep && ri + 1 == k) {
                    ws.h_seg1 = SegInfo{0ull, (unsigned long long)n, 0u, 0u, 0ull, 0ull};
                    ws.seg1.ensure(sizeof(SegInfo));
                    CDB_HIP(hipMemcpyAsync(ws.seg1.p, &ws.h_seg1, sizeof(SegInfo), hipMemcpyHostToDevice, s));
                    SegFinalKeepArgs ka = *keep;
                    ka.tile_seg = nullptr;
                    ka.segs = ws.seg1.as<SegInfo>();
                    ka.tiles = tiles;
                    ka.start_stride = 0;
                    hipLaunchKernelGGL((rs_onesweep_kernel<K, V, Cfg, NoGen, W, SegFinalKeepArgs>), dim3(grid_tiles), dim3(Cfg::NT), 0, s,
                                       (const K*)kb[cur], kb[cur ^ 1], (const V*)vin_p, vout_p, n, shift, dmask,
                                       (const unsigned long long*)(d_start + p * 256), ws.status.as<uint64_t>(),
                                       Cfg::GROUP > 0 ? ws.xticket_ptr(e) : ws.ticket_ptr(e), e, ws.err_ptr(), NoGen(),
                                       (const W*)wb[cur], wb[cur ^ 1], aux_shift, ka);
                    hipLaunchKernelGGL(rs_seg_edge_fix_kernel, dim3(tiles), dim3(256), 0, s, (const SegEdge*)ka.edges, (const uint32_t*)nullptr,
                                       (const SegInfo*)ws.seg1.as<SegInfo>(), tiles, ka.flags, ka.kbase, ka.kmagic, ka.tile_sums, ka.sums_tile, (const uint32_t*)ws.err_ptr());
                    kept = true;
                    ws.keep_applied = true;
                }
            }
            if (!kept)
            hipLaunchKernelGGL((rs_onesweep_kernel<K, V, Cfg, NoGen, W>), dim3(grid_tiles), dim3(Cfg::NT), 0, s,
                               (const K*)kb[cur], kb[cur ^ 1], (const V*)vin_p, vout_p, n, shift, dmask,
                               (const unsigned long long*)(d_start + p * 256), ws.status.as<uint64_t>(),
                               Cfg::GROUP > 0 ? ws.xticket_ptr(e) : ws.ticket_ptr(e), e,
                               ws.err_ptr(), NoGen(), (const W*)wb[cur], wb[cur ^ 1], aux_shift);
            prof.end(t, (std::string(rs_kernel_name<K, V>()) + (HAS_W ? (sizeof(W) == 1 ? "_w8" : (sizeof(W) == 2 ? "_w16" : "_w32")) : "") +
                         (kept ? "_flags" : "") + "_t" + std::to_string(TILE)).c_str(),
                     2 * n * pair_bytes + (kept ? n : 0), s);
        }
        cur ^= 1;
        if (stats) stats->passes_run++;
    }
    CDB_HIP(hipGetLastError());
    return cur;
}

// Sorts n (key, value) pairs by key bits [begin_bit, end_bit), stable.  Buffers 0 hold the input; the
// result ends up in buffers `return value` (0 or 1).  Passes whose digit is constant are skipped.
// `variant` selects a kernel configuration (0 = by size; others exist for A/B measurements).
template <typename K, typename V>
int radix_sort(hipStream_t s, RadixWorkspace& ws, Profiler& prof, K* k0, K* k1, V* v0, V* v1, uint64_t n,
               int begin_bit, int end_bit, SortStats* stats = nullptr, int variant = 0, int dbits = 8,
               const uint64_t* h_hist_in = nullptr, const TextGen* gen = nullptr, const unsigned long long* d_hist_in = nullptr) {
    constexpr bool HAS_V = !std::is_same<V, NoVal>::value;
    const bool atomrank = rs_atomic_rank_ok(s);
    if constexpr (!HAS_V) {
        if (n >= (1ull << 23)) {  // key-only: same 16 Ki-key tile as the pair sort
            if (atomrank && variant != 21 && variant != 33 && ws.allow_group && !ws.plain_order)
                return radix_sort_cfg<K, V, RsCfg<16, false, false, 1024, false, 1, 0, 4, false, true, true, 1, RS_GROUP>>(s, ws, prof, k0, k1, v0, v1, n, begin_bit, end_bit, stats, dbits, h_hist_in);
            if (atomrank && variant != 21)
                return radix_sort_cfg<K, V, RsCfg<16, false, false, 1024, false, 1, 0, 4, false, true>>(s, ws, prof, k0, k1, v0, v1, n, begin_bit, end_bit, stats, dbits, h_hist_in);
            return radix_sort_cfg<K, V, RsCfg<16, false, false, 1024, false, 1, 0, 4>>(s, ws, prof, k0, k1, v0, v1, n, begin_bit, end_bit, stats, dbits, h_hist_in);
        }
        if (atomrank && variant != 21)
            return radix_sort_cfg<K, V, RsCfg<16, false, false, 256, false, 1, 0, 1, false, true>>(s, ws, prof, k0, k1, v0, v1, n, begin_bit, end_bit, stats, dbits, h_hist_in);
        return radix_sort_cfg<K, V, RsCfg<16, false, false>>(s, ws, prof, k0, k1, v0, v1, n, begin_bit, end_bit, stats, dbits, h_hist_in);
    } else {
        // variant 0 picks by size: big tiles (16 Ki keys, one workgroup per CU) give the longest per-digit
        // runs and therefore the best-coalesced scatter, but need >= a few hundred tiles to fill 256 CUs;
        // 31/36/32 = the same tiles with one-atomic ranking (when the device passed rs_atomic_rank_ok)
        if (variant == 0)
            variant = atomrank ? (n >= (1ull << 23) ? 31 : (n >= (1ull << 19) ? 36 : 32))
                               : (n >= (1ull << 23) ? 21 : (n >= (1ull << 19) ? 26 : 1));
        if (variant == 31 && (!ws.allow_group || ws.plain_order)) variant = 33;
        if (!atomrank && variant == 33) variant = 21;
        if (!atomrank && (variant == 31 || variant == 36 || variant == 32)) variant -= variant == 32 ? 31 : 10;
#define CDB_RS(...)                                                                                                      \
    return radix_sort_cfg<K, V, RsCfg<__VA_ARGS__>>(s, ws, prof, k0, k1, v0, v1, n, begin_bit, end_bit, stats, dbits, h_hist_in, \
                                                    (const NoGen*)nullptr, (NoVal*)nullptr, (NoVal*)nullptr, 0, d_hist_in)
#define CDB_RS_GEN(...)                                                                                               \
    if (gen)                                                                                                          \
        return radix_sort_cfg<K, V, RsCfg<__VA_ARGS__>, TextGen>(s, ws, prof, k0, k1, v0, v1, n, begin_bit, end_bit, \
                                                                 stats, dbits, h_hist_in, gen);                       \
    CDB_RS(__VA_ARGS__)
        if (gen && !rs_variant_has_gen(variant))
            throw Error("radix_sort: this kernel configuration has no generated first pass (internal)");
        switch (variant) {
            // production configurations: IPT, REUSE, EARLYV, NT, NONTEMP, MINW, ABL, LB
            default:
            case 21: CDB_RS_GEN(16, true, true, 1024, false, 1, 0, 4);   // 16 Ki-key tile, 1 WG/CU
            case 26: CDB_RS_GEN(18, true, true, 256, false, 1, 0, 4);    // 4.5 Ki-key tile, 3 WG/CU
            case 1: CDB_RS_GEN(15, true, true, 256, false, 1, 0, 1);     // 3.75 Ki-key tile, 4 WG/CU
            // the same three with LDS-atomic ranking
            case 31: CDB_RS_GEN(16, true, true, 1024, false, 1, 0, 4, false, true, true, 1, RS_GROUP);  // + XCD-aware tile order
            case 33: CDB_RS_GEN(16, true, true, 1024, false, 1, 0, 4, false, true);                     // (plain ticket order)
            case 36: CDB_RS_GEN(18, true, true, 256, false, 1, 0, 4, false, true);
            case 32: CDB_RS_GEN(15, true, true, 256, false, 1, 0, 1, false, true);
            // kept for A/B measurements (tools/sort_bench.py): the first version and the LDS-DMA load path; the other
            // design points that were measured (tile shapes, look-back depths, tile orders, non-temporal accesses, timing
            // ablations) are recorded in DESIGN.md §4.1 and can be re-run with tools/experiments/pass_bench.hip
            case 4: CDB_RS(15, false, false, 256, false, 1, 0, 1);        // round-1 first version
            case 41: CDB_RS(16, true, true, 1024, false, 1, 0, 4, true);  // 16 Ki tile, LDS-DMA loads, ballot ranking
        }
#undef CDB_RS
#undef CDB_RS_GEN
    }
}

// throws if any look-back spin hit its bound (the sort result is then garbage)
inline void radix_check_error(hipStream_t s, RadixWorkspace& ws) {
    if (!ws.tickets.p) return;
    uint32_t e = 0;
    CDB_HIP(hipMemcpyAsync(&e, ws.err_ptr(), sizeof(e), hipMemcpyDeviceToHost, s));
    CDB_HIP(hipStreamSynchronize(s));
    if (e) {
        CDB_HIP(hipMemsetAsync(ws.err_ptr(), 0, sizeof(uint32_t), s));
        throw Error("radix sort look-back timed out (internal error)");
    }
    if (getenv("CDB_LOOKBACK_STATS")) {
        uint32_t c[3] = {0, 0, 0};
        CDB_HIP(hipMemcpy(c, ws.err_ptr(), sizeof(c), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[lookback] consumed=%u round_trips=%u\n", c[1], c[2]);
    }
}

// Split sort for keys of up to 32 + low_bits bits: the lowest one or two digits (W = u8 / u16, gen->low_bits =
// digits x dbits) are sorted first — the generated pass takes the lowest — and dropped from the key, so the
// remaining passes move (u32 key >> low_bits, value, W) = 9 or 10 bytes per element instead of 12 for a
// (u64, u32) pair.  h_hist = [all passes][256], lowest digit first.
// gen == nullptr: the records already exist in buffers 0 (their auxiliary array holds sizeof(W) digits).
template <typename V, typename W>
int radix_sort_split(hipStream_t s, RadixWorkspace& ws, Profiler& prof, uint32_t* k0, uint32_t* k1, V* v0, V* v1,
                     W* w0, W* w1, uint64_t n, int hi_bits, SortStats* stats, int variant, int dbits,
                     const uint64_t* h_hist, const TextGen* gen, int key_begin = 0, const unsigned long long* d_hist = nullptr,
                     int lead_digits = -1, const SegFinalKeepArgs* keep = nullptr) {
    // lead_digits: sort digits in the auxiliary array of materialised records (default: every byte of W); the bytes
    // above them are carried along untouched (the bucket-wise build keeps bits 32..39 of its entries there)
    const int lead_in = lead_digits >= 0 ? lead_digits : (int)sizeof(W);
    const bool atomrank = rs_atomic_rank_ok(s);
    // 8-byte values: a 12 Ki-key tile keeps staging + auxiliary bytes inside the 160 KB of LDS
    constexpr int IPT_BIG = sizeof(V) == 8 ? 12 : 16;
    if (variant == 0)
        variant = atomrank ? (n >= (1ull << 23) ? 31 : (n >= (1ull << 19) ? 36 : 32))
                           : (n >= (1ull << 23) ? 21 : (n >= (1ull << 19) ? 26 : 1));
    if (variant == 31 && (!ws.allow_group || ws.plain_order)) variant = 33;
    if (!atomrank && variant == 33) variant = 21;
    if (!atomrank && (variant == 31 || variant == 36 || variant == 32)) variant -= variant == 32 ? 31 : 10;
#define CDB_RS_SPLIT(...)                                                                                              \
    if constexpr (sizeof(W) <= 2) { /* (a generated first pass produces one or two low digits) */                      \
        if (gen)                                                                                                       \
            return radix_sort_cfg<uint32_t, V, RsCfg<__VA_ARGS__>, TextGen, W>(s, ws, prof, k0, k1, v0, v1, n,          \
                                                                               key_begin, hi_bits, stats, dbits,        \
                                                                               h_hist, gen, w0, w1, 0, nullptr, keep);  \
    }                                                                                                                  \
    return radix_sort_cfg<uint32_t, V, RsCfg<__VA_ARGS__>, NoGen, W>(s, ws, prof, k0, k1, v0, v1, n, key_begin, hi_bits, \
                                                                     stats, dbits, h_hist, (const NoGen*)nullptr, w0, w1, \
                                                                     lead_in, d_hist, keep)
    switch (variant) {
        default:
        case 21: CDB_RS_SPLIT(IPT_BIG, true, true, 1024, false, 1, 0, 4);
        case 26: CDB_RS_SPLIT(18, true, true, 256, false, 1, 0, 4);
        case 1: CDB_RS_SPLIT(15, true, true, 256, false, 1, 0, 1);
        case 31: CDB_RS_SPLIT(IPT_BIG, true, true, 1024, false, 1, 0, 4, false, true, true, 1, RS_GROUP);
        case 33: CDB_RS_SPLIT(IPT_BIG, true, true, 1024, false, 1, 0, 4, false, true);
        case 36: CDB_RS_SPLIT(18, true, true, 256, false, 1, 0, 4, false, true);
        case 32: CDB_RS_SPLIT(15, true, true, 256, false, 1, 0, 1, false, true);
    }
#undef CDB_RS_SPLIT
}

// ---------------------------------------------------------------------------------------------
// segmented sort of packed bucket records (the bucket-wise build of corpora >= 2^32)
// ---------------------------------------------------------------------------------------------
constexpr int RS_SEG_TILE = 16384;  // 1024 threads x 16 records (u32 key, u32 value, W)

// first tile of every segment is known on the host; the tile -> segment map is a search per tile
static __global__ __launch_bounds__(256) void rs_seg_tilemap_kernel(const SegInfo* __restrict__ segs, uint32_t nseg, uint32_t tiles,
                                                                    uint32_t* __restrict__ tile_seg) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tiles) return;
    uint32_t lo = 0, hi = nseg - 1;  // largest g with segs[g].tile_begin <= t
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        if (segs[mid].tile_begin <= t) lo = mid; else hi = mid - 1;
    }
    tile_seg[t] = lo;
}

// starts[g][p][d] = first output slot of digit d of pass p inside segment g (hist and starts: [nseg][8][256])
static __global__ __launch_bounds__(256) void rs_seg_digit_start_kernel(const unsigned long long* __restrict__ hist,
                                                                        const SegInfo* __restrict__ segs, int npass,
                                                                        unsigned long long* __restrict__ starts) {
    __shared__ unsigned long long s[256];
    const uint32_t g = blockIdx.x;
    const int t = threadIdx.x;
    const unsigned long long b0 = segs[g].begin;
    for (int p = 0; p < npass; ++p) {
        const unsigned long long c = hist[((size_t)g * 8 + p) * 256 + t];
        s[t] = c;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const unsigned long long v = t >= off ? s[t - off] : 0;
            __syncthreads();
            s[t] += v;
            __syncthreads();
        }
        starts[((size_t)g * 8 + p) * 256 + t] = b0 + s[t] - c;
        __syncthreads();
    }
}

// One stable LSD sort per segment, all segments of a pass in ONE launch.  Records (k32, value, aux) of m elements in
// buffers 0; `lead` low digits sit in the auxiliary word (sorted first), key_bits bits in k32.  The last pass writes
// entries and group flags through `fin` (eout / flags / edges filled in by the caller) instead of records.
// d_hist: digit counts [nseg][8][256], lowest digit first; d_starts: scratch of the same shape.
template <typename W>
void radix_sort_segmented(hipStream_t s, RadixWorkspace& ws, Profiler& prof, uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1,
                          W* w0, W* w1, uint64_t m, const SegInfo* d_segs, const uint32_t* d_tile_seg, uint32_t nseg, uint32_t tiles,
                          const unsigned long long* d_hist, unsigned long long* d_starts, int key_bits, int lead, SegFinalArgs fin,
                          SortStats* stats) {
    if (!rs_atomic_rank_ok(s)) throw Error("radix_sort_segmented: needs the one-atomic ranking (internal)");
    const int kpass = (int)ceil_div((uint64_t)key_bits, 8);
    const int npass = kpass + lead;
    if (npass < 1 || npass > 8 || kpass < 1) throw Error("radix_sort_segmented: unsupported pass count (internal)");
    const int last_bits = key_bits - 8 * (kpass - 1);
    const uint32_t last_mask = (1u << last_bits) - 1u;
    ws.prepare((uint64_t)tiles * RS_SEG_TILE, RS_SEG_TILE, s);
    hipLaunchKernelGGL(rs_seg_digit_start_kernel, dim3(nseg), dim3(256), 0, s, d_hist, d_segs, npass, d_starts);
    SegArgs sa;
    sa.tile_seg = d_tile_seg;
    sa.segs = d_segs;
    sa.tiles = tiles;
    sa.start_stride = 8 * 256;
    static_cast<SegArgs&>(fin) = sa;
    const bool grouped = ws.allow_group && !ws.plain_order;
    using CfgG = RsCfg<16, true, true, 1024, false, 1, 0, 4, false, true, true, 1, RS_GROUP>;
    using CfgP = RsCfg<16, true, true, 1024, false, 1, 0, 4, false, true>;
    const uint32_t grid = grouped ? (uint32_t)(ceil_div(tiles, 8u * RS_GROUP) * 8u * RS_GROUP) : tiles;
    uint32_t* kb[2] = {k0, k1};
    uint32_t* vb[2] = {v0, v1};
    W* wb[2] = {w0, w1};
    int cur = 0;
    const char* wn = sizeof(W) == 1 ? "_w8" : (sizeof(W) == 2 ? "_w16" : "_w32");
    const size_t rec = 8 + sizeof(W);
    for (int p = 0; p < npass; ++p) {
        const uint32_t e = ws.next_epoch(s);
        const uint32_t dmask = p == npass - 1 ? last_mask : 0xFFu;
        const int shift = 8 * (p - lead);
        const int aux_shift = p < lead ? 8 * p : -1;
        const unsigned long long* dstart = d_starts + (size_t)p * 256;
        uint32_t* tk = grouped ? ws.xticket_ptr(e) : ws.ticket_ptr(e);
        int t = prof.begin(s);
#define CDB_SEG_LAUNCH(CFG, SEGT, SEGV)                                                                                       \
    hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CFG, NoGen, W, SEGT>), dim3(grid), dim3(1024), 0, s,            \
                       (const uint32_t*)kb[cur], kb[cur ^ 1], (const uint32_t*)vb[cur], vb[cur ^ 1], m, shift, dmask, dstart, \
                       ws.status.as<uint64_t>(), tk, e, ws.err_ptr(), NoGen(), (const W*)wb[cur], wb[cur ^ 1], aux_shift,     \
                       SEGV)
        if (p == npass - 1) {
            if (grouped) CDB_SEG_LAUNCH(CfgG, SegFinalArgs, fin);
            else CDB_SEG_LAUNCH(CfgP, SegFinalArgs, fin);
            prof.end(t, (std::string("rs_seg_final") + wn + "_t16384").c_str(), m * (rec + (fin.elo ? 6 : 9)), s);
        } else {
            if (grouped) CDB_SEG_LAUNCH(CfgG, SegArgs, sa);
            else CDB_SEG_LAUNCH(CfgP, SegArgs, sa);
            prof.end(t, (std::string("rs_seg_k32_v32") + wn + "_t16384").c_str(), 2 * m * rec, s);
        }
#undef CDB_SEG_LAUNCH
        cur ^= 1;
        if (stats) stats->passes_run++;
    }
    int t = prof.begin(s);
    hipLaunchKernelGGL(rs_seg_edge_fix_kernel, dim3(tiles), dim3(256), 0, s, (const SegEdge*)fin.edges, d_tile_seg, d_segs, tiles,
                       fin.flags, fin.kbase, fin.kmagic, (unsigned long long*)nullptr, 1u, (const uint32_t*)ws.err_ptr());
    prof.end(t, "rs_seg_edge_fix", (uint64_t)tiles * 256 * sizeof(SegEdge), s);
    CDB_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// MSD-first sort of generated split records (suffix-array builds below 2^32 whose key is 33..40 bits wide)
// ---------------------------------------------------------------------------------------------
// The LSD split sort drops the digit its generated pass sorts on into a byte that then travels through every other
// pass (the group flags at the end need the whole key): 9 bytes per record and pass, scattered over the whole array.
// Here the generated pass sorts on the TOP digit instead (key >> 32): an element's bucket then implies that digit, the
// records are (u32 key, u32 entry) = 8 bytes, and the remaining four passes sort every bucket on its own — one
// segmented launch per pass, the scatter of a tile stays inside its bucket (tools/experiments/seg_bench.hip: 3.6 ms per
// pass of 2^30 records against 4.45 ms for the 9-byte records of the LSD form).  The price: the digit histograms of the
// buckets cannot come from the text sweep (bucket x pass x digit counters do not fit the LDS), they are counted from
// the partitioned keys (rs_seg_hist_kernel: 4 B read per record).
constexpr int RS_MSD_HIST_TILES = 32;  // consecutive tiles per workgroup of the histogram sweep

// hist[g][p][d] (the [nseg][8][256] layout of rs_seg_digit_start_kernel) from the materialised records of every segment:
// digit p < lead is byte p of the auxiliary word, digit lead + q byte q of the key (the order the segmented passes sort in)
template <typename W>
__global__ __launch_bounds__(1024) void rs_seg_hist_kernel(const uint32_t* __restrict__ keys, const W* __restrict__ aux, int lead, int npass,
                                                           const uint32_t* __restrict__ tile_seg, const SegInfo* __restrict__ segs,
                                                           uint32_t tiles, unsigned long long* __restrict__ hist) {
    // (round 5) NC copies of the counters, by lane, 64 / NC banks apart: inside a bucket the high digits take few values (the symbols
    // that can follow the bucket's symbol — 64 continuation bytes behind a UTF-8 lead byte), and 64 lanes on one copy collide on the
    // same ADDRESS (SQ_LDS_ADDR_CONFLICT as large as the kernel's LDS issue cycles, profiles/r05e_sq_counters.txt); a copy stride of a
    // multiple of 64 words would leave the four counters of one digit on one bank
    constexpr int NC = 4, HC = 8 * 256 + 64 / NC;  // (8 copies: -3 % at 16 GiB, +3 % at C1 where a workgroup flushes more often)
    __shared__ uint32_t sh4[NC * HC];
    const int tid = threadIdx.x;
    uint32_t* const sh = sh4 + (tid & (NC - 1)) * HC;
    const uint32_t t0 = blockIdx.x * RS_MSD_HIST_TILES;
    const uint32_t t1 = t0 + RS_MSD_HIST_TILES < tiles ? t0 + RS_MSD_HIST_TILES : tiles;
    uint32_t cur = ~0u;
    auto flush = [&]() {
        __syncthreads();
        if (cur != ~0u)
            for (int i = tid; i < npass * 256; i += 1024) {
                uint32_t c = 0;
#pragma unroll
                for (int q = 0; q < NC; ++q) c += sh4[q * HC + i];
                if (c) atomicAdd(&hist[((size_t)cur * 8 + i / 256) * 256 + i % 256], (unsigned long long)c);
            }
        __syncthreads();  // (the sums read all four copies: nobody clears a counter another thread still has to read)
        for (int i = tid; i < NC * HC; i += 1024) sh4[i] = 0;
        __syncthreads();
    };
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t g = tile_seg[t];
        if (g != cur) {  // (uniform)
            flush();
            cur = g;
        }
        const SegInfo si = segs[g];
        const uint64_t base = si.begin + (uint64_t)(t - si.tile_begin) * RS_SEG_TILE;
        const uint32_t valid = (uint32_t)((si.end - base) < (uint64_t)RS_SEG_TILE ? (si.end - base) : (uint64_t)RS_SEG_TILE);
        // 16-byte loads (four records per lane and instruction; round 4: 4-byte loads left the sweep issue-bound at 3.4 TB/s):
        // the tile's records [base, end) = up to three in front of the first 4-aligned index, whole quads, up to three behind
        auto count = [&](uint32_t k, uint32_t a) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                if (p < npass) {
                    const uint32_t dg = p < lead ? (a >> (8 * p)) & 0xFFu : (k >> (8 * (p - lead))) & 0xFFu;
                    atomicAdd(&sh[p * 256 + dg], 1u);
                }
            }
        };
        const uint64_t end = base + valid;
        const uint64_t a0 = (base + 3) & ~3ull;  // (the arrays are 256-byte aligned blocks: index alignment = address alignment)
        const uint64_t q0 = a0 < end ? a0 : end;
        const uint32_t nquad = (uint32_t)((end - q0) / 4);
        if ((uint64_t)tid < q0 - base) count(keys[base + tid], lead > 0 ? (uint32_t)aux[base + tid] : 0u);
        {
            const uint64_t tb = q0 + 4ull * nquad;
            if ((uint64_t)tid < end - tb) count(keys[tb + tid], lead > 0 ? (uint32_t)aux[tb + tid] : 0u);
        }
        constexpr int QPT = RS_SEG_TILE / 4 / 1024;  // quads per thread of a full tile
        uint4 kq[QPT];
        uint32_t aq[QPT][4];
#pragma unroll
        for (int r = 0; r < QPT; ++r) {
            const uint32_t g = (uint32_t)r * 1024u + (uint32_t)tid;
            kq[r] = make_uint4(0, 0, 0, 0);
            aq[r][0] = aq[r][1] = aq[r][2] = aq[r][3] = 0;
            if (g < nquad) {
                kq[r] = *reinterpret_cast<const uint4*>(keys + q0 + 4ull * g);
                if (lead > 0) {
                    if constexpr (sizeof(W) == 4) {
                        const uint4 w = *reinterpret_cast<const uint4*>(aux + q0 + 4ull * g);
                        aq[r][0] = w.x; aq[r][1] = w.y; aq[r][2] = w.z; aq[r][3] = w.w;
                    } else if constexpr (sizeof(W) == 2) {
                        const uint2 w = *reinterpret_cast<const uint2*>(aux + q0 + 4ull * g);
                        aq[r][0] = w.x & 0xFFFFu; aq[r][1] = w.x >> 16; aq[r][2] = w.y & 0xFFFFu; aq[r][3] = w.y >> 16;
                    } else {
                        const uint32_t w = *reinterpret_cast<const uint32_t*>(aux + q0 + 4ull * g);
                        aq[r][0] = w & 0xFFu; aq[r][1] = (w >> 8) & 0xFFu; aq[r][2] = (w >> 16) & 0xFFu; aq[r][3] = w >> 24;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < QPT; ++r) {
            const uint32_t g = (uint32_t)r * 1024u + (uint32_t)tid;
            if (g < nquad) {
                count(kq[r].x, aq[r][0]);
                count(kq[r].y, aq[r][1]);
                count(kq[r].z, aq[r][2]);
                count(kq[r].w, aq[r][3]);
            }
        }
    }
    flush();
}

// ---------------------------------------------------------------------------------------------
// tile bases of the look-back-free generated passes (TextGen::tile_base)
// ---------------------------------------------------------------------------------------------
// counts[rows][256] (u32) = how many elements of tile `row` carry each digit, from a counting sweep over the text (the digit of a
// generated pass is a function of one or two symbols).  base[row][d] = digit_start[d] + sum of counts[r][d] over r < row: the
// output slot of the tile's first element of digit d.  Column d of the result reads column src_col[d] of the counts (nullptr =
// identity; 0xFFFF = no such column: the bucket-wise passes count raw bytes and sort on bucket slots).  Three small kernels:
// column sums per block of RS_TB_ROWS rows, a scan over the blocks, the running sums inside every block.
constexpr int RS_GEN8_TILE = 8192;  // 512 threads x 16 keys: two workgroups per CU
constexpr uint32_t RS_TB_ROWS = 256;
static __global__ __launch_bounds__(256) void rs_tilecol_sum_kernel(const uint32_t* __restrict__ counts, uint32_t rows, const uint16_t* __restrict__ src_col,
                                                                    unsigned long long* __restrict__ partial) {
    const uint32_t d = threadIdx.x, b = blockIdx.x;
    const uint32_t c = src_col ? (uint32_t)src_col[d] : d;
    const uint32_t r0 = b * RS_TB_ROWS, r1 = r0 + RS_TB_ROWS < rows ? r0 + RS_TB_ROWS : rows;
    unsigned long long sum = 0;
    if (c < 256u)
        for (uint32_t r = r0; r < r1; ++r) sum += counts[(size_t)r * 256 + c];
    partial[(size_t)b * 256 + d] = sum;
}
// totals[d] = column sums over all blocks (what the host wants as the digit histogram).  ONE workgroup (the result is 256 numbers),
// but 1024 threads: four quarters of the blocks side by side, eight loads in flight per thread — a 256-thread loop with one load
// per trip took 2.9 ms for the 8192 blocks of a 16 GiB text, all of it latency
static __global__ __launch_bounds__(1024) void rs_tilecol_total_kernel(const unsigned long long* __restrict__ partial, uint32_t blocks,
                                                                       unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long s_q[4][256];
    const uint32_t d = threadIdx.x & 255u, g = threadIdx.x >> 8;
    const uint32_t per = (blocks + 3u) / 4u;
    const uint32_t b0 = g * per, b1 = b0 + per < blocks ? b0 + per : blocks;
    unsigned long long sum = 0;
    uint32_t b = b0;
    for (; b + 8 <= b1; b += 8) {
        unsigned long long v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = partial[(size_t)(b + i) * 256 + d];
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += v[i];
    }
    for (; b < b1; ++b) sum += partial[(size_t)b * 256 + d];
    s_q[g][d] = sum;
    __syncthreads();
    if (g == 0) totals[d] = s_q[0][d] + s_q[1][d] + s_q[2][d] + s_q[3][d];
}
// blockbase[b][d] = digit_start[d] + sum of partial[b'][d], b' < b (same shape: quarter sums first, then every quarter's running sums)
static __global__ __launch_bounds__(1024) void rs_tilecol_scan_kernel(const unsigned long long* __restrict__ partial, uint32_t blocks,
                                                                      const unsigned long long* __restrict__ digit_start,
                                                                      unsigned long long* __restrict__ blockbase) {
    __shared__ unsigned long long s_q[4][256];
    const uint32_t d = threadIdx.x & 255u, g = threadIdx.x >> 8;
    const uint32_t per = (blocks + 3u) / 4u;
    const uint32_t b0 = g * per < blocks ? g * per : blocks, b1 = b0 + per < blocks ? b0 + per : blocks;
    unsigned long long sum = 0;
    uint32_t b = b0;
    for (; b + 8 <= b1; b += 8) {
        unsigned long long v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = partial[(size_t)(b + i) * 256 + d];
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += v[i];
    }
    for (; b < b1; ++b) sum += partial[(size_t)b * 256 + d];
    s_q[g][d] = sum;
    __syncthreads();
    unsigned long long run = digit_start[d];
    for (uint32_t q = 0; q < g; ++q) run += s_q[q][d];
    b = b0;
    for (; b + 8 <= b1; b += 8) {
        unsigned long long v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = partial[(size_t)(b + i) * 256 + d];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            blockbase[(size_t)(b + i) * 256 + d] = run;
            run += v[i];
        }
    }
    for (; b < b1; ++b) {
        blockbase[(size_t)b * 256 + d] = run;
        run += partial[(size_t)b * 256 + d];
    }
}
static __global__ __launch_bounds__(256) void rs_tilecol_apply_kernel(const uint32_t* __restrict__ counts, uint32_t rows, const uint16_t* __restrict__ src_col,
                                                                      const unsigned long long* __restrict__ blockbase,
                                                                      unsigned long long* __restrict__ base) {
    const uint32_t d = threadIdx.x, b = blockIdx.x;
    const uint32_t c = src_col ? (uint32_t)src_col[d] : d;
    const uint32_t r0 = b * RS_TB_ROWS, r1 = r0 + RS_TB_ROWS < rows ? r0 + RS_TB_ROWS : rows;
    unsigned long long run = blockbase[(size_t)b * 256 + d];
    for (uint32_t r = r0; r < r1; ++r) {
        base[(size_t)r * 256 + d] = run;
        if (c < 256u) run += counts[(size_t)r * 256 + c];
    }
}
struct TileBaseWorkspace {
    DevBuf partial, blockbase, totals, base;
    void release() { partial.release(); blockbase.release(); totals.release(); base.release(); }
};
// column totals of the counts (device, [256] u64) — the digit histogram the host builds its buckets from
inline const unsigned long long* rs_tile_totals(hipStream_t s, TileBaseWorkspace& tb, const uint32_t* d_counts, uint32_t rows, const uint16_t* d_src_col) {
    const uint32_t blocks = (uint32_t)ceil_div((uint64_t)rows, (uint64_t)RS_TB_ROWS);
    tb.partial.ensure((size_t)blocks * 256 * sizeof(uint64_t));
    tb.totals.ensure(256 * sizeof(uint64_t));
    hipLaunchKernelGGL(rs_tilecol_sum_kernel, dim3(blocks), dim3(256), 0, s, d_counts, rows, d_src_col, tb.partial.as<unsigned long long>());
    hipLaunchKernelGGL(rs_tilecol_total_kernel, dim3(1), dim3(1024), 0, s, (const unsigned long long*)tb.partial.as<unsigned long long>(), blocks,
                       tb.totals.as<unsigned long long>());
    return tb.totals.as<unsigned long long>();
}
// ... and the tile bases, once the digit starts are on the device (the block sums are taken again: the column map may differ
// from the one the totals were made with — raw bytes there, bucket slots here)
inline const unsigned long long* rs_tile_bases(hipStream_t s, TileBaseWorkspace& tb, const uint32_t* d_counts, uint32_t rows, const uint16_t* d_src_col,
                                               const unsigned long long* d_digit_start) {
    const uint32_t blocks = (uint32_t)ceil_div((uint64_t)rows, (uint64_t)RS_TB_ROWS);
    tb.partial.ensure((size_t)blocks * 256 * sizeof(uint64_t));
    tb.blockbase.ensure((size_t)blocks * 256 * sizeof(uint64_t));
    tb.base.ensure((size_t)rows * 256 * sizeof(uint64_t));
    hipLaunchKernelGGL(rs_tilecol_sum_kernel, dim3(blocks), dim3(256), 0, s, d_counts, rows, d_src_col, tb.partial.as<unsigned long long>());
    hipLaunchKernelGGL(rs_tilecol_scan_kernel, dim3(1), dim3(1024), 0, s, (const unsigned long long*)tb.partial.as<unsigned long long>(), blocks,
                       d_digit_start, tb.blockbase.as<unsigned long long>());
    hipLaunchKernelGGL(rs_tilecol_apply_kernel, dim3(blocks), dim3(256), 0, s, d_counts, rows, d_src_col,
                       (const unsigned long long*)tb.blockbase.as<unsigned long long>(), tb.base.as<unsigned long long>());
    return tb.base.as<unsigned long long>();
}

// (records_sweep.h) the generated pass of radix_sort_msd in two phases: rank on the staged top digits, generate in output order
__global__ __launch_bounds__(512, 8) void rs_sweep_msd_kernel(TextGen gen, uint64_t n, uint32_t tiles, uint32_t* __restrict__ kout,
                                                              uint32_t* __restrict__ vout);

struct MsdWorkspace {
    DevBuf segs, tile_seg, hist, starts;
    void release() { segs.release(); tile_seg.release(); hist.release(); starts.release(); }
};

// Sorts the records the generator produces (key = up to 32 + 8 bits, msd_shift = 32) — see above.  h_top[256] = counts of
// the top digit (host).  Result: entries in v1, kept search keys (u32)(key >> 8) in k1 and key & 0xFF in wout, the group
// flags / edge fixes / tile sums of `keep` written by the last pass.  k0 / v0 are scratch.
inline void radix_sort_msd(hipStream_t s, RadixWorkspace& ws, MsdWorkspace& mw, Profiler& prof, uint32_t* k0, uint32_t* k1,
                           uint32_t* v0, uint32_t* v1, uint8_t* wout, uint64_t n, const uint64_t* h_top, const TextGen& gen_in,
                           unsigned long long msd_m, const SegFinalKeepArgs& keep_in, SortStats* stats,
                           const uint32_t* d_tile_counts = nullptr, TileBaseWorkspace* tbw = nullptr, bool sweep_form = true) {
    // gen_in says how the generated pass splits a key into (top digit, u32 rest): msd_shift = 32 (msd_m = 2^32) or the pair
    // form (msd_m = span * base^4); the last pass puts the full key together again as top * msd_m + rest
    if (!rs_atomic_rank_ok(s)) throw Error("radix_sort_msd: needs the one-atomic ranking (internal)");
    constexpr int TILE = RS_SEG_TILE;
    constexpr int KPASS = 4;
    using CfgG = RsCfg<16, true, true, 1024, false, 1, 0, 4, false, true, true, 1, RS_GROUP>;
    using CfgP = RsCfg<16, true, true, 1024, false, 1, 0, 4, false, true>;
    const bool grouped = ws.allow_group && !ws.plain_order;
    // segments = the non-empty buckets of the top digit
    std::vector<SegInfo> h_segs;
    uint32_t seg_tiles = 0;
    {
        uint64_t at = 0;
        for (int d = 0; d < 256; ++d) {
            if (!h_top[d]) continue;
            h_segs.push_back(SegInfo{(unsigned long long)at, (unsigned long long)(at + h_top[d]), seg_tiles, (uint32_t)d, 0ull, 0ull});
            seg_tiles += (uint32_t)ceil_div(h_top[d], (uint64_t)TILE);
            at += h_top[d];
        }
        if (at != n) throw Error("radix_sort_msd: top-digit histogram does not add up (internal)");
    }
    const uint32_t nseg = (uint32_t)h_segs.size();
    const uint32_t gen_tiles = (uint32_t)ceil_div(n, (uint64_t)TILE);
    ws.prepare((uint64_t)seg_tiles * TILE, TILE, s);
    ws.keep_applied = false;
    unsigned long long* d_hist = ws.hist.as<unsigned long long>();
    unsigned long long* d_start = d_hist + RS_MAX_PASSES * 256;
    mw.segs.ensure(nseg * sizeof(SegInfo));
    mw.tile_seg.ensure((size_t)seg_tiles * sizeof(uint32_t));
    mw.hist.ensure((size_t)nseg * 8 * 256 * sizeof(uint64_t));
    mw.starts.ensure((size_t)nseg * 8 * 256 * sizeof(uint64_t));
    CDB_HIP(hipMemcpyAsync(d_hist, h_top, 256 * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    CDB_HIP(hipMemcpyAsync(mw.segs.p, h_segs.data(), nseg * sizeof(SegInfo), hipMemcpyHostToDevice, s));
    CDB_HIP(hipStreamSynchronize(s));  // (pageable host memory)
    hipLaunchKernelGGL(rs_digit_start_kernel, dim3(1), dim3(256), 0, s, d_hist, d_start);
    hipLaunchKernelGGL(rs_seg_tilemap_kernel, dim3((unsigned)ceil_div(seg_tiles, 256u)), dim3(256), 0, s, mw.segs.as<SegInfo>(), nseg,
                       seg_tiles, mw.tile_seg.as<uint32_t>());
    CDB_HIP(hipMemsetAsync(mw.hist.p, 0, (size_t)nseg * 8 * 256 * sizeof(uint64_t), s));
    // ---- generated pass: partition by the top digit, records (u32 key, entry)
    if (d_tile_counts && tbw && gen_in.msd_pair) {
        // look-back-free form: per-tile counts of the top digit (sa_build.hip: sa_tile_paircount_kernel, rs_tile_totals already ran on
        // them) -> tile bases; 8 Ki-key tiles, two workgroups per CU, no status words, no waiting for other tiles
        constexpr int GT = RS_GEN8_TILE;
        using CfgG8 = RsCfg<16, true, true, 512, false, 1, 0, 4, false, true, true, 1, RS_GROUP>;
        using CfgP8 = RsCfg<16, true, true, 512, false, 1, 0, 4, false, true>;
        const uint32_t gen_tiles8 = (uint32_t)ceil_div(n, (uint64_t)GT);
        TextGenPair gp;
        static_cast<TextGen&>(gp) = gen_in;
        gp.low_bits = 0;
        gp.tile_base = rs_tile_bases(s, *tbw, d_tile_counts, gen_tiles8, nullptr, (const unsigned long long*)d_start);
        ws.tile_doc.ensure(((size_t)gen_tiles8 + 1) * sizeof(uint64_t));
        hipLaunchKernelGGL(rs_tiledoc_kernel, dim3((unsigned)ceil_div((uint64_t)gen_tiles8 + 1, 256)), dim3(256), 0, s, gp.doc_start, gp.ndocs,
                           n, (uint64_t)GT, (uint64_t)gen_tiles8, ws.tile_doc.as<uint64_t>());
        gp.tile_doc = ws.tile_doc.as<uint64_t>();
        const uint32_t e = ws.next_epoch(s);
        const uint32_t grid = grouped ? (uint32_t)(ceil_div(gen_tiles8, 8u * RS_GROUP) * 8u * RS_GROUP) : gen_tiles8;
        int t = prof.begin(s);
        if (sweep_form)  // (no record crosses the LDS, three workgroups per CU; static XCD-aware tile map)
            hipLaunchKernelGGL(rs_sweep_msd_kernel, dim3((uint32_t)(ceil_div(gen_tiles8, 8u * RS_GROUP) * 8u * RS_GROUP)), dim3(512), 0, s,
                               static_cast<const TextGen&>(gp), n, gen_tiles8, k1, v1);
        else if (grouped)
            hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CfgG8, TextGenPair, uint8_t>), dim3(grid), dim3(512), 0, s, (const uint32_t*)nullptr,
                               k1, (const uint32_t*)nullptr, v1, n, 0, 0xFFu, (const unsigned long long*)d_start, ws.status.as<uint64_t>(),
                               ws.xticket_ptr(e), e, ws.err_ptr(), gp, (const uint8_t*)nullptr, (uint8_t*)nullptr, 0);
        else
            hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CfgP8, TextGenPair, uint8_t>), dim3(grid), dim3(512), 0, s, (const uint32_t*)nullptr,
                               k1, (const uint32_t*)nullptr, v1, n, 0, 0xFFu, (const unsigned long long*)d_start, ws.status.as<uint64_t>(),
                               ws.ticket_ptr(e), e, ws.err_ptr(), gp, (const uint8_t*)nullptr, (uint8_t*)nullptr, 0);
        prof.end(t, sweep_form ? "rs_sweep_msd" : "rs_onesweep_textgen_msd_t8192", n * 9, s);
        if (stats) stats->passes_run++;
    } else {
        TextGen g2 = gen_in;
        g2.low_bits = 0;
        ws.tile_doc.ensure(((size_t)gen_tiles + 1) * sizeof(uint64_t));
        hipLaunchKernelGGL(rs_tiledoc_kernel, dim3((unsigned)ceil_div((uint64_t)gen_tiles + 1, 256)), dim3(256), 0, s, g2.doc_start, g2.ndocs,
                           n, (uint64_t)TILE, (uint64_t)gen_tiles, ws.tile_doc.as<uint64_t>());
        g2.tile_doc = ws.tile_doc.as<uint64_t>();
        const uint32_t e = ws.next_epoch(s);
        const uint32_t grid = grouped ? (uint32_t)(ceil_div(gen_tiles, 8u * RS_GROUP) * 8u * RS_GROUP) : gen_tiles;
        int t = prof.begin(s);
#define CDB_MSD_GEN(CFG, GENT, GENV, TICKET)                                                                                              \
    hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CFG, GENT, uint8_t>), dim3(grid), dim3(1024), 0, s, (const uint32_t*)nullptr, \
                       k1, (const uint32_t*)nullptr, v1, n, 0, 0xFFu, (const unsigned long long*)d_start, ws.status.as<uint64_t>(), TICKET, \
                       e, ws.err_ptr(), GENV, (const uint8_t*)nullptr, (uint8_t*)nullptr, 0)
        if (g2.msd_pair) {  // (the generator's form is part of the kernel's type)
            TextGenPair gp;
            static_cast<TextGen&>(gp) = g2;
            if (grouped) CDB_MSD_GEN(CfgG, TextGenPair, gp, ws.xticket_ptr(e));
            else CDB_MSD_GEN(CfgP, TextGenPair, gp, ws.ticket_ptr(e));
        } else {
            if (grouped) CDB_MSD_GEN(CfgG, TextGen, g2, ws.xticket_ptr(e));
            else CDB_MSD_GEN(CfgP, TextGen, g2, ws.ticket_ptr(e));
        }
#undef CDB_MSD_GEN
        prof.end(t, "rs_onesweep_textgen_msd_t16384", n * 9, s);
        if (stats) stats->passes_run++;
    }
    // ---- digit histograms of every bucket's four passes
    {
        int t = prof.begin(s);
        hipLaunchKernelGGL((rs_seg_hist_kernel<uint8_t>), dim3((unsigned)ceil_div(seg_tiles, (uint32_t)RS_MSD_HIST_TILES)), dim3(1024), 0, s,
                           (const uint32_t*)k1, (const uint8_t*)nullptr, 0, KPASS, (const uint32_t*)mw.tile_seg.as<uint32_t>(),
                           (const SegInfo*)mw.segs.as<SegInfo>(), seg_tiles, mw.hist.as<unsigned long long>());
        prof.end(t, "rs_seg_hist", n * 4, s);
        hipLaunchKernelGGL(rs_seg_digit_start_kernel, dim3(nseg), dim3(256), 0, s, (const unsigned long long*)mw.hist.as<unsigned long long>(),
                           (const SegInfo*)mw.segs.as<SegInfo>(), KPASS, mw.starts.as<unsigned long long>());
    }
    // ---- the buckets' LSD passes, one segmented launch each; the last one writes flags + kept keys
    SegArgs sa;
    sa.tile_seg = mw.tile_seg.as<uint32_t>();
    sa.segs = mw.segs.as<SegInfo>();
    sa.tiles = seg_tiles;
    sa.start_stride = 8 * 256;
    SegFinalKeepMsdArgs ka;
    static_cast<SegFinalKeepArgs&>(ka) = keep_in;
    static_cast<SegArgs&>(ka) = sa;
    ka.low_bits = 0;
    ka.msd_m = msd_m;
    const uint32_t grid = grouped ? (uint32_t)(ceil_div(seg_tiles, 8u * RS_GROUP) * 8u * RS_GROUP) : seg_tiles;
    uint32_t* kb[2] = {k0, k1};
    uint32_t* vb[2] = {v0, v1};
    int cur = 1;
    for (int p = 0; p < KPASS; ++p) {
        const uint32_t e = ws.next_epoch(s);
        const unsigned long long* dstart = mw.starts.as<unsigned long long>() + (size_t)p * 256;
        uint32_t* tk = grouped ? ws.xticket_ptr(e) : ws.ticket_ptr(e);
        int t = prof.begin(s);
        // (records of 8 bytes: keys AND values of a 16 Ki tile fit the staging buffer at once — one write-out phase instead
        //  of two, 3.45 instead of 3.62 ms per pass in tools/experiments/seg_bench.hip)
        using CfgG2 = RsCfg<16, false, true, 1024, false, 1, 0, 4, false, true, true, 1, RS_GROUP>;
        using CfgP2 = RsCfg<16, false, true, 1024, false, 1, 0, 4, false, true>;
        if (p + 1 < KPASS) {
#define CDB_MSD_LAUNCH(CFG)                                                                                                          \
    hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CFG, NoGen, NoVal, SegArgs>), dim3(grid), dim3(1024), 0, s,          \
                       (const uint32_t*)kb[cur], kb[cur ^ 1], (const uint32_t*)vb[cur], vb[cur ^ 1], n, 8 * p, 0xFFu, dstart,        \
                       ws.status.as<uint64_t>(), tk, e, ws.err_ptr(), NoGen(), (const NoVal*)nullptr, (NoVal*)nullptr, -1, sa)
            if (grouped) CDB_MSD_LAUNCH(CfgG2);
            else CDB_MSD_LAUNCH(CfgP2);
#undef CDB_MSD_LAUNCH
            prof.end(t, "rs_seg_k32_v32_t16384", 2 * n * 8, s);
        } else {
#define CDB_MSD_FINAL(CFG)                                                                                                           \
    hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CFG, NoGen, uint8_t, SegFinalKeepMsdArgs>), dim3(grid), dim3(1024), 0, s, \
                       (const uint32_t*)kb[cur], kb[cur ^ 1], (const uint32_t*)vb[cur], vb[cur ^ 1], n, 8 * p, 0xFFu, dstart,        \
                       ws.status.as<uint64_t>(), tk, e, ws.err_ptr(), NoGen(), (const uint8_t*)nullptr, wout, -1, ka)
            if (grouped) CDB_MSD_FINAL(CfgG2);
            else CDB_MSD_FINAL(CfgP2);
#undef CDB_MSD_FINAL
            prof.end(t, "rs_seg_k32_v32_flags_t16384", n * (8 + 9 + 1), s);
        }
        cur ^= 1;
        if (stats) stats->passes_run++;
    }
    // (cur == 1: four passes from buffers 1 end in buffers 1)
    {
        int t = prof.begin(s);
        hipLaunchKernelGGL(rs_seg_edge_fix_kernel, dim3(seg_tiles), dim3(256), 0, s, (const SegEdge*)ka.edges, (const uint32_t*)mw.tile_seg.as<uint32_t>(),
                           (const SegInfo*)mw.segs.as<SegInfo>(), seg_tiles, ka.flags, ka.kbase, ka.kmagic, ka.tile_sums, ka.sums_tile, (const uint32_t*)ws.err_ptr());
        prof.end(t, "rs_seg_edge_fix", (uint64_t)seg_tiles * 256 * sizeof(SegEdge), s);
    }
    ws.keep_applied = true;
    CDB_HIP(hipGetLastError());
}

// Bucket-wise build (>= 2^32 suffixes), fused form: ONE generated pass writes the packed bucket records (TextGen::rec_mode)
// of every first-symbol bucket into buffers (k, v, w) — the entries are never partitioned on their own and no gather
// re-reads the text in bucket order — and a sweep over the records counts the digit histograms of every bucket's passes
// (d_hist_out: [nseg][8][256], zeroed here).  h_first[256]: suffixes per bucket slot.  radix_sort_segmented follows.
template <typename W>
void radix_gen_records(hipStream_t s, RadixWorkspace& ws, Profiler& prof, uint32_t* k, uint32_t* v, W* w, uint64_t n, const uint64_t* h_first,
                       const TextGen& gen_in, const uint32_t* d_tile_seg, const SegInfo* d_segs, uint32_t nseg, uint32_t seg_tiles, int lead,
                       int npass, unsigned long long* d_hist_out, SortStats* stats,
                       const uint32_t* d_tile_counts = nullptr, const uint16_t* d_src_col = nullptr, TileBaseWorkspace* tbw = nullptr) {
    if (!rs_atomic_rank_ok(s)) throw Error("radix_gen_records: needs the one-atomic ranking (internal)");
    constexpr int TILE = RS_SEG_TILE;
    using CfgG = RsCfg<16, true, true, 1024, false, 1, 0, 4, false, true, true, 1, RS_GROUP>;
    using CfgP = RsCfg<16, true, true, 1024, false, 1, 0, 4, false, true>;
    const bool grouped = ws.allow_group && !ws.plain_order;
    const uint32_t gen_tiles = (uint32_t)ceil_div(n, (uint64_t)TILE);
    ws.prepare((uint64_t)std::max(gen_tiles, seg_tiles) * TILE, TILE, s);
    unsigned long long* d_hist = ws.hist.as<unsigned long long>();
    unsigned long long* d_start = d_hist + RS_MAX_PASSES * 256;
    CDB_HIP(hipMemcpyAsync(d_hist, h_first, 256 * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    CDB_HIP(hipStreamSynchronize(s));  // (pageable host memory)
    hipLaunchKernelGGL(rs_digit_start_kernel, dim3(1), dim3(256), 0, s, d_hist, d_start);
    CDB_HIP(hipMemsetAsync(d_hist_out, 0, (size_t)nseg * 8 * 256 * sizeof(uint64_t), s));
    TextGenRec g2;
    static_cast<TextGen&>(g2) = gen_in;
    ws.tile_doc.ensure(((size_t)gen_tiles + 1) * sizeof(uint64_t));
    hipLaunchKernelGGL(rs_tiledoc_kernel, dim3((unsigned)ceil_div((uint64_t)gen_tiles + 1, 256)), dim3(256), 0, s, g2.doc_start, g2.ndocs, n,
                       (uint64_t)TILE, (uint64_t)gen_tiles, ws.tile_doc.as<uint64_t>());
    g2.tile_doc = ws.tile_doc.as<uint64_t>();
    const uint32_t e = ws.next_epoch(s);
    const uint32_t grid = grouped ? (uint32_t)(ceil_div(gen_tiles, 8u * RS_GROUP) * 8u * RS_GROUP) : gen_tiles;
    int t = prof.begin(s);
    if (d_tile_counts && tbw) {
        // look-back-free form (TextGen::tile_base): per-tile byte counts (sa_build.hip: sa_tile_bytecount_kernel), columns mapped
        // byte -> bucket slot, scanned over the tiles; 8 Ki-key tiles, two workgroups per CU
        constexpr int GT = RS_GEN8_TILE;
        using CfgG8 = RsCfg<16, true, true, 512, false, 1, 0, 4, false, true, true, 1, RS_GROUP>;
        using CfgP8 = RsCfg<16, true, true, 512, false, 1, 0, 4, false, true>;
        const uint32_t tiles8 = (uint32_t)ceil_div(n, (uint64_t)GT);
        g2.tile_base = rs_tile_bases(s, *tbw, d_tile_counts, tiles8, d_src_col, (const unsigned long long*)d_start);
        ws.tile_doc.ensure(((size_t)tiles8 + 1) * sizeof(uint64_t));
        hipLaunchKernelGGL(rs_tiledoc_kernel, dim3((unsigned)ceil_div((uint64_t)tiles8 + 1, 256)), dim3(256), 0, s, g2.doc_start, g2.ndocs, n,
                           (uint64_t)GT, (uint64_t)tiles8, ws.tile_doc.as<uint64_t>());
        g2.tile_doc = ws.tile_doc.as<uint64_t>();
        const uint32_t grid8 = grouped ? (uint32_t)(ceil_div(tiles8, 8u * RS_GROUP) * 8u * RS_GROUP) : tiles8;
        uint32_t* tk = grouped ? ws.xticket_ptr(e) : ws.ticket_ptr(e);
#define CDB_REC8(CFG, GENT, GENV)                                                                                                      \
    hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CFG, GENT, W>), dim3(grid8), dim3(512), 0, s, (const uint32_t*)nullptr, k, \
                       (const uint32_t*)nullptr, v, n, 0, 0xFFu, (const unsigned long long*)d_start, ws.status.as<uint64_t>(), tk, e,    \
                       ws.err_ptr(), GENV, (const W*)nullptr, w, -1)
        if (grouped) CDB_REC8(CfgG8, TextGenRec, g2);
        else CDB_REC8(CfgP8, TextGenRec, g2);
#undef CDB_REC8
    } else if (grouped)
        hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CfgG, TextGenRec, W>), dim3(grid), dim3(1024), 0, s, (const uint32_t*)nullptr, k,
                           (const uint32_t*)nullptr, v, n, 0, 0xFFu, (const unsigned long long*)d_start, ws.status.as<uint64_t>(),
                           ws.xticket_ptr(e), e, ws.err_ptr(), g2, (const W*)nullptr, w, -1);
    else
        hipLaunchKernelGGL((rs_onesweep_kernel<uint32_t, uint32_t, CfgP, TextGenRec, W>), dim3(grid), dim3(1024), 0, s, (const uint32_t*)nullptr, k,
                           (const uint32_t*)nullptr, v, n, 0, 0xFFu, (const unsigned long long*)d_start, ws.status.as<uint64_t>(),
                           ws.ticket_ptr(e), e, ws.err_ptr(), g2, (const W*)nullptr, w, -1);
    prof.end(t, (std::string("rs_onesweep_textgen_records") + (sizeof(W) == 1 ? "_w8" : (sizeof(W) == 2 ? "_w16" : "_w32")) + "_t16384").c_str(),
             n * (1 + 8 + sizeof(W)), s);
    if (stats) stats->passes_run++;
    t = prof.begin(s);
    hipLaunchKernelGGL((rs_seg_hist_kernel<W>), dim3((unsigned)ceil_div(seg_tiles, (uint32_t)RS_MSD_HIST_TILES)), dim3(1024), 0, s, (const uint32_t*)k,
                       (const W*)w, lead, npass, d_tile_seg, d_segs, seg_tiles, d_hist_out);
    prof.end(t, "rs_seg_hist", n * (4 + (lead > 0 ? sizeof(W) : 0)), s);
    CDB_HIP(hipGetLastError());
}

}  // namespace cdb
