// index_impl.h — state of one GPU string index (the object behind the opaque cdb_index handle).
#pragma once
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "common.h"
#include "radix_sort.h"

namespace cdb {

constexpr int TEXT_PAD = 128;  // readable zero bytes behind library-owned text

struct BuildStats {
    double build_ms = 0, alloc_ms = 0, free_ms = 0;
    int rounds = 0;              // refinement rounds after the initial key sort
    int ext_rounds = 0;          // ... of which text-extension rounds
    int dbl_rounds = 0;          // ... of which prefix-doubling rounds
    uint64_t unresolved_initial = 0;
    uint64_t unresolved_max = 0;
    int sort_passes = 0;         // onesweep launches
    int sort_passes_skipped = 0;
    int isa_built = 0;
    int fused_keygen = 0;
    int key_layout = 0;          // sort records: 0 = (u64 key, entry), 1 = (u32 key, entry), 2 / 3 = (u32 key, entry, u8 / u16 low digits)
    int dense_keys = 0;          // initial keys in base (alphabet + 1) instead of bit-aligned symbols
    int bucketed = 0;            // streamed bucket-wise initial sort (corpora >= 2^32)
    int bucket_low_digits = 0;   // ... low digits (bytes) carried beside the 32-bit bucket key
    int bucket_groups = 0;       // ... bucket groups whose records were gathered in one text-ordered sweep
    int root_folded = 0;         // ... first-symbol buckets laid out in the reference's root order (bytes >= 0x80 first)
    int flags_in_last_pass = 0;  // single sort: group flags written by the last radix pass (no flag kernel)
    int msd_first = 0;           // single sort: top digit first, then (u32, u32) records sorted bucket by bucket (radix_sort_msd)
    int gen_prebased = 0;        // ... generated pass without look-back (counted tile bases)
    int fused_records = 0;       // ... bucket records written by the generated pass itself (one bucket group): no partition + gather
    int sweep_records = 0;       // ... bucket records of every bucket group written by a sweep over the text (records_sweep.h): no partition + gather
    int pairclass_fused = 0;     // reference order, one level below the root: next-byte classes counted beside the bytes (no second sweep over the text)
    int list_rounds = 0;         // refinement rounds compacted from the previous round's list instead of the whole flag array
    int group_sorts = 0, group_sort_fallbacks = 0;  // refinement rounds sorted inside their groups in one pass / sent to the general sort
    int partial_levels = 0;      // ... leftover key bits hold the next symbol quantised to this many levels (sweep form; 0 / 1 = none)
    int vl_key_bits = 0;         // ... keys = the first vl_key_bits - 1 bits of the alphabetic code stream + a "continues" bit (vl_code.h); 0 = dense keys
    double vl_avg_len = 0, vl_rate = 0, vl_est_unresolved = 0, fixed_est_unresolved = 0;
    int segmented = 0;           // ... sorted by segmented passes: one launch per pass for all buckets of a group
    uint64_t gather_items = 0;
    int key_symbols = 0, symbol_bits = 0, alphabet = 0, digit_bits = 8;
    uint64_t final_depth = 0;    // symbols compared when the last group was resolved
    uint64_t compat_rotations = 0, compat_depth = 0;  // reference_compat pass (bytes >= 0x80)
};

struct QueryStats {
    double query_ms = 0;
    double upload_ms = 0, device_ms = 0, download_ms = 0;  // host batches: patterns in, search + rows, results out
    uint64_t nhits = 0, nrows = 0;
};

// kept sort keys for the lone-keyword kernels (query.hip): suffixes starting with the keyword's first min(m, nsym) symbols
// are exactly those with key in [klo, khi] (coded on the host); decisive = the keyword has at most nsym symbols
struct SingleKeys {
    const uint64_t* keys64 = nullptr;
    const uint32_t* keys32 = nullptr;
    const void* keylow = nullptr;
    int low_bits = 0, low_bytes = 0, nsym = 0;
    bool decisive = false;
    uint64_t klo = 0, khi = 0;
    // slots the lower bound can lie in, from the host-side key directory (query.hip: query_keydir_ensure): every slot in
    // front of lo0 holds a key below klo, slot hi0 (if it exists) a key above khi
    bool ranged = false;
    uint64_t lo0 = 0, hi0 = 0;
};

// ---- suffix-array storage as the kernels see it ------------------------------------------------------------------------
// Plain: an array of V (u32 / u64 — the reference's own widths, index.cpp:203-208).  Packed: 8-byte entries whose bits all lie
// below 2^40 (document bits + offset bits <= 40: every >= 2^32 configuration of BASELINE.json) are STORED as u32 low words +
// u8 high bytes — 5 instead of 8 bytes per suffix; cdb_sa_copy / cdb_save expand them to the reference's u64 encoding, so
// nothing outside the library sees the difference.  Kernels are templated on a tag T (uint32_t, uint64_t, Packed40);
// SaOf<T>::ptr is what they index, SaOf<T>::val the entry type they compute with.
struct Sa40 {
    const uint32_t* __restrict__ lo;
    const uint8_t* __restrict__ hi;
    __device__ __forceinline__ uint64_t operator[](uint64_t i) const { return (uint64_t)lo[i] | ((uint64_t)hi[i] << 32); }
};
struct Packed40 {};
template <typename T> struct SaOf {
    using val = T;
    using ptr = const T* __restrict__;
};
template <> struct SaOf<Packed40> {
    using val = uint64_t;
    using ptr = Sa40;
};
// ... and as the build's refinement writes it
template <typename V> struct SaRW {
    V* p;
    using val = V;
    __device__ __forceinline__ V load(uint64_t i) const { return p[i]; }
    __device__ __forceinline__ void store(uint64_t i, V v) const { p[i] = v; }
};
struct Sa40RW {
    uint32_t* lo;
    uint8_t* hi;
    using val = uint64_t;
    __device__ __forceinline__ uint64_t load(uint64_t i) const { return (uint64_t)lo[i] | ((uint64_t)hi[i] << 32); }
    __device__ __forceinline__ void store(uint64_t i, uint64_t v) const {
        lo[i] = (uint32_t)v;
        hi[i] = (uint8_t)(v >> 32);
    }
};

struct Index {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;  // serialises device work of one index (queries from several host threads)
    // coalescing of concurrent single-keyword queries (capi.hip: cdb_query)
    std::mutex qmu;
    std::condition_variable qcv;
    std::vector<void*> qpending;
    bool qleader = false;
    bool coalesce_queries = true;
    bool keep_keys = true;        // keep d_keys when it costs <= 16 GiB (8 bytes per suffix)
    bool use_wave_rows = true;    // wavefront-per-pattern row building when every hit list has <= 64 entries
    bool use_single_query = true; // a lone cdb_query is answered by one wavefront in one launch (query.hip)
    void* h_single = nullptr;     // host-mapped result block of that kernel (+ its device address)
    void* d_single = nullptr;
    // resident_query: one workgroup stays on the device and answers lone keywords from a host-mapped mailbox (query.hip)
    // resident_mode (option resident_query): 0 = never, 1 = always, 2 = automatic (default) — a caller that sends lone keywords back
    // to back (interface.cpp:79-113 loops over a query's keywords; RESIDENT_AUTO_STREAK calls less than RESIDENT_AUTO_GAP_US apart)
    // gets the resident workgroup without asking for it; it idles out by itself ~3 ms after the last keyword.  resident_query is
    // what the call in progress uses (decided in query_single_launch under ix.mu).
    int resident_mode = 2;
    bool resident_query = false;
    uint32_t single_streak = 0;
    int64_t last_single_ns = 0;
    uint64_t res_answers = 0, launched_answers = 0;  // lone keywords answered by the resident workgroup / by a launched kernel (stats)
    void* h_res = nullptr;        // the mailbox (+ its device address)
    void* d_res = nullptr;
    hipStream_t res_stream = nullptr;
    // second stream of the build (sa_build.hip): the pair count of the MSD-first sort runs beside the key-width sample
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_ev[2] = {nullptr, nullptr};
    bool overlap_paircount = true;  // option: 0 = the pair count waits for the sample (one stream)
    bool res_running = false;
    uint32_t res_seq = 0;         // requests posted so far
    SingleKeys res_keys;          // the key arrays the running workgroup was started with
    bool use_fast_search = true;  // pivot-table / galloping search on sorted arrays (query.hip)

    // ---- host staging (cdb_add)
    std::vector<int64_t> ids;
    std::vector<uint64_t> doc_start{0};
    std::string host_text;
    bool host_tables_valid = true;  // false after cdb_build_resident until ids / doc_start are fetched back
    bool host_text_valid = true;    // false while the column lives on the device only (after cdb_build frees its staging
                                    // copy, cdb_load, device builds): cdb_add* fetch it back first (capi.hip)

    // ---- reference-visible parameters (src/index.h:56-57)
    uint64_t bits = 1, mask = 1, size = 0;
    int off_bits = 1;  // bits of the largest document length (the offset field really in use)
    int width = 0;  // 4 / 8, 0 = never built
    uint64_t ndocs = 0;

    // ---- device state
    DevBuf d_text_owned;
    const uint8_t* d_text = nullptr;  // n bytes (+TEXT_PAD when owned)
    bool text_padded = false;
    DevBuf d_doc_start;               // ndocs + 1 u64
    DevBuf d_ids;                     // ndocs i64
    DevBuf d_sa;                      // size * width bytes — or, packed, the entries' low words (size * 4 bytes)
    DevBuf d_sa_hi;                   // packed: bits 32..39 of every entry (size bytes)
    bool sa_packed = false;           // 5-byte storage of 8-byte entries (Sa40 above)
    bool gen_prebased = true;         // option: generated passes take their tile bases from counted per-tile digits (no look-back, two
                                      // 8 Ki-key workgroups per CU; radix_sort.h: TextGen::tile_base)
    TileBaseWorkspace tbw;
    bool key_cost_model = true;       // option: bucket-wise builds weigh one key symbol fewer (a pass saved) against the refinement it costs
    bool pack_sa = true;              // option: builds with 8-byte entries below 2^40 store them packed
    void release_sa() {
        d_sa.release();
        d_sa_hi.release();
        sa_packed = false;
    }
    template <typename T> typename SaOf<T>::ptr sa_view() const;  // (below)
    bool sa_sorted = false;           // SA is globally sorted in unsigned byte order (false only for
                                      // reference_compat orderings of text with bytes >= 0x80)
    DevBuf d_keys;                    // optional: the sorted initial keys (first key_nsym symbol codes of every
                                      // suffix, packed) kept for the search: one load decides most probes
    DevBuf d_symmap_q;                // byte -> symbol code (u16[256]) matching d_keys
    uint16_t h_symmap_q[256] = {};    // ... and its host copy (the lone-keyword path codes its keyword on the host)
    DevBuf d_keys32, d_keylow;        // ... or, after a narrow / split sort, key >> key_low_bits as u32 and (split) the
    int key_low_bits = 0;             // low digit(s) in one or two bytes per suffix: 5-6 instead of 8 bytes per suffix
    int key_low_bytes = 0;
    int key_nsym = 0;
    uint32_t key_base = 0;            // key = the first key_nsym symbol codes as a number in this base
    void drop_keys() {
        d_keys.release();
        d_keys32.release();
        d_keylow.release();
        key_nsym = 0;
        key_low_bits = 0;
        key_low_bytes = 0;
        h_keydir.clear();
        h_keydir.shrink_to_fit();
        keydir_tried = false;
    }
    // Host-side key directory of the lone-keyword path: h_keydir[c] = first slot whose kept key is >= c << keydir_shift
    // (2^keydir_bits + 1 entries), built once per index at its first lone keyword.  The 64-ary search then starts from the
    // few dozen slots between two directory entries instead of the whole array: 1-2 rounds of dependent loads instead of 5-6.
    std::vector<uint32_t> h_keydir;
    int keydir_shift = 0, keydir_bits = 0;
    bool keydir_tried = false;
    bool key_directory = true;  // option: 0 = every lone keyword searches the whole array
    DevBuf d_pivots;                  // top levels of the lower-bound search tree (query.hip), built lazily
    int pivot_levels = 0;

    // ---- scratch kept across calls
    RadixWorkspace rws;
    MsdWorkspace msd_ws;  // segments / per-bucket digit tables of the MSD-first initial sort
    DevBuf scan_partials;
    uint64_t q_spec_cap = 0;  // > 0: hits the next batch's buffers are sized for without asking (query.hip)
    DevBuf q_spec;
    DevBuf q_pat, q_offs, q_left, q_right, q_hoff, q_keys0, q_keys1, q_flags, q_rowptr, q_ids, q_counts, q_hitptr, q_hitoff;

    // ---- options
    bool reference_compat = true;   // bit-parity with the reference also for bytes >= 0x80
    bool force_doubling = false;
    int initial_passes = 0;
    int key_symbols = 0;       // test hook: symbols in the initial sort key (0 = chosen from the text)
    int sort_variant = 0;
    int search_lanes = 0;      // lanes per keyword in the fast batched search: 0 = by batch size, 1 or 8
    bool flags_in_last_pass = true;  // builds below 2^32: the last radix pass writes the group flags (0 = the flag kernel)
    bool fold_depth1 = true;     // ... and (segmented sort) the two byte blocks of big first-symbol buckets swapped by the last pass
    bool fold_root = true;       // bucket-wise build under reference_compat: bucket order = the reference's root child order
    bool segmented_sort = true;  // bucket-wise build: one launch per radix pass for all buckets of a group, entries and flags
                                 // written by the last pass (0 = one sort per bucket + assemble + flag kernels: round 2)
    bool fuse_records = true;  // bucket-wise build with ONE bucket group: the generated pass writes the records (0 = partition + gather)
    bool fuse_pairclass = true; // bucket-wise build of text with bytes >= 0x80: next-byte classes counted by the tile byte count (0 = separate sweep)
    bool list_rounds = true;    // text-extension rounds behind the first compact from the previous round's list (0 = from the flag array)
    int group_sort_cap = 4096;  // ... members of a group on one side of an entry beyond which the pass gives up (its work is also bounded, sa_build.hip)
    bool group_sort = true;     // refinement rounds: one pass inside the groups instead of the general sort (0 = always the general sort)
    bool partial_symbol = true; // bucket-wise build, dense keys in the sweep form: leftover key bits hold the next symbol, quantised (0 = off)
    int vl_keys = 2;           // bucket-wise build, variable-length keys (vl_code.h): 0 off, 2 when the cost model says so, 1 always, 16..56 always with that many key bits
    bool vl_off_once = false;  // (this build only: the sweep form turned out not to apply — dense keys after all)
    bool sweep_records = true; // bucket-wise build with several bucket groups: one sweep over the text per group writes its records (0 = partition + gather)
    bool pack_entries = true;  // bucket-wise build: 8-byte entries below 2^40 travel through the bucket sorts as u32 + u8
    bool msd_first = true;    // keys of 33..40 bits below 2^32 suffixes: top digit first, then every bucket on its own with
                              // (u32, u32) records (radix_sort_msd); 0 = LSD split sort with the low digit as a travelling byte
    bool msd_pair = true;     // ... 6-symbol keys: top digit from the first two symbols (pair count of the text, 32-bit part arithmetic)
    bool keyhist3 = true;     // 24-bit part arithmetic in the key-histogram sweep when the key has 3 P symbols (0 = rolling 64-bit keys)
    int dense_key_retries = 0;  // builds redone with dense keys after the sweep-only key form did not apply (sa_build.hip: RetryWithDenseKeys)
    int group_fallbacks = 0;  // builds redone in plain ticket order after a starved XCD-ordered pass (sa_build.hip)
    uint64_t bucket_group_limit = 0;  // test hook: cap on suffixes per bucket group (0 = what memory allows)
    uint64_t self_check_pairs = 0;    // adjacent pairs the last build's check compared (size - 1 with self_check = 2)
    double self_check_ms = 0;
    // 0 off; 1 a sample of random adjacent pairs behind every build (verify.hip): a failure makes the build fall back to the ballot
    // ranking once, then fail; 2 EVERY adjacent pair before the build returns (a proof, 1.2-1.5 x the build); 3 (default) the sample
    // before the build returns + the proof AFTER it, off the caller's path (verify.hip: proof_start)
    int self_check = 3;
    int self_check_fallbacks = 0;
    bool debug_fail_self_check = false;  // test hook: the first spot check of a build reports a failure
    // ---- proof after publish (self_check = 3).  The reference's array is sorted by construction (std::sort leaves, index.cpp:86-95);
    // here the order rests on an observed LDS lane order (radix_sort.h:12-15) and a sample proves nothing about one stray pair.  So
    // the build publishes as before and a helper thread then compares EVERY adjacent pair against the text on a low-priority stream
    // of its own, slice by slice, beside the queries (it reads arrays that nothing changes until proof_stop).  Damage found: the
    // thread takes ix.mu (queries wait: a wrong array is not served while it is being replaced), switches the device to the ballot
    // ranking, rebuilds with the full check inline and counts self_check_fallbacks.  Every path that replaces or frees the arrays
    // (reset_unbuilt, build_suffix_array, cdb_destroy) calls proof_stop first; it may be called with ix.mu held.
    struct Proof {
        std::thread th;
        std::atomic<int> state{0};       // 0 none, 1 running, 2 proved, 3 damage found and repaired, 4 repair failed, 5 cancelled, 6 could not run
        std::atomic<bool> cancel{false};
        hipStream_t stream = nullptr;    // lowest priority
        void* d_out = nullptr;           // 2 x u64 on the device
        double ms = 0, repair_ms = 0;    // wall time of the sweep / of the repair
        uint64_t pairs = 0;              // adjacent pairs compared
        uint64_t found[2] = {0, 0};      // pairs out of order, invalid entries
        uint64_t skipped = 0;            // pairs left unjudged (0: mixed pairs are judged in place; kept as a stat)
        uint64_t mixed = 0;              // pairs whose first differing bytes lie on both sides of 0x80, judged by the size of their bucket
        uint64_t runs = 0;
        bool of_loaded_file = false;     // the array came from cdb_load: damage says nothing about this device's ranking
        std::atomic<bool> busy{false};   // the helper thread is at work (pre-mapping and / or proof)
        bool want_proof = false;         // (this run of the thread: self_check >= 3)
        double premap_ms = 0;
        uint64_t premap_bytes = 0;       // device memory the helper mapped for the next generation (DevPool::premap)
    } proof;
    // option (default off): behind every build the helper thread maps, into the block cache, the blocks a REBUILD beside this index
    // will ask for and not find (the arrays this index keeps) — database.cpp:276-280 builds the next generation while this one
    // serves.  Off by default: hipMalloc of tens of GB holds a driver lock that kernel launches wait for — the lone keywords of the
    // first half second after a 4 GiB build took 496 instead of 21 ms.  cdb_reserve maps the second generation BEFORE the process
    // serves anything (start-up: server.cpp:43-44), which is where that time belongs.
    bool premap_generation = false;
    bool proof_in_repair = false;        // (build_suffix_array called BY the proof thread: no stop / start of itself)
    uint64_t debug_damage_after_build = 0;  // test hook: swap entries k, k + 1 behind the build's own check (once)
    bool debug_no_segcap = false;   // test hook: the bucket-wise build takes its per-bucket fallback ("a bucket does not fit the record memory")
    int debug_starve_group = 0;     // test hook: a build in XCD-aware tile order reports a look-back timeout once
    bool debug_fail_build = false;  // test hook: the build throws after its sorts (exercises the failure paths)
    bool force_big_path = false;  // test hook: use the >= 2^32 code path (u64 ranks, bucket-wise sort) at any size
    bool fuse_keygen = true;  // first radix pass computes keys from the text (no key/entry materialisation)
    int digit_bits = 0;
    bool narrow_keys = true;      // 32-bit sort keys (+ a byte for the dropped low digit) when the key width allows
    int key_coding = 0;           // initial sort keys: 0 = dense when that saves a pass, 1 = bit-aligned symbols, 2 = dense
    uint64_t query_hit_budget = 1ull << 31;  // hits resolved per chunk of a batch (16 B of scratch each)

    double host_upload_ms = 0, host_free_ms = 0;  // cdb_build: staged column to the device / staging copy released
    Profiler prof;
    BuildStats bstats;
    QueryStats qstats;
    mutable std::mutex err_mu;  // guards err (concurrent failing queries)
    std::string err;
};

// capi.hip — ids / doc_start on the host (fetched from the device after a resident build)
void ensure_host_tables(Index& ix);
void ensure_host_staging(Index& ix);

// capi.hip — pieces shared with shards.hip
void read_raw_dir(const char* dir, const char* key, std::vector<int64_t>& ids, std::vector<uint64_t>& doc_start, std::string& text,
                  uint64_t& nrec, uint64_t& nadd);

// sa_build.hip
void build_suffix_array(Index& ix);

// verify.hip — out = {inversions, tie-order violations, wrapped sum of entries, invalid entries, expected sum}
void verify_suffix_array(Index& ix, uint64_t out[5]);
void sa_expand(Index& ix, uint64_t first, uint64_t cnt, uint64_t* d_out);  // verify.hip: packed entries -> u64 (on ix.stream)
void sa_pack_inplace(Index& ix);                                           // verify.hip: u64 entries in d_sa -> packed storage
void sa_pack_chunk(hipStream_t s, const uint64_t* d_in, uint64_t cnt, uint32_t* lo, uint8_t* hi, uint64_t first);  // verify.hip
inline bool sa_packable(const Index& ix) { return ix.pack_sa && ix.width == 8 && (int)ix.bits + ix.off_bits <= 40; }
void spot_check_suffix_array(Index& ix, uint32_t samples, uint64_t out[2]);  // verify.hip: the check behind every build
void proof_start(Index& ix);   // verify.hip: the full check behind a published build, on its own thread and stream (ix.mu held)
void proof_stop(Index& ix);    // ... cancelled and joined (before the arrays change; callable with ix.mu held)
void proof_forget(Index& ix);  // ... and the handle forgotten (cdb_destroy)
// sizes (>= 16 MiB) of the device blocks a published index holds on to: its arrays and the work spaces a handle keeps across calls —
// what a second generation built beside it will ask the block cache for once more (cdb_reserve, DevPool::premap)
inline std::vector<size_t> retained_block_sizes(const Index& ix) {
    std::vector<size_t> out;
    for (const DevBuf* b : {&ix.d_text_owned, &ix.d_sa, &ix.d_sa_hi, &ix.d_keys, &ix.d_keys32, &ix.d_keylow, &ix.d_doc_start, &ix.d_ids,
                            &ix.rws.status, &ix.rws.tile_doc, &ix.msd_ws.segs, &ix.msd_ws.tile_seg, &ix.msd_ws.hist, &ix.msd_ws.starts,
                            &ix.scan_partials, &ix.tbw.partial, &ix.tbw.blockbase, &ix.tbw.totals, &ix.tbw.base})
        if (b->p && b->bytes >= (16u << 20)) out.push_back(b->bytes);
    return out;
}
void debug_swap_entries(Index& ix, uint64_t k);  // verify.hip (test hook): entries k and k + 1 of the finished array swapped
// the REFERENCE's order (signed child order inside radix nodes, unsigned below), checked pair by pair:
// out = {pairs out of reference order, pairs whose next bytes differ in sign class, of those inside radix nodes,
// equal suffixes not ascending by document}
void verify_reference_order(Index& ix, uint64_t out[4]);
// number of entries of a suffix array (device pointers) that do not name a real (document, offset)
uint64_t count_invalid_entries(hipStream_t s, const void* d_sa, int width, uint64_t n, const uint64_t* d_doc_start, uint64_t ndocs,
                               int bits, uint64_t mask);

// query.hip — patterns already on the device; leaves CSR results in ix.q_rowptr / q_ids / q_counts
struct DeviceCsr {
    uint64_t npat = 0, nrows = 0, nhits = 0;
};
// with_offsets: additionally ix.q_hitptr[nrows+1] / ix.q_hitoff[nhits] = byte offsets of every occurrence,
// grouped by result row, ascending
DeviceCsr query_batch_on_device(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat,
                                bool with_offsets = false);
// one keyword through the single-wavefront kernel; false = not applicable, use the batched path
bool query_single_on_device(Index& ix, const char* kw, size_t len, int64_t** ids_out, int64_t** counts_out, size_t* nrows);
// ... in two halves (several indexes queried at once, shards.hip): launch, then collect (false = handed over to the
// batched path); Absent = a byte the text never holds, the answer is empty without a launch (query_single_empty)
enum class SingleLaunch { NotApplicable, Absent, Launched };
SingleLaunch query_single_launch(Index& ix, const char* kw, size_t len);
bool query_single_collect(Index& ix, int64_t** ids_out, int64_t** counts_out, size_t* nrows);
void query_single_empty(Index& ix, int64_t** ids_out, int64_t** counts_out, size_t* nrows);
void query_resident_stop(Index& ix);  // the resident workgroup leaves (before the arrays it reads are replaced or freed)
// highlight spans of all documents matching any pattern: ids -> ix.q_ids, span_ptr -> ix.q_rowptr, span begins ->
// ix.q_keys0, inclusive span ends -> ix.q_keys1
struct SpanResult {
    uint64_t ndocs = 0, nspans = 0, nhits = 0;
};
SpanResult query_spans_on_device(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat,
                                 uint64_t total_pattern_bytes);
// union over the patterns by object id with summed counts, rows ascending by id, in ix.q_ids / q_counts
DeviceCsr query_or_on_device(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat);
// ... filtered to lo <= count < hi and ranked: descending count, ties ascending id, at most `limit` rows (0 = all)
DeviceCsr query_ranked_on_device(Index& ix, const uint8_t* d_blob, const uint64_t* d_offs, uint64_t npat, int64_t lo, int64_t hi,
                                 uint64_t limit);

// AND across keys (interface.cpp:114-134): intersection of row lists (each ascending by id, ids unique within a list)
// by object id with summed counts, on ix's stream; rows land in ix.q_ids / q_counts — ascending id, or filtered to
// lo <= count < hi and ranked (descending count, ties ascending id, at most `limit` rows) when `ranked`
struct DeviceRows {
    const int64_t* d_ids;
    const int64_t* d_counts;
    uint64_t n;
};
DeviceCsr and_merge_on_device(Index& ix, const std::vector<DeviceRows>& lists, bool ranked, int64_t lo, int64_t hi, uint64_t limit);


template <> inline SaOf<uint32_t>::ptr Index::sa_view<uint32_t>() const { return d_sa.as<uint32_t>(); }
template <> inline SaOf<uint64_t>::ptr Index::sa_view<uint64_t>() const { return d_sa.as<uint64_t>(); }
template <> inline SaOf<Packed40>::ptr Index::sa_view<Packed40>() const { return Sa40{d_sa.as<uint32_t>(), d_sa_hi.as<uint8_t>()}; }
// f(tag) with the tag of the index's storage: the one place that knows the three forms
template <typename F> inline auto sa_dispatch(const Index& ix, F&& f) {
    if (ix.sa_packed) return f(Packed40{});
    if (ix.width == 8) return f(uint64_t{});
    return f(uint32_t{});
}

}  // namespace cdb

// the object behind the C ABI's opaque handle (capi.hip, shards.hip)
struct cdb_index {
    cdb::Index ix;
};
struct cdb_key_query;
namespace cdb {
int query_and_with_lead(cdb_index* lead, const cdb_key_query* keys, int nkeys, int ranked, int64_t corr_lo, int64_t corr_hi,
                        uint64_t limit, int64_t** ids, int64_t** counts, size_t* nrows);
}
