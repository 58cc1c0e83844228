/* coffeedb_gpu.h — C ABI of libcoffeedb_gpu.so, the MI355X (gfx950) implementation of CoffeeDB's
 * string-index hot path (suffix-array build + substring-match scan).
 *
 * Every entry point states the reference interface it replaces (paths relative to the CoffeeDB tree,
 * /root/reference in the build container).  The reference-side binding is a ~60-line C++ shim
 * (coffeedb_amd/csrc/index.h + index.cpp, shown in INTEGRATION.md) that keeps the reference's
 * `index` / `string_index` classes so src/database.cpp compiles and behaves unchanged.
 *
 * Conventions: plain C types, no exceptions across the boundary.  Functions returning `int` return
 * CDB_OK (0) or a CDB_E_* code; the message for the last failure on a handle is cdb_last_error(h)
 * and uses the reference's own wording where the reference throws (index.cpp:196,199,240).
 * A handle may be used from several host threads at once for queries (reference contract:
 * query() is const and runs under a shared lock, database.cpp:388); build is exclusive.
 * Device pointers handed to the *_device entry points must hold complete data when the call is made:
 * the library works on its own non-blocking HIP stream and does not wait for the caller's streams
 * (callers synchronise first, e.g. torch.cuda.synchronize()); every entry point returns only after its
 * own device work has finished.
 */
#ifndef COFFEEDB_GPU_H
#define COFFEEDB_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDB_OK 0
#define CDB_E_INVALID 1   /* bad argument / empty keyword / capacity limit (reference: std::runtime_error) */
#define CDB_E_DEVICE 2    /* HIP runtime failure, no usable gfx950 device, out of device memory */
#define CDB_E_INTERNAL 3  /* internal invariant violated (e.g. bounded look-back spin expired) */

typedef struct cdb_index cdb_index; /* opaque; replaces `string_index` (src/index.h:54-86) */

/* Result of a batched query in CSR form: pattern j owns rows [row_ptr[j], row_ptr[j+1]); each row is
 * (ids[r], counts[r]) = (object id, number of overlapping occurrences = $correlation), rows of one
 * pattern ascending by document insertion index — the order string_index::query returns
 * (src/index.cpp:316-322).  Host memory, owned by the library until cdb_result_free(). */
typedef struct cdb_result {
    uint64_t npat;
    uint64_t nrows;
    uint64_t nhits;      /* total suffix-array entries matched over the batch */
    uint64_t* row_ptr;   /* npat + 1 */
    int64_t* ids;        /* nrows */
    int64_t* counts;     /* nrows */
} cdb_result;

/* ---- lifetime -------------------------------------------------------------------------------- */

/* replaces std::make_unique<string_index>() (src/database.cpp:255, :303).  device < 0 selects the
 * current HIP device.  Fails with CDB_E_DEVICE when no gfx950 GPU is usable: there is no CPU path. */
int cdb_create(cdb_index** out, int device);

/* replaces string_index::~string_index (src/index.h:80-84) */
void cdb_destroy(cdb_index* h);

const char* cdb_last_error(const cdb_index* h);

/* ---- ingest ---------------------------------------------------------------------------------- */

/* replaces string_index::add(int64_t id, std::string_view value) (src/index.cpp:174-177).
 * The bytes are copied into a host staging buffer (the reference keeps a non-owning view instead).  cdb_build
 * frees that staging copy once the column is on the device; adding to an index that was built, loaded
 * (cdb_load) or built from device memory first fetches the column back, so "load, add, rebuild" works. */
int cdb_add(cdb_index* h, int64_t id, const char* value, size_t len);

/* Bulk form of cdb_add for callers that already hold a concatenated column:
 * document d = blob[doc_start[d] .. doc_start[d+1]), ndocs documents. */
int cdb_add_bulk(cdb_index* h, const int64_t* ids, const char* blob, const uint64_t* doc_start, uint64_t ndocs);

/* Raw-record ingest — the step before add() in database.cpp:170-275: CoffeeDB keeps one
 * binary file per object (writer database.cpp:334-378):
 *   int64 id | int32 nfields | nfields x { int32 keylen | key | int8 type | value },
 *   value = 1 byte (type 0 bool) | 8 bytes (1 integer, 2 double) | int32 len + bytes (3 string).
 * cdb_raw_record_find_string locates the string value stored under `key` in one in-memory record
 * (returns 1 found, 0 absent / not a string, -1 malformed; no handle or GPU needed);
 * cdb_add_raw_record adds that value under the record's id (a record without the key is skipped). */
int cdb_raw_record_find_string(const void* record, size_t len, const char* key, int64_t* id, const char** value,
                               size_t* value_len);
int cdb_add_raw_record(cdb_index* h, const char* key, const void* record, size_t len);
/* Bulk form: every record file of directory `dir` (CoffeeDB's storage_location/raw/, one file per object) in ascending
 * file-name order; the string under `key` of each record goes straight into the staged column.  *records = files
 * parsed, *added = documents added.  All or nothing: a malformed or unreadable file fails the call and leaves the
 * staged column as it was. */
int cdb_add_raw_dir(cdb_index* h, const char* dir, const char* key, uint64_t* records, uint64_t* added);

/* Persistence of a built index (the reference has none and rebuilds every index at start,
 * server.cpp:44): header + ids + doc_start + text + suffix array.  cdb_load restores a queryable index
 * without rebuilding. */
int cdb_save(cdb_index* h, const char* path);
int cdb_load(cdb_index* h, const char* path);

/* ---- build ----------------------------------------------------------------------------------- */

/* replaces string_index::build() (src/index.cpp:178-236): computes bits/mask/size and the entry width
 * exactly as the reference does (index.cpp:182-208), uploads the staged text and constructs the
 * suffix array on the GPU.  Errors reuse the reference's messages (index.cpp:196,199). */
int cdb_build(cdb_index* h);

/* Build straight from the caller's host column, without the staging copy cdb_add_bulk makes — string_index itself
 * holds non-owning string_views into the caller's strings (src/index.h:58, database.cpp:262-264).  ids[ndocs],
 * doc_start[ndocs + 1] (offsets into blob, non-decreasing; blob[doc_start[0] .. doc_start[ndocs]) is the column).
 * Replaces whatever cdb_add* staged.  The text travels through a chunked pinned upload (~50 GB/s); afterwards the
 * handle holds device copies only (a later cdb_add fetches the column back first). */
int cdb_build_view(cdb_index* h, const int64_t* ids, const char* blob, const uint64_t* doc_start, uint64_t ndocs);
/* The same for documents that are separate strings: document d = lens[d] bytes at ptrs[d] — what string_index::add
 * collects (index.cpp:174-177) and what the shim's build() passes.  The strings must stay valid during the call only. */
int cdb_build_views(cdb_index* h, const int64_t* ids, const char* const* ptrs, const uint64_t* lens, uint64_t ndocs);

/* Same build, but over text that already resides in device memory (HBM-resident timing in bench.py,
 * multi-GPU shards).  d_text must stay valid for the lifetime of the index (the reference's
 * string_view contract, database.cpp:262-264); doc_start/ids are host arrays and are copied. */
int cdb_build_device(cdb_index* h, const void* d_text, const uint64_t* doc_start, const int64_t* ids,
                     uint64_t ndocs);

/* The same with the document table resident as well: d_doc_start (ndocs + 1 offsets into d_text, d_doc_start[0]
 * = 0) and d_ids are DEVICE arrays; they are validated and copied on the device, nothing but a few scalars
 * crosses PCIe.  (Host copies of the tables — and of the text, for cdb_add* followed by a rebuild — are fetched
 * back from the device when cdb_save / cdb_add* need them later; d_text must still be valid then.) */
int cdb_build_resident(cdb_index* h, const void* d_text, const uint64_t* d_doc_start, const int64_t* d_ids,
                       uint64_t ndocs);

/* ---- query ----------------------------------------------------------------------------------- */

/* replaces string_index::query(const std::string& keyword) (src/index.cpp:237-326).  On success
 * *ids / *counts point to *nrows entries allocated by the library (release with cdb_free).
 * Empty keyword -> CDB_E_INVALID, message "Empty keywords are not allowed" (index.cpp:239-241).
 * A never-built index returns zero rows (the reference reads uninitialised state there). */
int cdb_query(cdb_index* h, const char* keyword, size_t len, int64_t** ids, int64_t** counts, size_t* nrows);
void cdb_free(void* p);

/* Batched form (no reference counterpart — interface.cpp:79-113 loops query() per keyword): pattern j
 * = blob[offsets[j] .. offsets[j+1]).  Any empty pattern fails the whole call like cdb_query. */
int cdb_query_batch(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out);
void cdb_result_free(cdb_result* r);

/* Batched query that also emits WHERE every occurrence sits (BASELINE config 2: "highlight offset
 * emission"): row r of the CSR result owns occurrences [hit_ptr[r], hit_ptr[r+1]) of `offsets`, the byte
 * offsets of the keyword inside that row's document, ascending.  The offsets are the `entry >> bits`
 * fields of the matched suffix-array range, sorted along with (pattern, document). */
typedef struct cdb_hits {
    uint64_t* hit_ptr;  /* nrows + 1 */
    uint64_t* offsets;  /* nhits */
} cdb_hits;
int cdb_query_batch_offsets(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out,
                            cdb_hits* hits);
void cdb_hits_free(cdb_hits* x);

/* OR over the keywords of ONE string key — replaces the per-key merge loop of interface.cpp:78-113
 * (query each keyword, sort by id, merge lists by id summing the counts): returns the union of the
 * matching objects with their summed $correlation, rows ascending by object id — exactly the list
 * filter() holds for that key before the AND across keys, the $correlation range filter and the final
 * ranking (which stay host-side in the shim: coffeedb_amd/csrc/shim/ranking.h).  Object ids are assumed
 * unique (they are insertion timestamps, interface.cpp:151,178).  Release ids/counts with cdb_free. */
int cdb_query_or(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t nkw, int64_t** ids, int64_t** counts,
                 size_t* nrows);

/* The same union, then the $correlation range filter (interface.cpp:137-143: corr_lo <= count < corr_hi) and
 * the final ranking (interface.cpp:144-146) on the device: rows by descending count, at most `limit` of them
 * (0 = all).  The reference ranks with an unstable std::sort, so its order among equal counts is arbitrary;
 * here ties ascend by object id (the shim's host-side rank_by_correlation reproduces the reference's own
 * tie order when that is wanted bit for bit).  Release ids/counts with cdb_free. */
int cdb_query_ranked(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t nkw, int64_t corr_lo, int64_t corr_hi,
                     uint64_t limit, int64_t** ids, int64_t** counts, size_t* nrows);

/* AND across keys — replaces the second half of filter() (interface.cpp:114-146): the per-key row lists are intersected
 * by object id with the counts added up, then optionally filtered by the $correlation range and ranked, all on the
 * device.  A key is either a string column (index + its keyword list: resolved here with the OR of cdb_query_or) or a
 * row list resolved elsewhere (index = NULL: numeric and bool keys return (id, 0) rows, ascending by id, each id once).
 * ranked = 0: rows ascending by object id (the list filter() holds after line 134; corr_* and limit ignored);
 * ranked = 1: corr_lo <= count < corr_hi, descending count, ties ascending id, at most `limit` rows (0 = all).
 * All string keys must live on one GPU; errors are reported on the first string key's handle.  Release ids / counts
 * with cdb_free. */
typedef struct cdb_key_query {
    cdb_index* index;        /* string key, or NULL */
    const char* blob;        /* its keywords: blob[offsets[j] .. offsets[j+1]) */
    const uint64_t* offsets;
    uint64_t nkw;
    const int64_t* ids;      /* index == NULL: host rows */
    const int64_t* counts;
    size_t nrows;
} cdb_key_query;
int cdb_query_and(const cdb_key_query* keys, int nkeys, int ranked, int64_t corr_lo, int64_t corr_hi, uint64_t limit, int64_t** ids,
                  int64_t** counts, size_t* nrows);

/* Highlight spans — replaces the per-document re-scan of ac_automaton::render (database.cpp:58-76) that
 * select() runs for every returned object (database.cpp:394-441): for the keyword list of ONE string
 * key, every matching document's merged highlight spans [begin, end] (byte offsets, end inclusive), with
 * the reference's merge rule — overlapping keyword occurrences fuse, merely adjacent ones do not.
 * Documents ascend by insertion index; document r owns spans [span_ptr[r], span_ptr[r+1]).  The
 * occurrences come straight from the suffix-array ranges (offset = entry >> bits), no text is scanned. */
typedef struct cdb_spans {
    uint64_t ndocs, nspans;
    int64_t* ids;        /* ndocs */
    uint64_t* span_ptr;  /* ndocs + 1 */
    uint64_t* begin;     /* nspans */
    uint64_t* end;       /* nspans, inclusive */
} cdb_spans;
int cdb_query_spans(cdb_index* h, const char* blob, const uint64_t* offsets, uint64_t nkw, cdb_spans* out);
void cdb_spans_free(cdb_spans* r);

/* Batched query with patterns and results left in device memory (multi-GPU merge over RCCL, HBM-
 * resident timing).  d_blob/d_offsets are device pointers.  On return the library-owned device arrays
 * d_row_ptr (npat+1 u64), d_ids (nrows i64), d_counts (nrows i64) stay valid until the next query on
 * this handle or cdb_destroy.  The offsets are not pre-validated on the host: an empty pattern yields zero
 * rows here (the host entry points reject it like the reference does). */
typedef struct cdb_device_result {
    uint64_t npat, nrows, nhits;
    const uint64_t* d_row_ptr;
    const int64_t* d_ids;
    const int64_t* d_counts;
} cdb_device_result;
int cdb_query_batch_device(cdb_index* h, const void* d_blob, const uint64_t* d_offsets, uint64_t npat,
                           uint64_t blob_bytes, cdb_device_result* out);

/* The same with occurrence offsets (cdb_query_batch_offsets), everything left in device memory: d_hit_ptr
 * (nrows + 1 u64) and d_offsets (nhits u64) are library-owned like the arrays of cdb_device_result. */
typedef struct cdb_device_hits {
    const uint64_t* d_hit_ptr;
    const uint64_t* d_offsets;
} cdb_device_hits;
int cdb_query_batch_offsets_device(cdb_index* h, const void* d_blob, const uint64_t* d_offsets, uint64_t npat,
                                   uint64_t blob_bytes, cdb_device_result* out, cdb_device_hits* hits);

/* ---- several GPUs: document-aligned shards (no reference counterpart: the reference is one process on one host) ----
 * Suffixes never cross documents (src/index.h:61-65) and a result row belongs to one document (src/index.cpp:317-321),
 * so a column splits into document-aligned byte ranges, one independent suffix array per GPU.  Every shard answers
 * the whole pattern batch; the per-shard match lists are merged on the devices: all-gather of the row counts, one scan,
 * all-gatherv of the rows over RCCL / xGMI (grouped broadcasts, every peer one hop away), a placement kernel — rows of a
 * pattern stay ascending in document order because shards are document ranges.
 *
 * cdb_shards: ONE process owning several GPUs — what replaces string_index when the column exceeds one GPU (the shim
 * switches to it).  Same surface as cdb_index: add / build / query / query_batch.  build() spreads the
 * column over as many devices as max_shard_bytes requires (option "max_shard_bytes", default 12 GiB per GPU;
 * "use_all_devices" = 1 spreads over all of them regardless); other options are forwarded to every shard. */
typedef struct cdb_shards cdb_shards;
int cdb_shards_create(cdb_shards** out, const int* devices, int ndev); /* devices may repeat (tests on a one-GPU box) */
void cdb_shards_destroy(cdb_shards* h);
const char* cdb_shards_last_error(const cdb_shards* h);
int cdb_shards_add(cdb_shards* h, int64_t id, const char* value, size_t len);                 /* index.cpp:174-177 */
int cdb_shards_add_bulk(cdb_shards* h, const int64_t* ids, const char* blob, const uint64_t* doc_start, uint64_t ndocs);
int cdb_shards_set_option(cdb_shards* h, const char* name, int64_t value);
int cdb_shards_build(cdb_shards* h);                                                           /* index.cpp:178-236 */
/* build straight from the caller's separate strings (cdb_build_views per shard; replaces whatever cdb_shards_add* staged) */
int cdb_shards_build_views(cdb_shards* h, const int64_t* ids, const char* const* ptrs, const uint64_t* lens, uint64_t ndocs);
int cdb_shards_query(cdb_shards* h, const char* keyword, size_t len, int64_t** ids, int64_t** counts, size_t* nrows);
int cdb_shards_query_batch(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out);
/* cdb_query_or / cdb_query_ranked / cdb_query_spans over all shards (object ids are disjoint across shards: the shard
 * answers are concatenated and ordered the way the single-GPU calls order them) */
int cdb_shards_query_or(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t nkw, int64_t** ids, int64_t** counts,
                        size_t* nrows);
int cdb_shards_query_ranked(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t nkw, int64_t corr_lo, int64_t corr_hi,
                            uint64_t limit, int64_t** ids, int64_t** counts, size_t* nrows);
int cdb_shards_query_spans(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t nkw, cdb_spans* out);
/* cdb_query_batch_offsets over all shards (occurrence offsets are relative to their document, so they need no re-basing) */
int cdb_shards_query_batch_offsets(cdb_shards* h, const char* blob, const uint64_t* offsets, uint64_t npat, cdb_result* out,
                                   cdb_hits* hits);
/* AND across keys (interface.cpp:114-146) with sharded string keys: a key is a sharded column + its keyword list, or a
 * row list resolved elsewhere (shards = NULL), exactly like cdb_key_query.  Different columns are cut at different
 * documents, so every sharded key is first resolved with its own OR over its shards; the row lists then meet in one device
 * merge (on the first shard of the first sharded key).  Same result rules as cdb_query_and. */
typedef struct cdb_shards_key_query {
    cdb_shards* shards;      /* sharded string key, or NULL */
    const char* blob;
    const uint64_t* offsets;
    uint64_t nkw;
    const int64_t* ids;      /* shards == NULL: host rows */
    const int64_t* counts;
    size_t nrows;
} cdb_shards_key_query;
int cdb_shards_query_and(const cdb_shards_key_query* keys, int nkeys, int ranked, int64_t corr_lo, int64_t corr_hi, uint64_t limit,
                         int64_t** ids, int64_t** counts, size_t* nrows);
/* raw-file ingest into the sharded column (cdb_add_raw_dir, database.cpp:170-275) and persistence (cdb_save / cdb_load per
 * shard: `path` holds the shard count and document bounds, `path.<i>` shard i's file).  Loading replaces the column;
 * a failed load leaves the serving shards untouched. */
int cdb_shards_add_raw_dir(cdb_shards* h, const char* dir, const char* key, uint64_t* records, uint64_t* added);
int cdb_shards_save(cdb_shards* h, const char* path);
int cdb_shards_load(cdb_shards* h, const char* path);
/* introspection: shards in use after build, the handle of shard i (per-shard parity: its suffix array is that of its
 * documents alone), its first document, and how the shards exchange ("rccl" / "device copies" / "none") */
int cdb_shards_count(const cdb_shards* h);
cdb_index* cdb_shards_get(cdb_shards* h, int i);
uint64_t cdb_shards_first_doc(const cdb_shards* h, int i);
const char* cdb_shards_transport(const cdb_shards* h);

/* cdb_comm: one process per GPU (bench.py under torchrun; MPI-style services).  Rank 0 draws a 128-byte id
 * (ncclGetUniqueId) and distributes it by whatever means the processes share; every rank then creates its communicator
 * (ncclCommInitRank) and merges its shard's device-resident result with the others' — a collective call.  On return
 * `merged` (identical on every rank) points to device arrays owned by the communicator, valid until its next merge. */
typedef struct cdb_comm cdb_comm;
int cdb_comm_unique_id(void* id128);
int cdb_comm_create(cdb_comm** out, const void* id128, int rank, int world, int device);
/* The same communicator for ranks that live in ONE process (a host thread per rank calls the collectives): out[world]
 * receives one handle per rank, rank i on devices[i].  Distinct devices exchange over RCCL (ncclCommInitAll); ranks that
 * share a device — e.g. BASELINE config 3's four 8 GiB shards co-resident on one MI355X — through device-to-device copies
 * (cdb_comm_transport says which).  Destroy every handle with cdb_comm_destroy. */
int cdb_comm_create_group(cdb_comm** out, int world, const int* devices);
void cdb_comm_destroy(cdb_comm* c);
const char* cdb_comm_last_error(const cdb_comm* c);
int cdb_comm_merge(cdb_comm* c, const cdb_device_result* local, cdb_device_result* merged);
/* Counts-only merge for consumers that keep (or download) their own slice — SURVEY §8e: when the consumer is not on the
 * GPU "each GPU D2H's its slice".  Collective; exchanges nothing but the per-pattern row counts (npat x u32 per rank).
 * Every rank learns the merged row_ptr and, per pattern, where ITS rows start in the merged row stream:
 * merged rows [d_row_base[j], d_row_base[j] + local count of j) are this rank's rows of pattern j, in order.  No rank holds
 * another rank's rows (the full all-gatherv of C4 — 10^7 patterns x 8 shards — would put ~10^9 rows on every GPU). */
typedef struct cdb_shard_slice {
    uint64_t npat, nrows_total, nrows_local;
    const uint64_t* d_row_ptr;   /* npat + 1, merged, identical on every rank; device memory owned by the communicator */
    const uint64_t* d_row_base;  /* npat, this rank's first merged row of every pattern */
} cdb_shard_slice;
int cdb_comm_merge_counts(cdb_comm* c, const cdb_device_result* local, cdb_shard_slice* out);
int cdb_comm_world(const cdb_comm* c);
const char* cdb_comm_transport(const cdb_comm* c);

/* ---- introspection (parity tests; mirrors the private members src/index.h:56-60) -------------- */
uint64_t cdb_size(const cdb_index* h);  /* number of suffixes = text bytes */
uint64_t cdb_bits(const cdb_index* h);  /* doc-index bits of an entry */
uint64_t cdb_mask(const cdb_index* h);
int cdb_sa_width(const cdb_index* h);   /* 4 or 8 bytes per entry; 0 before build */
/* copies the suffix array ((offset << bits) | doc entries, cdb_sa_width bytes each) to host memory */
int cdb_sa_copy(cdb_index* h, void* host_out, uint64_t capacity_bytes);

/* ---- options & measurements -------------------------------------------------------------------- */
/* name: "profile" (0/1: time kernels with HIP events), "reference_compat" (default 1: reproduce the
 * reference's signed-char bucket order for text with bytes >= 0x80 — SURVEY.md Q2 — so that the suffix
 * array and the (then partly wrong) counts are bit-identical to the reference's; 0 = plain unsigned
 * order with true counts; no effect and no cost on pure-ASCII text), "initial_passes" (radix
 * passes of the initial key sort, 0 = automatic), "force_doubling" (0/1), "sort_variant" (radix kernel
 * configuration, 0 = default); further tuning and test hooks are listed in DESIGN.md. */
int cdb_set_option(cdb_index* h, const char* name, int64_t value);

/* statistic by name: "build_ms", "rounds", "unresolved_after_initial", "sort_passes", "isa_built",
 * "key_symbols", "symbol_bits", "query_ms", ... returns CDB_E_INVALID for unknown names */
int cdb_get_stat(const cdb_index* h, const char* name, double* value);

/* accumulated HIP-event time of one kernel family since cdb_profile_reset (needs option profile=1).
 * bytes = algorithmic bytes moved by those launches. */
int cdb_profile_get(cdb_index* h, const char* kernel, double* total_ms, uint64_t* launches, uint64_t* bytes);
/* writes up to cap bytes of a NUL-terminated, newline-separated "name ms launches bytes" table */
int cdb_profile_dump(cdb_index* h, char* buf, size_t cap);
void cdb_profile_reset(cdb_index* h);

/* The first build of a fresh process pays for device memory the driver maps (and scrubs) on first use: 0.3-2.6 s for a 4 GiB
 * column against 0.13 s warm.  The reference loads its data and only then builds (server.cpp:43-44: init(); build();), so
 * that cost can hide behind the ingest: cdb_reserve — called as soon as the size of the column is roughly known, e.g. with
 * the size of the raw directory at the start of init() — builds and destroys a throw-away index over synthetic text of that
 * size on a helper thread (bytes drawn from the byte histogram of `sample`, e.g. the first document; NULL = printable ASCII;
 * ndocs = 0: 1 KiB documents), which leaves the build's working set in the block cache below — and then maps, as spare blocks
 * of that cache, twins of the arrays an index of that size KEEPS: the first `build` operation after start-up constructs the
 * next generation while this one serves (database.cpp:276-280) and would otherwise pay 0.6-1 s of hipMalloc for them (4 GiB
 * column: second build 947 -> 244 ms).  Skipped where the device has no room for it.  Returns at once; cdb_build* / cdb_load
 * wait for a reservation in flight, cdb_reserve_wait() does so explicitly.  Best effort. */
int cdb_reserve(int device, uint64_t text_bytes, uint64_t ndocs, const char* sample, size_t sample_len);
void cdb_reserve_wait(void);

/* Device blocks released by builds/queries are cached process-wide for the next build (hipMalloc of the
 * ~30 GiB working set of a 1 GiB build costs ~1 s on MI355X — the driver maps and clears VRAM).  This
 * returns every cached block to the driver; cdb_cached_memory_bytes reports the cache size.  Result
 * arrays of 1 MiB and more (cdb_query_batch*, cdb_query_or, cdb_query_spans) are pinned host blocks from a
 * similar cache (up to 8 GiB kept; device-to-host copies into fresh pageable memory run at a few GB/s);
 * always release them through cdb_result_free / cdb_hits_free / cdb_spans_free / cdb_free. */
void cdb_release_cached_memory(void);
uint64_t cdb_cached_memory_bytes(void);
/* upper bound of the block cache (default: unlimited); blocks released beyond it go back to the driver */
void cdb_set_cache_limit(uint64_t bytes);
/* Device memory of the whole process as the library's allocator sees it: bytes handed out right now (indexes + builds
 * in flight), the most ever handed out since cdb_memory_reset_peak(), and the bytes sitting in the block cache.  What
 * database.cpp:276-280 needs to know before it builds a new index beside the serving one: peak - in_use of one build is
 * the head room a rebuild wants.  Buffers the caller owns (cdb_build_device / _resident: the text) are not counted. */
void cdb_memory_stats(uint64_t* in_use_bytes, uint64_t* peak_bytes, uint64_t* cached_bytes);
void cdb_memory_reset_peak(void);

/* Test hook: size-independent checks of the built suffix array, computed on the GPU by plain adjacent-
 * suffix comparison (verify.hip).  out[0] = adjacent pairs out of unsigned byte order, out[1] = equal
 * suffixes not ascending by document, out[2] = wrapped sum of all entries, out[3] = entries that are not
 * a valid (doc, off), out[4] = closed-form expected value of out[2].  (With reference_compat and bytes
 * >= 0x80 the reference's order is not globally sorted, so out[0] > 0 is expected there.) */
int cdb_debug_verify(cdb_index* h, uint64_t out[5]);

/* Test hook: the same kind of check against the REFERENCE's order (reference_compat = 1): inside a radix
 * node — a bucket of more than chuck_size = max(4096, n / 256) suffixes (index.cpp:96-126,218) — children follow the
 * signed-char symbol order of index.h:66-73 (end of document, 0x80..0xFF, 0x00..0x7F); smaller buckets are in unsigned
 * order (std::sort leaves, index.cpp:86-95).  Adjacent pairs are classified by their common prefix and, where the two
 * orders disagree, by the size of the bucket sharing that prefix (found by galloping over the array).
 * out[0] = pairs out of reference order (0 for a correct array), out[1] = pairs whose next bytes lie on different
 * sides of 0x80, out[2] = of those, pairs inside radix nodes (laid out high byte first), out[3] = equal suffixes not
 * ascending by document.  On pure-ASCII text, or with reference_compat = 0 on text without big mixed buckets, it
 * degenerates to the plain sortedness check. */
int cdb_debug_verify_reference(cdb_index* h, uint64_t out[4]);
/* Test hook: the check every build ends with (option self_check), run on the index as it stands — on 2^15 random adjacent
 * pairs, or (full != 0) on every adjacent pair.  out[0] = pairs out of order, out[1] = entries that are no valid (doc, off). */
int cdb_debug_self_check(cdb_index* h, int full, uint64_t out[2]);

/* The order proof behind a published build (option self_check = 3, the default).  The reference's array is sorted by
 * construction (std::sort leaves, index.cpp:86-95); this library's passes rest on an observed LDS lane order, so after
 * cdb_build* / cdb_load return (with a sample of adjacent pairs checked) a helper thread compares EVERY adjacent pair of the
 * published array against the text on a low-priority stream of its own, beside the queries (text with bytes >= 0x80 in the
 * reference's order: also the pairs whose order depends on the size of the reference's radix buckets, index.h:66-73).  cdb_get_stat "order_proved" goes
 * 0 -> 1 when it is through; damage makes the handle rebuild itself with the ballot ranking under its lock (queries wait; stat
 * "self_check_fallbacks" counts it).  cdb_proof_wait blocks until the proof of the CURRENT array has ended, at most timeout_ms
 * (< 0: no limit), and returns its state: 0 no proof was started (self_check < 3, or not built), 1 still running, 2 proved,
 * 3 damage found and repaired (the replacement was checked pair by pair before it was served), 4 damage found and the rebuild
 * failed (the handle is unbuilt), 5 cancelled by a build that replaced the array, 6 the proof could not run; < 0: an error. */
int cdb_proof_wait(cdb_index* h, double timeout_ms);

/* Test hook for the radix-sort primitive (tests/test_gpu_sort.py, tools/sort_bench.py): stable sort of
 * n 64-bit keys (+ optional 4- or 8-byte values, val_bytes = 0/4/8) held in DEVICE memory by key bits
 * [0, key_bits), in place.  variant = kernel configuration (0 = default).  Reports the summed HIP-event
 * time of the onesweep launches and how many passes ran. */
int cdb_debug_radix_sort(int device, void* d_keys, void* d_vals, uint64_t n, int val_bytes, int key_bits,
                         int variant, double* onesweep_ms, int* passes);

/* Measurement hook: wall time of `reps` back-to-back cdb_query calls for each of `nkw` keywords (blob + offsets like
 * cdb_query_batch), taken inside the library with a steady clock — what a C++ caller such as database.cpp:392 sees,
 * without a language binding in between.  us_out[nkw] receives the MEDIAN microseconds per call of every keyword;
 * results are freed immediately. */
int cdb_debug_query_latency(cdb_index* h, const char* blob, const uint64_t* offsets, size_t nkw, int reps, double* us_out);

/* The build prologue's layout rule as a pure host function (no device, no handle): entry layout of an index over
 * `ndocs` documents whose longest has `longest` bytes — index.cpp:182-208: masks grown by `mask = (mask << 1) + 1`,
 * bits = popcount, 4-byte entries while bits + offset bits <= 32.  Returns CDB_OK and fills bits / mask / width /
 * off_bits, or CDB_E_INVALID with the reference's own message in `err` (index.cpp:195-200: "The amount of data
 * exceeds the maximum range that CoffeeDB can handle" beyond 64 bits, "The number of objects exceeds ..." beyond 2^32
 * documents).  What cdb_build* apply before they touch the device. */
int cdb_layout_rule(uint64_t ndocs, uint64_t longest, uint64_t* bits, uint64_t* mask, int* width, int* off_bits,
                    char* err, size_t err_cap);

#ifdef __cplusplus
}
#endif
#endif /* COFFEEDB_GPU_H */
