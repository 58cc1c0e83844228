// index.cpp — the reference-side binding: CoffeeDB's index classes on top of the C ABI of
// libcoffeedb_gpu.so.  Replaces /root/reference/src/index.cpp; everything the reference computed on
// the CPU for string columns (index.cpp:75-128, 174-326) now happens behind cdb_*.
#include "index.h"

#include <algorithm>
#include <cstdlib>

#include "../../../include/coffeedb_gpu.h"
#ifdef CDB_USE_REFERENCE_UTILITY
#include "utility.h"  // the reference's own parse_range (src/utility.h:69-86)
#else
#include "range_parse.h"
using cdb_shim::parse_range;
#endif

// ---- base class (index.h:16-21)
index::result_type index::query(const std::string&) const { throw std::logic_error("Unimplemented method index::query"); }
void index::build() { throw std::logic_error("Unimplemented method index::build"); }

// ---- numeric indexes: sorted (value, id) pairs, half-open lower_bound window (index.cpp:63-74, 129-173)
template <typename T, int8_t Tag>
void numeric_index<T, Tag>::build() {
    std::sort(rows.begin(), rows.end());
    rows.shrink_to_fit();
}
template <typename T, int8_t Tag>
index::result_type numeric_index<T, Tag>::query(const std::string& range) const {
    const auto [lo, hi] = parse_range<T>(range);
    const auto first = std::lower_bound(rows.begin(), rows.end(), lo);
    const auto last = std::lower_bound(rows.begin(), rows.end(), hi);
    result_type out;
    if (first < last) out.reserve((size_t)(last - first));
    for (auto it = first; it < last; ++it) out.emplace_back(it->second, 0);
    return out;
}
template class numeric_index<int64_t, 1>;
template class numeric_index<double, 2>;

void bool_index::add(int64_t id, bool value) { data[value ? 1 : 0].push_back(id); }
void bool_index::build() {
    data[0].shrink_to_fit();
    data[1].shrink_to_fit();
}
index::result_type bool_index::query(const std::string& range) const {
    int which = -1;
    if (range == "false") which = 0;
    if (range == "true") which = 1;
    if (which < 0) throw std::runtime_error("Invalid query: \"" + range + "\"");  // index.cpp:146
    result_type out;
    out.reserve(data[which].size());
    for (int64_t id : data[which]) out.emplace_back(id, 0);
    return out;
}

// ---- string index: forwards to the GPU library
namespace {
[[noreturn]] void rethrow_msg(const std::string& msg, int rc) {
    // the library reports the reference's own wording for conditions the reference throws on
    if (rc == CDB_E_INVALID) throw std::runtime_error(msg);
    throw std::runtime_error("GPU index: " + msg);
}
[[noreturn]] void rethrow(const cdb_index* h, int rc) { rethrow_msg(cdb_last_error(h), rc); }
[[noreturn]] void rethrow(const cdb_shards* h, int rc) { rethrow_msg(cdb_shards_last_error(h), rc); }

// COFFEEDB_GPUS = "4" (devices 0..3) or "0,2,5": string columns may spread over these GPUs.  The library shards a
// column only once it exceeds what one GPU should hold (cdb_shards: max_shard_bytes), so small columns keep living
// on the first device.  Unset or one device: plain single-GPU handles.
std::vector<int> shard_devices() {
    std::vector<int> dev;
    const char* e = std::getenv("COFFEEDB_GPUS");
    if (!e || !*e) return dev;
    const std::string v(e);
    if (v.find(',') == std::string::npos) {
        const int n = std::atoi(v.c_str());
        for (int i = 0; i < n; ++i) dev.push_back(i);
    } else {
        size_t p = 0;
        while (p < v.size()) {
            const size_t q = v.find(',', p);
            dev.push_back(std::atoi(v.substr(p, q == std::string::npos ? std::string::npos : q - p).c_str()));
            if (q == std::string::npos) break;
            p = q + 1;
        }
    }
    if (dev.size() < 2) dev.clear();
    return dev;
}

std::pair<std::string, std::vector<uint64_t>> pack(const std::vector<std::string>& keywords) {
    std::pair<std::string, std::vector<uint64_t>> out;
    out.second.push_back(0);
    for (const auto& k : keywords) {
        out.first += k;
        out.second.push_back(out.first.size());
    }
    return out;
}
}  // namespace

string_index::string_index() {
    const std::vector<int> dev = shard_devices();
    if (!dev.empty()) {
        if (cdb_shards_create(&shards, dev.data(), (int)dev.size()) != CDB_OK || !shards)
            throw std::runtime_error("GPU index: no usable MI355X (gfx950) devices for COFFEEDB_GPUS");
        if (const char* all = std::getenv("COFFEEDB_SHARD_ALL"); all && *all == '1')  // spread even small columns (tests)
            (void)cdb_shards_set_option(shards, "use_all_devices", 1);
        if (const char* r = std::getenv("COFFEEDB_RESIDENT_QUERY"); r && *r == '1') (void)cdb_shards_set_option(shards, "resident_query", 1);
        return;
    }
    if (cdb_create(&handle, -1) != CDB_OK || !handle)
        throw std::runtime_error("GPU index: no usable MI355X (gfx950) device");
    // The resident workgroup that answers lone query() calls (database.cpp:392) in ~6.7 us instead of ~12 us per call switches
    // itself on for keywords arriving back to back (library default resident_query = 2); COFFEEDB_RESIDENT_QUERY=1 pins it on
    // from the first keyword (INTEGRATION.md says what it costs)
    if (const char* r = std::getenv("COFFEEDB_RESIDENT_QUERY"); r && *r == '1') (void)cdb_set_option(handle, "resident_query", 1);
}
void string_index::reserve(uint64_t bytes, std::string_view sample) {
    if (!shard_devices().empty()) return;  // (several GPUs: every shard maps its own share when it builds)
    (void)cdb_reserve(-1, bytes, 0, sample.data(), sample.size());
}
bool string_index::settle() const {
    if (!handle) return true;  // (several GPUs: every shard settles behind its own build)
    const int st = cdb_proof_wait(handle, -1.0);
    return st == 2 || st == 3 || st == 0;
}
string_index::~string_index() {
    cdb_destroy(handle);
    cdb_shards_destroy(shards);
}

// index.cpp:174-177: like the reference, add() only remembers the id and a VIEW of the caller's string (database.cpp keeps
// the strings alive in its `data` map, database.cpp:262-264); build() hands the views to the library, which gathers them
// straight into its pinned upload chunks — no staging copy of the column on the host.
void string_index::add(int64_t id, std::string_view value) {
    ids.push_back(id);
    ptrs.push_back(value.data());
    lens.push_back(value.size());
}

void string_index::build() {
    if (shards) {
        const int rc = cdb_shards_build_views(shards, ids.data(), ptrs.data(), lens.data(), ids.size());
        if (rc != CDB_OK) rethrow(shards, rc);
        return;
    }
    const int rc = cdb_build_views(handle, ids.data(), ptrs.data(), lens.data(), ids.size());
    if (rc != CDB_OK) rethrow(handle, rc);
}

index::result_type string_index::query(const std::string& keyword) const {
    int64_t *ids = nullptr, *counts = nullptr;
    size_t rows = 0;
    if (shards) {
        const int rc = cdb_shards_query(shards, keyword.data(), keyword.size(), &ids, &counts, &rows);
        if (rc != CDB_OK) rethrow(shards, rc);
    } else {
        const int rc = cdb_query(handle, keyword.data(), keyword.size(), &ids, &counts, &rows);
        if (rc != CDB_OK) rethrow(handle, rc);
    }
    result_type out;
    out.reserve(rows);
    for (size_t r = 0; r < rows; ++r) out.emplace_back(ids[r], counts[r]);
    cdb_free(ids);
    cdb_free(counts);
    return out;
}

std::vector<index::result_type> string_index::query_batch(const std::vector<std::string>& keywords) const {
    const auto [blob, offs] = pack(keywords);
    cdb_result res;
    if (shards) {
        const int rc = cdb_shards_query_batch(shards, blob.data(), offs.data(), keywords.size(), &res);
        if (rc != CDB_OK) rethrow(shards, rc);
    } else {
        const int rc = cdb_query_batch(handle, blob.data(), offs.data(), keywords.size(), &res);
        if (rc != CDB_OK) rethrow(handle, rc);
    }
    std::vector<result_type> out(keywords.size());
    for (size_t j = 0; j < keywords.size(); ++j) {
        out[j].reserve((size_t)(res.row_ptr[j + 1] - res.row_ptr[j]));
        for (uint64_t r = res.row_ptr[j]; r < res.row_ptr[j + 1]; ++r) out[j].emplace_back(res.ids[r], res.counts[r]);
    }
    cdb_result_free(&res);
    return out;
}

// The per-key operations: one call either way (cdb_shards_* concatenate the shards' answers — documents, hence object
// ids, are disjoint across shards — and order them as the single-GPU calls do).
index::result_type string_index::query_any(const std::vector<std::string>& keywords) const {
    const auto [blob, offs] = pack(keywords);
    int64_t *ids = nullptr, *counts = nullptr;
    size_t rows = 0;
    if (shards) {
        const int rc = cdb_shards_query_or(shards, blob.data(), offs.data(), keywords.size(), &ids, &counts, &rows);
        if (rc != CDB_OK) rethrow(shards, rc);
    } else {
        const int rc = cdb_query_or(handle, blob.data(), offs.data(), keywords.size(), &ids, &counts, &rows);
        if (rc != CDB_OK) rethrow(handle, rc);
    }
    result_type out;
    out.reserve(rows);
    for (size_t r = 0; r < rows; ++r) out.emplace_back(ids[r], counts[r]);
    cdb_free(ids);
    cdb_free(counts);
    return out;
}

index::result_type string_index::query_ranked(const std::vector<std::string>& keywords, int64_t corr_lo, int64_t corr_hi,
                                              uint64_t limit) const {
    const auto [blob, offs] = pack(keywords);
    int64_t *ids = nullptr, *counts = nullptr;
    size_t rows = 0;
    if (shards) {
        const int rc = cdb_shards_query_ranked(shards, blob.data(), offs.data(), keywords.size(), corr_lo, corr_hi, limit, &ids, &counts, &rows);
        if (rc != CDB_OK) rethrow(shards, rc);
    } else {
        const int rc = cdb_query_ranked(handle, blob.data(), offs.data(), keywords.size(), corr_lo, corr_hi, limit, &ids, &counts, &rows);
        if (rc != CDB_OK) rethrow(handle, rc);
    }
    result_type out;
    out.reserve(rows);
    for (size_t r = 0; r < rows; ++r) out.emplace_back(ids[r], counts[r]);
    cdb_free(ids);
    cdb_free(counts);
    return out;
}

std::vector<std::pair<int64_t, std::vector<std::pair<uint64_t, uint64_t>>>> string_index::highlight_spans(
    const std::vector<std::string>& keywords) const {
    const auto [blob, offs] = pack(keywords);
    cdb_spans sp;
    if (shards) {
        const int rc = cdb_shards_query_spans(shards, blob.data(), offs.data(), keywords.size(), &sp);
        if (rc != CDB_OK) rethrow(shards, rc);
    } else {
        const int rc = cdb_query_spans(handle, blob.data(), offs.data(), keywords.size(), &sp);
        if (rc != CDB_OK) rethrow(handle, rc);
    }
    std::vector<std::pair<int64_t, std::vector<std::pair<uint64_t, uint64_t>>>> out(sp.ndocs);
    for (uint64_t d = 0; d < sp.ndocs; ++d) {
        out[d].first = sp.ids[d];
        for (uint64_t k = sp.span_ptr[d]; k < sp.span_ptr[d + 1]; ++k) out[d].second.emplace_back(sp.begin[k], sp.end[k]);
    }
    cdb_spans_free(&sp);
    return out;
}
