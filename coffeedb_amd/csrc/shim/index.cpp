// index.cpp — the reference-side binding: CoffeeDB's index classes on top of the C ABI of
// libcoffeedb_gpu.so.  Replaces /root/reference/src/index.cpp; everything the reference computed on
// the CPU for string columns (index.cpp:75-128, 174-326) now happens behind cdb_*.
#include "index.h"

#include <algorithm>

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
[[noreturn]] void rethrow(const cdb_index* h, int rc) {
    // the library reports the reference's own wording for conditions the reference throws on
    const std::string msg = cdb_last_error(h);
    if (rc == CDB_E_INVALID) throw std::runtime_error(msg);
    throw std::runtime_error("GPU index: " + msg);
}
}  // namespace

string_index::string_index() {
    if (cdb_create(&handle, -1) != CDB_OK || !handle)
        throw std::runtime_error("GPU index: no usable MI355X (gfx950) device");
}
string_index::~string_index() { cdb_destroy(handle); }

void string_index::add(int64_t id, std::string_view value) {
    const int rc = cdb_add(handle, id, value.data(), value.size());
    if (rc != CDB_OK) rethrow(handle, rc);
}

void string_index::build() {
    const int rc = cdb_build(handle);
    if (rc != CDB_OK) rethrow(handle, rc);
}

index::result_type string_index::query(const std::string& keyword) const {
    int64_t *ids = nullptr, *counts = nullptr;
    size_t rows = 0;
    const int rc = cdb_query(handle, keyword.data(), keyword.size(), &ids, &counts, &rows);
    if (rc != CDB_OK) rethrow(handle, rc);
    result_type out;
    out.reserve(rows);
    for (size_t r = 0; r < rows; ++r) out.emplace_back(ids[r], counts[r]);
    cdb_free(ids);
    cdb_free(counts);
    return out;
}

std::vector<index::result_type> string_index::query_batch(const std::vector<std::string>& keywords) const {
    std::string blob;
    std::vector<uint64_t> offs{0};
    for (const auto& k : keywords) {
        blob += k;
        offs.push_back(blob.size());
    }
    cdb_result res;
    const int rc = cdb_query_batch(handle, blob.data(), offs.data(), keywords.size(), &res);
    if (rc != CDB_OK) rethrow(handle, rc);
    std::vector<result_type> out(keywords.size());
    for (size_t j = 0; j < keywords.size(); ++j) {
        out[j].reserve((size_t)(res.row_ptr[j + 1] - res.row_ptr[j]));
        for (uint64_t r = res.row_ptr[j]; r < res.row_ptr[j + 1]; ++r) out[j].emplace_back(res.ids[r], res.counts[r]);
    }
    cdb_result_free(&res);
    return out;
}

index::result_type string_index::query_any(const std::vector<std::string>& keywords) const {
    std::string blob;
    std::vector<uint64_t> offs{0};
    for (const auto& k : keywords) {
        blob += k;
        offs.push_back(blob.size());
    }
    int64_t *ids = nullptr, *counts = nullptr;
    size_t rows = 0;
    const int rc = cdb_query_or(handle, blob.data(), offs.data(), keywords.size(), &ids, &counts, &rows);
    if (rc != CDB_OK) rethrow(handle, rc);
    result_type out;
    out.reserve(rows);
    for (size_t r = 0; r < rows; ++r) out.emplace_back(ids[r], counts[r]);
    cdb_free(ids);
    cdb_free(counts);
    return out;
}

index::result_type string_index::query_ranked(const std::vector<std::string>& keywords, int64_t corr_lo, int64_t corr_hi,
                                              uint64_t limit) const {
    std::string blob;
    std::vector<uint64_t> offs{0};
    for (const auto& k : keywords) {
        blob += k;
        offs.push_back(blob.size());
    }
    int64_t *ids = nullptr, *counts = nullptr;
    size_t rows = 0;
    const int rc = cdb_query_ranked(handle, blob.data(), offs.data(), keywords.size(), corr_lo, corr_hi, limit, &ids, &counts, &rows);
    if (rc != CDB_OK) rethrow(handle, rc);
    result_type out;
    out.reserve(rows);
    for (size_t r = 0; r < rows; ++r) out.emplace_back(ids[r], counts[r]);
    cdb_free(ids);
    cdb_free(counts);
    return out;
}

std::vector<std::pair<int64_t, std::vector<std::pair<uint64_t, uint64_t>>>> string_index::highlight_spans(
    const std::vector<std::string>& keywords) const {
    std::string blob;
    std::vector<uint64_t> offs{0};
    for (const auto& k : keywords) {
        blob += k;
        offs.push_back(blob.size());
    }
    cdb_spans sp;
    const int rc = cdb_query_spans(handle, blob.data(), offs.data(), keywords.size(), &sp);
    if (rc != CDB_OK) rethrow(handle, rc);
    std::vector<std::pair<int64_t, std::vector<std::pair<uint64_t, uint64_t>>>> out(sp.ndocs);
    for (uint64_t d = 0; d < sp.ndocs; ++d) {
        out[d].first = sp.ids[d];
        for (uint64_t k = sp.span_ptr[d]; k < sp.span_ptr[d + 1]; ++k) out[d].second.emplace_back(sp.begin[k], sp.end[k]);
    }
    cdb_spans_free(&sp);
    return out;
}
