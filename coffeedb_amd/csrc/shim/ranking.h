// ranking.h — the steps that follow the per-key query in CoffeeDB's filter() (reference
// src/interface.cpp:114-146), restated for callers of the GPU index: AND across keys, the $correlation
// range filter, and the final ranking.  These stay on the host on purpose: the reference ranks with an
// UNSTABLE std::sort over the id-sorted list (interface.cpp:144-146), so calling the same libstdc++
// std::sort with the same comparator on the same input is the only way to reproduce its order among
// equal counts bit for bit; the heavy part — resolving and OR-merging the keywords of a key — is what
// string_index::query_any() moves to the GPU (cdb_query_or).
#pragma once
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

namespace cdb_shim {

using rows_t = std::vector<std::pair<int64_t, int64_t>>;

// AND across keys (interface.cpp:114-134): ids present in both id-sorted lists, counts summed.
inline rows_t and_merge(const rows_t& result, const rows_t& answer) {
    rows_t out;
    for (size_t i = 0, j = 0; i < result.size() && j < answer.size();) {
        if (result[i].first == answer[j].first) {
            out.emplace_back(result[i].first, result[i].second + answer[j].second);
            ++i;
            ++j;
        } else if (result[i] < answer[j]) {
            ++i;
        } else {
            ++j;
        }
    }
    return out;
}

// $correlation range filter (interface.cpp:137-143): keep L <= count < R.
inline void correlation_filter(rows_t& answer, int64_t L, int64_t R) {
    answer.erase(std::remove_if(answer.begin(), answer.end(),
                                [L, R](const std::pair<int64_t, int64_t>& p) { return !(p.second >= L && p.second < R); }),
                 answer.end());
}

// final ranking (interface.cpp:144-146): descending $correlation, the reference's own unstable sort.
inline void rank_by_correlation(rows_t& answer) {
    std::sort(answer.begin(), answer.end(), [](auto x, auto y) { return x.second > y.second; });
}

}  // namespace cdb_shim
