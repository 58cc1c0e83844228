// highlight.h — rendering of highlight spans, the second half of ac_automaton::render (reference
// src/database.cpp:77-90): `left` is inserted before the first byte of every span and `right` after its
// last byte.  The spans themselves come from string_index::highlight_spans() (one GPU call for all
// matching documents, cdb_query_spans) instead of an Aho–Corasick pass over every selected document.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace cdb_shim {

using spans_t = std::vector<std::pair<uint64_t, uint64_t>>;  // [begin, end], end inclusive, ascending

inline std::string render_spans(const std::string& text, const spans_t& spans, const std::string& left,
                                const std::string& right) {
    std::string out;
    out.reserve(text.size() + (left.size() + right.size()) * spans.size() + 1);
    auto it = spans.begin();
    for (uint64_t i = 0; i < text.size(); ++i) {
        if (it != spans.end() && i == it->first) out += left;
        out += text[i];
        if (it != spans.end() && i == it->second) {
            out += right;
            ++it;
        }
    }
    return out;
}

}  // namespace cdb_shim
