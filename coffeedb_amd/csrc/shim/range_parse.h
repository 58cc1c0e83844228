// range_parse.h — numeric range syntax of CoffeeDB queries, restated from the reference's
// src/utility.h:49-86 (value_conv / parse_range) so that the numeric indexes of this shim behave the
// same when it is compiled outside the reference tree.  Grammar: optional blanks, '[' or '(', lower
// value, ',', upper value, ']' or ')', optional blanks; values are from_chars numbers or inf / -inf
// (case-insensitive).  Inside the reference tree index.cpp can include the reference's utility.h
// instead (define CDB_USE_REFERENCE_UTILITY).
#pragma once
#include <algorithm>
#include <cctype>
#include <charconv>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>

namespace cdb_shim {

template <typename T>
void convert_value(std::string text, T& out) {
    for (auto& c : text) c = (char)std::tolower((unsigned char)c);
    if (text == "-inf") {
        out = std::numeric_limits<T>::min();  // sic: for double this is the smallest positive value (utility.h:54-56)
    } else if (text == "inf") {
        out = std::numeric_limits<T>::max();
    } else {
        const char* b = text.data();
        const char* e = b + text.size();
        auto [ptr, ec] = std::from_chars(b, e, out);
        if (ec != std::errc{} || ptr != e) throw std::runtime_error("Invalid value: " + text);
    }
}

// Returns ((lower, tag), (upper, tag)) for lower_bound over (value, id) pairs: an exclusive lower
// bound and an inclusive upper bound carry tag INT64_MAX so that every id with that value is skipped
// respectively included (utility.h:69-86).
template <typename T>
std::pair<std::pair<T, int64_t>, std::pair<T, int64_t>> parse_range(const std::string& range) {
    auto fail = [&]() -> void { throw std::runtime_error("Invalid range: " + range); };
    size_t i = 0, n = range.size();
    auto blanks = [&] { while (i < n && std::isspace((unsigned char)range[i])) ++i; };
    blanks();
    if (i >= n || (range[i] != '[' && range[i] != '(')) fail();
    const bool open_low = range[i++] == '(';
    blanks();
    // the reference's regex is greedy: the lower value runs up to the LAST comma that still leaves a
    // closing bracket behind it
    size_t end = n;
    while (end > i && std::isspace((unsigned char)range[end - 1])) --end;
    if (end <= i || (range[end - 1] != ']' && range[end - 1] != ')')) fail();
    const bool closed_high = range[end - 1] == ']';
    const size_t body_end = end - 1;
    const size_t comma = range.rfind(',', body_end == 0 ? 0 : body_end - 1);
    if (comma == std::string::npos || comma < i) fail();
    // ".+" on both sides: at least one character each (blanks before the comma belong to the value
    // in the reference's regex and make from_chars fail there too)
    std::string lo = range.substr(i, comma - i);
    size_t j = comma + 1;
    while (j < body_end && std::isspace((unsigned char)range[j])) ++j;
    std::string hi = range.substr(j, body_end - j);
    if (lo.empty() || hi.empty()) fail();
    std::pair<T, int64_t> L{}, R{};
    convert_value(lo, L.first);
    convert_value(hi, R.first);
    if (open_low) L.second = std::numeric_limits<int64_t>::max();
    if (closed_high) R.second = std::numeric_limits<int64_t>::max();
    return {L, R};
}

}  // namespace cdb_shim
