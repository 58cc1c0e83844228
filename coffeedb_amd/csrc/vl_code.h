// vl_code.h — order-preserving variable-length symbol codes for the sort keys of skewed text (round 5).
//
// The bucket-wise build (sa_build.hip) sorts every first-symbol bucket on a fixed-width key made of the next symbols.  Coded as
// a dense base-(alphabet + 1) number a symbol costs log2(alphabet + 1) bits whatever its frequency: the 64-symbol Zipf text of
// BASELINE config 2 pays 6.02 bits per symbol for 4.86 bits of entropy, and because equal prefixes are made of FREQUENT symbols
// the key bits buy even less resolution than that ratio says (0.45 bits of collision exponent per key bit).  An ALPHABETIC prefix
// code — codewords in the order of the symbols, so that the concatenated code words of two strings compare like the strings —
// gives frequent symbols short words: the first B bits of the code stream then cover more symbols exactly where collisions
// happen (0.87 bits per key bit on that text).  40 key bits resolve what 54 bits of the dense number do: two radix passes fewer,
// over 10- instead of 12-byte records.
//
// The code is the optimal alphabetic tree of Garsia and Wachs (Hu-Tucker's result by a simpler procedure) over the symbols
// END < code 1 < ... < code sigma with their counts as weights, its depth limited to VL_MAX_LEN bits by raising the weight floor
// (bit offsets inside a tile are 16-bit).  END — the end of a document — is the first leaf, so its word is all zeros: a suffix
// that ends inside the key is simply padded with zeros, which is END followed by nothing, and sorts in front of every
// continuation.  reference: the keys are the radix-node symbols of src/index.cpp:96-126 / src/index.h:66-73 under another
// order-preserving coding; DESIGN.md §4.2 "Variable-length keys".
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace cdb {

constexpr int VL_MAX_LEN = 7;  // longest code word in bits: (8192 + 64) positions x 7 bits < 2^16 (tile-local bit offsets are u16)

struct VlCode {
    int nsyms = 0;            // alphabet + 1 (END first)
    uint8_t len[257] = {0};   // [0] = END, [c] = symbol code c
    uint16_t bits[257] = {0};
    int end_len = 0, max_len = 0, min_len = 0;
    double avg_len = 0;       // expected bits per text symbol
    double rate = 0;          // collision exponent per key bit: two random suffixes agree on B code bits with probability ~ q 2^(-rate B)
    double q = 0;             // sum of p^2 (the first symbol, which is the bucket)
};

// leaf depths of an optimal alphabetic tree (Garsia-Wachs), weights in symbol order
inline std::vector<int> vl_gw_depths(const std::vector<double>& w) {
    const int n = (int)w.size();
    if (n == 1) return {1};
    struct Node { double w; int id; };
    std::vector<Node> lst;
    std::vector<std::pair<int, int>> child;  // internal node id - n -> children
    for (int i = 0; i < n; ++i) lst.push_back({w[i], i});
    int next = n;
    while (lst.size() > 1) {
        size_t i = 1;
        while (i + 1 < lst.size() && lst[i - 1].w > lst[i + 1].w) ++i;  // leftmost locally minimal pair
        const Node nn{lst[i - 1].w + lst[i].w, next++};
        child.push_back({lst[i - 1].id, lst[i].id});
        lst.erase(lst.begin() + (i - 1), lst.begin() + (i + 1));
        long j = (long)i - 2;
        while (j >= 0 && lst[j].w < nn.w) --j;  // ... moves left behind the nearest node that is not lighter
        lst.insert(lst.begin() + (j + 1), nn);
    }
    std::vector<int> depth(n, 0);
    std::vector<std::pair<int, int>> st{{lst[0].id, 0}};
    while (!st.empty()) {
        const auto [x, d] = st.back();
        st.pop_back();
        if (x < n) {
            depth[x] = d;
        } else {
            st.push_back({child[x - n].first, d + 1});
            st.push_back({child[x - n].second, d + 1});
        }
    }
    return depth;
}

// counts[0] = documents (END), counts[c] = occurrences of symbol code c (1 .. sigma).  false: no code within VL_MAX_LEN bits
// (more than 2^VL_MAX_LEN symbols) or a degenerate alphabet.
inline bool vl_build(const uint64_t* counts, int sigma, VlCode& out) {
    const int n = sigma + 1;
    if (sigma < 2 || n > (1 << VL_MAX_LEN)) return false;
    double total = 0;
    for (int i = 0; i < n; ++i) total += (double)counts[i];
    if (total <= 0) return false;
    std::vector<double> w(n);
    for (int i = 0; i < n; ++i) w[i] = std::max((double)counts[i], 0.5) / total;
    std::vector<int> d = vl_gw_depths(w);
    auto deepest = [](const std::vector<int>& v) {
        int m = 0;
        for (int x : v) m = std::max(m, x);
        return m;
    };
    if (deepest(d) > VL_MAX_LEN) {  // the smallest weight floor that keeps the tree within the limit (the depth falls as the floor rises)
        double lo = 0, hi = 1.0;
        for (int it = 0; it < 40; ++it) {
            const double mid = 0.5 * (lo + hi);
            std::vector<double> wf(w);
            for (double& x : wf) x += mid;
            if (deepest(vl_gw_depths(wf)) <= VL_MAX_LEN) hi = mid; else lo = mid;
        }
        std::vector<double> wf(w);
        for (double& x : wf) x += hi;
        d = vl_gw_depths(wf);
        if (deepest(d) > VL_MAX_LEN) return false;
    }
    // code words from the depths, in symbol order: the next word is the previous one plus one, at the new length
    out = VlCode{};
    out.nsyms = n;
    uint32_t c = 0;
    for (int i = 0; i < n; ++i) {
        if (i > 0) {
            c += 1;
            if (d[i] >= d[i - 1]) {
                c <<= (d[i] - d[i - 1]);
            } else {
                if (c & ((1u << (d[i - 1] - d[i])) - 1u)) return false;  // (cannot happen for the depths of a tree)
                c >>= (d[i - 1] - d[i]);
            }
        }
        if (d[i] < 1 || d[i] > VL_MAX_LEN || c >= (1u << d[i])) return false;
        out.len[i] = (uint8_t)d[i];
        out.bits[i] = (uint16_t)c;
    }
    if (out.bits[0] != 0) return false;  // END must be the all-zero word
    out.end_len = d[0];
    out.max_len = deepest(d);
    out.min_len = VL_MAX_LEN;
    double q = 0, avg = 0, tsym = 0;
    for (int i = 1; i < n; ++i) tsym += (double)counts[i];
    for (int i = 1; i < n; ++i) {
        const double p = (double)counts[i] / tsym;
        q += p * p;
        avg += p * d[i];
        out.min_len = std::min(out.min_len, d[i]);
    }
    out.q = q;
    out.avg_len = avg;
    // collision exponent per bit: the x > 1 with sum p_i^2 x^(len_i) = 1 (a colliding pair extends by symbol i with probability
    // p_i^2 and spends len_i bits on it); two suffixes agree on B code bits with probability ~ x^(-B)
    double lo = 1.0, hi = 4.0;
    for (int it = 0; it < 60; ++it) {
        const double mid = 0.5 * (lo + hi);
        double s = 0;
        for (int i = 1; i < n; ++i) {
            const double p = (double)counts[i] / tsym;
            s += p * p * std::pow(mid, (double)d[i]);
        }
        if (s > 1.0) hi = mid; else lo = mid;
    }
    out.rate = std::log2(lo);
    return true;
}

}  // namespace cdb
