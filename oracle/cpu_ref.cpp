// oracle/cpu_ref.cpp — CPU restatement of CoffeeDB's string index (TEST INFRASTRUCTURE ONLY).
//
// This file is the parity ORACLE for the MI355X text-index path.  It is NOT part of the product:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The shipped
// library (coffeedb_amd/csrc) never links, loads or calls anything in oracle/.
//
// Pinning status: PINNED against the reference's own known answers —
//   * README.md:80-92   query "010" over {"3010103","301022"} -> $correlation 2 and 1
//   * test/test-string.py:14-19,52-56  overlapping brute-force count per document (property oracle)
//   * SURVEY.md §8c golden vectors recorded from the unmodified reference (full SA of the README
//     corpus, empty index, empty doc + duplicate id) — committed under tests/golden/.
//   * tests/golden/model_cases.json — canonical arrays (SHA-256) and keyword rows for radix nodes over several
//     levels, ragged documents, bytes >= 0x80 and the u32 / u64 width boundary, from an independent pure-Python
//     restatement of index.cpp (tests/ref_model.py); not outputs of the reference binary.
// The reference itself cannot be compiled in this image without a stand-in <format> header
// (progress_bar.h:10 includes <format>; g++ 11.4 / ROCm clang 22 do not ship it), so no oracle/_ref
// build exists; see DESIGN.md "Oracle".
//
// What is restated (reference file:line, relative to /root/reference/src):
//   entry encoding (off << bits) | doc, width rule, limits      index.cpp:178-215
//   work list + worker threads                                   index.cpp:19-62, 216-231
//   MSD 257-way American-flag radix node, signed-char symbols    index.cpp:96-126, index.h:66-73
//   comparison-sorted leaves (unsigned lexicographic)            index.cpp:86-95,  index.h:61-65
//   query: lower bound / prefix upper bound / gather / sort / RLE index.cpp:237-326
//
// Data layout differs from the reference on purpose (this is a restatement, not a copy): documents
// are held as one concatenated byte buffer plus a doc_start[] offset table instead of a vector of
// string_views; a suffix is therefore (doc_start[doc] + off, doc_start[doc+1]).
#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Task {
    uint64_t lo, hi;   // half-open range of SA slots
    uint64_t depth;    // number of leading symbols already known equal
};

struct Oracle {
    std::vector<int64_t> ids;
    std::vector<uint64_t> doc_start{0};
    std::string text;
    uint64_t bits = 1, mask = 1, size = 0;
    int width = 0;  // 4 or 8 once built, 0 before
    std::vector<uint32_t> sa32;
    std::vector<uint64_t> sa64;
    std::string err;
    bool built = false;
};

template <typename T>
struct View {
    const Oracle& ix;
    T* sa;
    inline uint64_t doc_of(T e) const { return (uint64_t)e & ix.mask; }
    inline uint64_t off_of(T e) const { return (uint64_t)e >> ix.bits; }
    // index.h:61-65 — suffix as (pointer, length); offsets never exceed the doc length on this path.
    inline void suffix(T e, uint64_t depth, const unsigned char*& p, uint64_t& len) const {
        const uint64_t d = doc_of(e);
        const uint64_t b = ix.doc_start[d] + off_of(e) + depth;
        const uint64_t end = ix.doc_start[d + 1];
        p = (const unsigned char*)ix.text.data() + b;
        len = end - b;
    }
    // index.h:66-73 — 0 at end of document, otherwise (int)char - CHAR_MIN + 1 with SIGNED char
    // (x86-64), i.e. bytes 0x80..0xFF map to 1..128 and 0x00..0x7F to 129..256.
    inline int symbol(T e, uint64_t depth) const {
        const uint64_t d = doc_of(e);
        const uint64_t b = ix.doc_start[d] + off_of(e) + depth;
        if (b == ix.doc_start[d + 1]) return 0;
        return (int)(signed char)ix.text[b] + 128 + 1;
    }
    // unsigned lexicographic "a < b", shorter-is-smaller on a common prefix (std::string_view <).
    inline bool less(T a, T b, uint64_t depth) const {
        const unsigned char *pa, *pb;
        uint64_t la, lb;
        suffix(a, depth, pa, la);
        suffix(b, depth, pb, lb);
        const int c = std::memcmp(pa, pb, la < lb ? la : lb);
        return c != 0 ? c < 0 : la < lb;
    }
};

// Shared work list (semantics of index.cpp:19-62: FIFO of (range, depth), workers run until every
// suffix has been retired).  A mutex-protected deque replaces the fixed 1e6-slot array.
struct WorkList {
    std::mutex mu;
    std::deque<Task> q;
    std::atomic<int64_t> queued{0};  // lets idle workers poll without taking the lock (the
                                     // reference's idle workers only read two atomics, index.cpp:36-40)
    std::atomic<uint64_t> remaining{0};
    void push(Task t) {
        std::lock_guard<std::mutex> g(mu);
        q.push_back(t);
        queued.fetch_add(1, std::memory_order_release);
    }
    bool pop(Task& t) {
        if (queued.load(std::memory_order_acquire) <= 0) return false;
        std::lock_guard<std::mutex> g(mu);
        if (q.empty()) return false;
        t = q.front();
        q.pop_front();
        queued.fetch_sub(1, std::memory_order_release);
        return true;
    }
};

template <typename T>
void worker(const View<T> v, WorkList& wl, uint64_t leaf_max) {
    uint64_t bucket_end[264], fill[264];
    while (wl.remaining.load(std::memory_order_acquire) != 0) {
        Task t;
        if (!wl.pop(t)) {
            std::this_thread::yield();
            continue;
        }
        T* a = v.sa + t.lo;
        const uint64_t len = t.hi - t.lo;
        if (len <= leaf_max) {
            // leaf (index.cpp:86-95): comparison sort on the remaining suffix text
            wl.remaining.fetch_sub(len, std::memory_order_acq_rel);
            const uint64_t depth = t.depth;
            std::sort(a, a + len, [&v, depth](T x, T y) { return v.less(x, y, depth); });
            continue;
        }
        // radix node (index.cpp:96-126)
        std::fill(bucket_end, bucket_end + 264, 0);
        for (uint64_t i = 0; i < len; ++i) bucket_end[v.symbol(a[i], t.depth)] += 1;
        for (int c = 1; c < 264; ++c) bucket_end[c] += bucket_end[c - 1];
        std::copy(bucket_end, bucket_end + 264, fill);
        int cur = 0;
        for (uint64_t i = 0; i < len; ++i) {
            while (i == bucket_end[cur]) ++cur;       // slot i belongs to bucket `cur`
            for (;;) {                                // cycle until slot i holds a `cur` element
                const int c = v.symbol(a[i], t.depth);
                if (c == cur) break;
                fill[c] -= 1;                         // buckets are filled from the back
                std::swap(a[i], a[fill[c]]);
            }
        }
        wl.remaining.fetch_sub(bucket_end[0], std::memory_order_acq_rel);  // end-of-doc bucket is final
        for (int c = 1; c < 264; ++c) {
            if (bucket_end[c] > bucket_end[c - 1])
                wl.push(Task{t.lo + bucket_end[c - 1], t.lo + bucket_end[c], t.depth + 1});
        }
    }
}

template <typename T>
void build_typed(Oracle& ix, T* sa, unsigned nthreads) {
    // fill doc-major, offset-minor (index.cpp:209-215)
    T* p = sa;
    const uint64_t ndocs = ix.ids.size();
    for (uint64_t d = 0; d < ndocs; ++d) {
        const uint64_t len = ix.doc_start[d + 1] - ix.doc_start[d];
        for (uint64_t j = 0; j < len; ++j) *p++ = (T)((j << ix.bits) | d);
    }
    if (ix.size == 0) return;
    WorkList wl;
    wl.remaining.store(ix.size);
    wl.push(Task{0, ix.size, 0});
    const uint64_t leaf_max = std::max<uint64_t>(4096, ix.size / 256);  // index.cpp:218
    View<T> v{ix, sa};
    std::vector<std::thread> pool;
    for (unsigned i = 0; i + 1 < nthreads; ++i) pool.emplace_back([&] { worker<T>(v, wl, leaf_max); });
    worker<T>(v, wl, leaf_max);
    for (auto& th : pool) th.join();
}

int popcount64(uint64_t x) { return __builtin_popcountll(x); }

template <typename T>
void query_typed(const Oracle& ix, const T* sa, const char* kw, uint64_t m,
                 std::vector<int64_t>& out_ids, std::vector<int64_t>& out_cnt) {
    View<T> v{ix, const_cast<T*>(sa)};
    const unsigned char* k = (const unsigned char*)kw;
    // lower bound (index.cpp:260-274): smallest M in [0, size-1] with keyword <= suffix(sa[M])
    int64_t L = 0, R = (int64_t)ix.size - 1;
    while (L < R) {
        const int64_t M = L + (R - L) / 2;
        const unsigned char* s;
        uint64_t sl;
        v.suffix(sa[M], 0, s, sl);
        const int c = std::memcmp(k, s, m < sl ? m : sl);
        const bool kw_le = c != 0 ? c < 0 : m <= sl;
        if (kw_le) R = M; else L = M + 1;
    }
    const int64_t left = L;
    // upper bound (index.cpp:275-287): largest M with suffix(sa[M]) starting with keyword
    L = left - 1;
    R = (int64_t)ix.size - 1;
    while (L < R) {
        const int64_t M = L + (R - L + 1) / 2;
        const unsigned char* s;
        uint64_t sl;
        v.suffix(sa[M], 0, s, sl);
        const bool pref = sl >= m && std::memcmp(k, s, m) == 0;
        if (pref) L = M; else R = M - 1;
    }
    const int64_t right = L + 1;
    if (left >= right) return;
    // gather + sort doc indices (index.cpp:288-315)
    std::vector<uint64_t> docs;
    docs.reserve(right - left + 1);
    for (int64_t i = left; i < right; ++i) docs.push_back((uint64_t)sa[i] & ix.mask);
    const uint64_t radix_mask = (1u << 17) - 1;
    const uint64_t h = docs.size();
    if (h < radix_mask) {
        std::sort(docs.begin(), docs.end());
    } else {
        // two stable counting passes on 17-bit digits at shifts 0 and 16 (index.cpp:299-314)
        std::vector<uint64_t> tmp(h);
        for (int pass = 0; pass < 2; ++pass) {
            const int sh = pass == 0 ? 0 : 16;
            std::vector<uint64_t> cnt(radix_mask + 8, 0);
            std::vector<uint64_t>& src = pass == 0 ? docs : tmp;
            std::vector<uint64_t>& dst = pass == 0 ? tmp : docs;
            for (uint64_t j = 0; j < h; ++j) cnt[(src[j] >> sh) & radix_mask]++;
            for (uint64_t j = 1; j <= radix_mask; ++j) cnt[j] += cnt[j - 1];
            for (int64_t j = (int64_t)h - 1; j >= 0; --j) dst[--cnt[(src[j] >> sh) & radix_mask]] = src[j];
        }
    }
    // run-length -> (ids[doc], count), ascending doc index (index.cpp:316-322)
    docs.push_back(~0ull);
    for (uint64_t last = 0, i = 1; i < docs.size(); ++i) {
        if (docs[i] != docs[i - 1]) {
            out_ids.push_back(ix.ids[docs[last]]);
            out_cnt.push_back((int64_t)(i - last));
            last = i;
        }
    }
}

template <typename T>
uint64_t canonicalize_typed(Oracle& ix, T* sa) {
    // sort every maximal run of equal suffixes by doc index (entry & mask).
    View<T> v{ix, sa};
    uint64_t runs = 0, i = 0;
    while (i < ix.size) {
        uint64_t j = i + 1;
        while (j < ix.size && !v.less(sa[i], sa[j], 0) && !v.less(sa[j], sa[i], 0)) ++j;
        if (j - i > 1) {
            ++runs;
            std::sort(sa + i, sa + j, [&](T x, T y) { return ((uint64_t)x & ix.mask) < ((uint64_t)y & ix.mask); });
        }
        i = j;
    }
    return runs;
}

// the same over [i0, i1) of the array on `nthreads` threads: a thread owns the runs that START in its chunk (a run is found by
// comparing neighbours, so a thread first skips the entries that continue a run of the chunk in front) — what bench.py's
// full-size bit-exact check uses on 2^30 entries, where the single-threaded walk takes minutes
template <typename T>
uint64_t canonicalize_mt_typed(Oracle& ix, T* sa, unsigned nthreads) {
    if (nthreads <= 1 || ix.size < (1u << 20)) return canonicalize_typed(ix, sa);
    // phase 1 (read-only): every thread lists the runs of two and more that start in its chunk; phase 2: the runs are sorted
    // (they are disjoint, so nobody reads what another thread writes)
    std::vector<std::vector<std::pair<uint64_t, uint64_t>>> found(nthreads);
    const uint64_t n = ix.size;
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthreads; ++t)
            th.emplace_back([&, t] {
                View<T> v{ix, sa};
                auto same = [&](uint64_t a, uint64_t b) { return !v.less(sa[a], sa[b], 0) && !v.less(sa[b], sa[a], 0); };
                uint64_t i = n * t / nthreads;
                const uint64_t end = n * (t + 1) / nthreads;
                while (i > 0 && i < end && same(i - 1, i)) ++i;
                while (i < end) {
                    uint64_t j = i + 1;
                    while (j < n && same(i, j)) ++j;
                    if (j - i > 1) found[t].push_back({i, j});
                    i = j;
                }
            });
        for (auto& x : th) x.join();
    }
    uint64_t total = 0;
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthreads; ++t)
            th.emplace_back([&, t] {
                for (auto& r : found[t])
                    std::sort(sa + r.first, sa + r.second, [&](T x, T y) { return ((uint64_t)x & ix.mask) < ((uint64_t)y & ix.mask); });
            });
        for (auto& x : th) x.join();
    }
    for (auto& f : found) total += f.size();
    return total;
}

template <typename T>
uint64_t inversions_typed(const Oracle& ix, const T* sa) {
    View<T> v{ix, const_cast<T*>(sa)};
    uint64_t bad = 0;
    for (uint64_t i = 1; i < ix.size; ++i) bad += v.less(sa[i], sa[i - 1], 0) ? 1 : 0;
    return bad;
}

}  // namespace

extern "C" {

void* orc_create() { return new Oracle(); }
void orc_destroy(void* h) { delete (Oracle*)h; }
const char* orc_last_error(void* h) { return ((Oracle*)h)->err.c_str(); }

// string_index::add (index.cpp:174-177)
void orc_add(void* h, int64_t id, const char* p, uint64_t len) {
    Oracle& ix = *(Oracle*)h;
    ix.ids.push_back(id);
    ix.text.append(p, len);
    ix.doc_start.push_back(ix.text.size());
}

// bulk add: ndocs documents given as one blob + doc_start[ndocs+1]
void orc_add_bulk(void* h, const int64_t* ids, const char* blob, const uint64_t* doc_start, uint64_t ndocs) {
    Oracle& ix = *(Oracle*)h;
    const uint64_t base = ix.text.size();
    ix.text.append(blob + doc_start[0], doc_start[ndocs] - doc_start[0]);
    for (uint64_t d = 0; d < ndocs; ++d) {
        ix.ids.push_back(ids[d]);
        ix.doc_start.push_back(base + doc_start[d + 1] - doc_start[0]);
    }
}

// string_index::build (index.cpp:178-236).  nthreads == 0 -> hardware_concurrency (index.cpp:225).
int orc_build(void* h, unsigned nthreads) {
    Oracle& ix = *(Oracle*)h;
    if (nthreads == 0) nthreads = std::max(1u, std::thread::hardware_concurrency());
    const uint64_t ndocs = ix.ids.size();
    ix.size = 0;
    uint64_t mask1 = 1, mask2 = 1;
    while (mask1 < ndocs) mask1 = (mask1 << 1) + 1;
    for (uint64_t d = 0; d < ndocs; ++d) {
        const uint64_t len = ix.doc_start[d + 1] - ix.doc_start[d];
        ix.size += len;
        while (mask2 < len) mask2 = (mask2 << 1) + 1;
    }
    const int bits1 = popcount64(mask1), bits2 = popcount64(mask2);
    if (bits1 + bits2 > 64) {
        ix.err = "The amount of data exceeds the maximum range that CoffeeDB can handle";
        return 1;
    }
    if (bits1 > 32) {
        ix.err = "The number of objects exceeds the maximum range that CoffeeDB can handle";
        return 1;
    }
    ix.mask = mask1;
    ix.bits = (uint64_t)bits1;
    ix.sa32.clear();
    ix.sa64.clear();
    if (bits1 + bits2 <= 32) {
        ix.width = 4;
        ix.sa32.resize(ix.size);
        build_typed<uint32_t>(ix, ix.sa32.data(), nthreads);
    } else {
        ix.width = 8;
        ix.sa64.resize(ix.size);
        build_typed<uint64_t>(ix, ix.sa64.data(), nthreads);
    }
    ix.built = true;
    return 0;
}

uint64_t orc_size(void* h) { return ((Oracle*)h)->size; }
uint64_t orc_bits(void* h) { return ((Oracle*)h)->bits; }
uint64_t orc_mask(void* h) { return ((Oracle*)h)->mask; }
int orc_sa_width(void* h) { return ((Oracle*)h)->width; }
const void* orc_sa_data(void* h) {
    Oracle& ix = *(Oracle*)h;
    return ix.width == 4 ? (const void*)ix.sa32.data() : (const void*)ix.sa64.data();
}

// string_index::query (index.cpp:237-326).  Returns 0 ok, 1 error (empty keyword).
// Results are written to caller-owned arrays of capacity `cap`; *nrows receives the row count
// (if it exceeds cap the arrays hold the first cap rows).
int orc_query(void* h, const char* kw, uint64_t m, int64_t* ids, int64_t* counts, uint64_t cap, uint64_t* nrows) {
    Oracle& ix = *(Oracle*)h;
    if (m == 0) {
        ix.err = "Empty keywords are not allowed";
        return 1;
    }
    std::vector<int64_t> oi, oc;
    if (ix.built) {
        if (ix.width == 4) query_typed<uint32_t>(ix, ix.sa32.data(), kw, m, oi, oc);
        else query_typed<uint64_t>(ix, ix.sa64.data(), kw, m, oi, oc);
    }
    *nrows = oi.size();
    const uint64_t k = std::min<uint64_t>(cap, oi.size());
    if (k) {
        std::memcpy(ids, oi.data(), k * 8);
        std::memcpy(counts, oc.data(), k * 8);
    }
    return 0;
}

// Batch loop used by tests (CSR output) and by bench.py's cpu_baseline leg (nthreads workers, each
// calling the single-pattern query — the reference serves concurrent requests from a thread pool,
// package/httplib.h:97-101).  row_ptr has npat+1 entries; ids/counts may be null to only count.
int orc_query_batch(void* h, const char* blob, const uint64_t* offs, uint64_t npat, unsigned nthreads,
                    uint64_t* row_ptr, int64_t* ids, int64_t* counts, uint64_t cap, uint64_t* total_hits) {
    Oracle& ix = *(Oracle*)h;
    if (nthreads == 0) nthreads = 1;
    std::vector<std::vector<int64_t>> ri(npat), rc(npat);
    std::atomic<uint64_t> next{0};
    std::atomic<int> bad{0};
    auto run = [&] {
        for (;;) {
            const uint64_t j = next.fetch_add(1);
            if (j >= npat) break;
            const uint64_t m = offs[j + 1] - offs[j];
            if (m == 0) { bad.store(1); continue; }
            if (!ix.built) continue;
            if (ix.width == 4) query_typed<uint32_t>(ix, ix.sa32.data(), blob + offs[j], m, ri[j], rc[j]);
            else query_typed<uint64_t>(ix, ix.sa64.data(), blob + offs[j], m, ri[j], rc[j]);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(run);
    run();
    for (auto& th : pool) th.join();
    if (bad.load()) {
        ix.err = "Empty keywords are not allowed";
        return 1;
    }
    uint64_t rows = 0, hits = 0;
    for (uint64_t j = 0; j < npat; ++j) {
        if (row_ptr) row_ptr[j] = rows;
        for (size_t r = 0; r < ri[j].size(); ++r) {
            if (ids && rows < cap) { ids[rows] = ri[j][r]; counts[rows] = rc[j][r]; }
            hits += (uint64_t)rc[j][r];
            ++rows;
        }
    }
    if (row_ptr) row_ptr[npat] = rows;
    if (total_hits) *total_hits = hits;
    return 0;
}

// interface.cpp:78-113 — the per-key OR over a list of keywords: query each, sort the (id, count) pairs,
// merge lists by id summing the counts.  Output ascends by id.  Returns the row count (first `cap`
// rows are written).
uint64_t orc_filter_or(void* h, const char* blob, const uint64_t* offs, uint64_t nkw, int64_t* ids, int64_t* counts,
                       uint64_t cap) {
    Oracle& ix = *(Oracle*)h;
    using Row = std::pair<int64_t, int64_t>;
    std::vector<Row> result;
    for (uint64_t k = 0; k < nkw; ++k) {
        std::vector<int64_t> qi, qc;
        const uint64_t m = offs[k + 1] - offs[k];
        if (ix.built && m) {
            if (ix.width == 4) query_typed<uint32_t>(ix, ix.sa32.data(), blob + offs[k], m, qi, qc);
            else query_typed<uint64_t>(ix, ix.sa64.data(), blob + offs[k], m, qi, qc);
        }
        std::vector<Row> now(qi.size());
        for (size_t r = 0; r < qi.size(); ++r) now[r] = Row(qi[r], qc[r]);
        std::sort(now.begin(), now.end());
        if (k == 0) {
            result.swap(now);
            continue;
        }
        std::vector<Row> merged;
        size_t i = 0, j = 0;
        while (i < now.size() && j < result.size()) {
            if (now[i].first == result[j].first) {
                merged.push_back(Row(now[i].first, now[i].second + result[j].second));
                ++i;
                ++j;
            } else if (now[i] < result[j]) {
                merged.push_back(now[i++]);
            } else {
                merged.push_back(result[j++]);
            }
        }
        while (i < now.size()) merged.push_back(now[i++]);
        while (j < result.size()) merged.push_back(result[j++]);
        result.swap(merged);
    }
    for (uint64_t r = 0; r < result.size() && r < cap; ++r) {
        ids[r] = result[r].first;
        counts[r] = result[r].second;
    }
    return result.size();
}

// database.cpp:26-138 — multi-keyword automaton (dense goto table, failure links, longest keyword
// ending at each state) and the span merging of ac_automaton::render (database.cpp:58-76), restated.
// Writes the spans of every document that has at least one; returns the span count.
// doc_of_span / begin / end (inclusive) have capacity cap.
uint64_t orc_highlight_spans(void* h, const char* blob, const uint64_t* offs, uint64_t nkw, uint64_t* doc_of_span,
                             uint64_t* begin, uint64_t* end, uint64_t cap) {
    Oracle& ix = *(Oracle*)h;
    std::vector<std::array<int32_t, 256>> go(1);
    std::vector<int32_t> fail(1, 0), longest(1, 0);
    go[0].fill(0);
    for (uint64_t k = 0; k < nkw; ++k) {  // trie insertion
        int32_t u = 0;
        for (uint64_t i = offs[k]; i < offs[k + 1]; ++i) {
            const unsigned c = (unsigned char)blob[i];
            if (!go[u][c]) {
                go[u][c] = (int32_t)go.size();
                go.emplace_back();
                go.back().fill(0);
                fail.push_back(0);
                longest.push_back(0);
            }
            u = go[u][c];
        }
        longest[u] = (int32_t)(offs[k + 1] - offs[k]);
    }
    std::deque<int32_t> bfs;  // failure links, breadth first
    for (int c = 0; c < 256; ++c)
        if (go[0][c]) bfs.push_back(go[0][c]);
    while (!bfs.empty()) {
        const int32_t r = bfs.front();
        bfs.pop_front();
        for (int c = 0; c < 256; ++c) {
            const int32_t u = go[r][c];
            if (!u) {
                go[r][c] = go[fail[r]][c];
                continue;
            }
            bfs.push_back(u);
            int32_t v = fail[r];
            while (v && !go[v][c]) v = fail[v];
            fail[u] = go[v][c];
            longest[u] = std::max(longest[u], longest[fail[u]]);
        }
    }
    uint64_t total = 0;
    const uint64_t ndocs = ix.ids.size();
    for (uint64_t d = 0; d < ndocs; ++d) {
        const unsigned char* t = (const unsigned char*)ix.text.data() + ix.doc_start[d];
        const uint64_t len = ix.doc_start[d + 1] - ix.doc_start[d];
        std::vector<std::pair<uint64_t, uint64_t>> spans;
        int32_t node = 0;
        for (uint64_t i = 0; i < len; ++i) {
            node = go[node][t[i]];
            if (longest[node]) {
                const uint64_t b = i - (uint64_t)longest[node] + 1;
                while (!spans.empty() && b <= spans.back().first) spans.pop_back();
                if (!spans.empty() && b <= spans.back().second) spans.back().second = i;
                else spans.emplace_back(b, i);
            }
        }
        for (auto& sp : spans) {
            if (total < cap) {
                doc_of_span[total] = d;
                begin[total] = sp.first;
                end[total] = sp.second;
            }
            ++total;
        }
    }
    return total;
}

// SURVEY.md §8c tie canonicalisation; returns the number of runs that were reordered.
uint64_t orc_canonicalize(void* h) {
    Oracle& ix = *(Oracle*)h;
    if (!ix.built) return 0;
    return ix.width == 4 ? canonicalize_typed<uint32_t>(ix, ix.sa32.data())
                         : canonicalize_typed<uint64_t>(ix, ix.sa64.data());
}

uint64_t orc_canonicalize_mt(void* h, unsigned nthreads) {
    Oracle& ix = *(Oracle*)h;
    if (!ix.built) return 0;
    return ix.width == 4 ? canonicalize_mt_typed<uint32_t>(ix, ix.sa32.data(), nthreads)
                         : canonicalize_mt_typed<uint64_t>(ix, ix.sa64.data(), nthreads);
}

// number of adjacent pairs out of unsigned-lexicographic order (0 for pure-ASCII text; >0 exposes
// the signed/unsigned quirk Q2 for bytes >= 0x80).
uint64_t orc_inversions(void* h) {
    Oracle& ix = *(Oracle*)h;
    if (!ix.built) return 0;
    return ix.width == 4 ? inversions_typed<uint32_t>(ix, ix.sa32.data())
                         : inversions_typed<uint64_t>(ix, ix.sa64.data());
}

// test/test-string.py:14-19 — overlapping brute-force occurrence count of kw in every document.
void orc_brute_count(const char* blob, const uint64_t* doc_start, uint64_t ndocs, const char* kw, uint64_t m,
                     int64_t* counts) {
    for (uint64_t d = 0; d < ndocs; ++d) {
        const char* s = blob + doc_start[d];
        const uint64_t len = doc_start[d + 1] - doc_start[d];
        int64_t c = 0;
        if (m != 0 && len >= m)
            for (uint64_t i = 0; i + m <= len; ++i) c += std::memcmp(s + i, kw, m) == 0;
        counts[d] = c;
    }
}

}  // extern "C"
