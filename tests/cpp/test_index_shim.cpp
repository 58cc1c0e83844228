// Exercises the drop-in index.h / index.cpp the way the reference's database.cpp does
// (make_unique<string_index>, dynamic_cast, add, build through index*, query through index*).
// usage: test_index_shim numeric   — CPU-only parts (no GPU needed)
//        test_index_shim all       — also the GPU string index (README known answers)
//        test_index_shim views [docs] [doclen] [reps] [utf8]  — timing: string_index::build() over separately allocated strings (JSON)
//        test_index_shim cold [bytes] [noreserve]      — timing: a fresh process builds one large UTF-8 column once (JSON); unless
//                                                        "noreserve", string_index::reserve() runs while the strings are made
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <thread>
#include <vector>
#include <memory>
#include <string>

#include "index.h"
#include "highlight.h"
#include "ranking.h"

static int failures = 0;
#define CHECK(c)                                                   \
    do {                                                           \
        if (!(c)) {                                                \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            ++failures;                                            \
        }                                                          \
    } while (0)

using R = std::vector<std::pair<int64_t, int64_t>>;

static void numeric() {
    std::map<std::string, std::unique_ptr<index>> indices;
    indices["n"] = std::make_unique<integer_index>();
    auto* ip = dynamic_cast<integer_index*>(indices["n"].get());
    CHECK(ip != nullptr);
    const int64_t vals[] = {123, 234, 999, 100, 200};
    for (int i = 0; i < 5; ++i) ip->add(10 + i, vals[i]);
    indices["n"]->build();
    bool threw = false;
    CHECK((indices["n"]->query("[100,200]") == R{{13, 0}, {10, 0}, {14, 0}}));
    CHECK((indices["n"]->query("(100,200)") == R{{10, 0}}));
    CHECK((indices["n"]->query(" [ 100, 200) ") == R{{13, 0}, {10, 0}}));
    threw = false;  // blanks BEFORE the comma belong to the value in the reference's regex -> from_chars fails
    try { indices["n"]->query("[100 ,200]"); } catch (const std::runtime_error& e) { threw = std::string(e.what()) == "Invalid value: 100 "; }
    CHECK(threw);
    CHECK((indices["n"]->query("[-inf,inf]").size() == 5));
    threw = false;
    try { indices["n"]->query("100..200"); } catch (const std::runtime_error& e) { threw = std::string(e.what()) == "Invalid range: 100..200"; }
    CHECK(threw);

    double_index di;
    di.add(1, 1.7724); di.add(2, -3.5); di.add(3, 2.0);
    di.build();
    CHECK((di.query("[1.5,2.0]") == R{{1, 0}, {3, 0}}));
    CHECK((di.query("[1.5,2.0)") == R{{1, 0}}));

    bool_index bi;
    bi.add(7, true); bi.add(8, false); bi.add(9, true);
    bi.build();
    CHECK((bi.query("true") == R{{7, 0}, {9, 0}}));
    CHECK((bi.query("false") == R{{8, 0}}));
    threw = false;
    try { bi.query("maybe"); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);

    // interface.cpp:114-146 restated in ranking.h
    {
        cdb_shim::rows_t a{{1, 2}, {3, 1}, {5, 4}, {9, 1}}, b{{3, 2}, {4, 7}, {5, 1}};
        auto m = cdb_shim::and_merge(a, b);
        CHECK((m == cdb_shim::rows_t{{3, 3}, {5, 5}}));
        cdb_shim::rows_t c{{1, 1}, {2, 5}, {3, 2}, {4, 9}};
        cdb_shim::correlation_filter(c, 2, 9);
        CHECK((c == cdb_shim::rows_t{{2, 5}, {3, 2}}));
        cdb_shim::rows_t r{{1, 1}, {2, 5}, {3, 2}, {4, 9}, {5, 3}};
        cdb_shim::rank_by_correlation(r);
        CHECK((r == cdb_shim::rows_t{{4, 9}, {2, 5}, {5, 3}, {3, 2}, {1, 1}}));
    }
    index base;
    threw = false;
    try { base.build(); } catch (const std::logic_error&) { threw = true; }
    CHECK(threw);
    CHECK(string_index::number == 3 && double_index::number == 2 && integer_index::number == 1 && bool_index::number == 0);
}

static void gpu_string() {
    // README.md:80-92 / SURVEY §8c: "010" in {"3010103","301022","01011010"}
    std::map<std::string, std::unique_ptr<index>> indices;
    indices["secret"] = std::make_unique<string_index>();
    auto* sp = dynamic_cast<string_index*>(indices["secret"].get());
    CHECK(sp != nullptr);
    std::string docs[] = {"3010103", "301022", "01011010"};
    for (int i = 0; i < 3; ++i) sp->add(100 + i, docs[i]);
    CHECK(indices["secret"]->query("010").empty());  // before build: nothing (reference: undefined)
    indices["secret"]->build();
    CHECK((indices["secret"]->query("010") == R{{100, 2}, {101, 1}, {102, 2}}));
    CHECK((indices["secret"]->query("3") == R{{100, 2}, {101, 1}}));
    CHECK(indices["secret"]->query("zzz").empty());
    bool threw = false;
    try { indices["secret"]->query(""); } catch (const std::runtime_error& e) { threw = std::string(e.what()) == "Empty keywords are not allowed"; }
    CHECK(threw);
    auto batch = sp->query_batch({"010", "0", "!"});
    CHECK(batch.size() == 3 && batch[0] == (R{{100, 2}, {101, 1}, {102, 2}}) && batch[1] == (R{{100, 3}, {101, 2}, {102, 4}}) && batch[2].empty());
    CHECK((sp->query_any({"010", "3"}) == R{{100, 4}, {101, 2}, {102, 2}}));
    CHECK((sp->query_ranked({"010", "3"}, 1, 1000) == R{{100, 4}, {101, 2}, {102, 2}}));   // descending count, ties by id
    CHECK((sp->query_ranked({"010", "3"}, 1, 3) == R{{101, 2}, {102, 2}}));
    CHECK((sp->query_ranked({"010", "3"}, 1, 1000, 1) == R{{100, 4}}));
    {   // README.md:107-110: "010" highlighted in "3010103" gives "3<b>01010</b>3"
        auto spans = sp->highlight_spans({"010"});
        CHECK(spans.size() == 3 && spans[0].first == 100);
        CHECK(cdb_shim::render_spans(docs[0], spans[0].second, "<b>", "</b>") == "3<b>01010</b>3");
        CHECK(cdb_shim::render_spans(docs[2], spans[2].second, "[", "]") == "[010]11[010]");
    }
    // rebuild-and-swap as database.cpp:170-281 does: new object built while the old one still answers
    auto fresh = std::make_unique<string_index>();
    fresh->add(1, "hello world");
    fresh->build();
    CHECK((indices["secret"]->query("0").size() == 3));
    indices["secret"] = std::move(fresh);
    CHECK((indices["secret"]->query("o") == R{{1, 2}}));
}

// ---- what the reference's caller pays (bench.py: pcie_inclusive.build_views, cold_start) -------------------------------
// database.cpp:262-264 keeps every value in its own std::string and hands string_index::add a VIEW of it; build() then has
// to gather a million scattered strings (cdb_build_views).  `views` times exactly that at a given shape; `cold` is
// server.cpp:44 — a fresh process (no torch, no warm block cache) building one large column from host memory once.
static inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// valid UTF-8: 50 % ASCII, 30 % two-byte, 20 % three-byte code points (SURVEY §8d's C4 mix), cut at code-point boundaries
static void fill_utf8(std::string& s, size_t want, uint64_t seed) {
    s.clear();
    s.reserve(want + 4);
    uint64_t k = seed * 0x100000001B3ull;
    while (s.size() < want) {
        const uint64_t r = mix64(k++);
        const uint32_t kind = (uint32_t)(r % 10);
        if (kind < 5) {
            s.push_back((char)(0x20 + (r >> 8) % 95));
        } else if (kind < 8) {
            const uint32_t cp = 0x80 + (uint32_t)((r >> 8) % (0x800 - 0x80));
            s.push_back((char)(0xC0 | (cp >> 6)));
            s.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            uint32_t cp = 0x800 + (uint32_t)((r >> 8) % (0x10000 - 0x800));
            if (cp >= 0xD800 && cp < 0xE000) cp -= 0x800;  // (no surrogates)
            s.push_back((char)(0xE0 | (cp >> 12)));
            s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
}
// a keyword of `want` ASCII bytes that occurs in `v` (UTF-8 text: the first run of that many bytes below 0x80)
static std::string ascii_keyword(const std::string& v, size_t want) {
    size_t run = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        run = ((unsigned char)v[i] < 0x80) ? run + 1 : 0;
        if (run == want) return v.substr(i + 1 - want, want);
    }
    return v.substr(0, std::min(want, v.size()));
}
static int bench_views(size_t docs, size_t doclen, int reps, bool utf8 = false) {
    // one heap block per value, like the map<string, var> values of database.cpp:22 (allocated in shuffled order so that
    // neighbouring documents are not neighbours in memory)
    std::vector<std::string> values(docs);
    std::vector<uint32_t> order(docs);
    for (size_t i = 0; i < docs; ++i) order[i] = (uint32_t)i;
    for (size_t i = docs; i > 1; --i) std::swap(order[i - 1], order[mix64(i) % i]);
    const unsigned T = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            for (size_t k = t; k < docs; k += T) {
                std::string& v = values[order[k]];
                if (utf8) {
                    fill_utf8(v, doclen, order[k]);
                    continue;
                }
                v.resize(doclen);
                uint64_t x = mix64(order[k]);
                for (size_t b = 0; b < doclen; ++b) {
                    if ((b & 7) == 0) x = mix64(x);
                    v[b] = (char)(0x20 + ((x >> (8 * (b & 7))) & 0xFF) % 95);
                }
            }
        });
    for (auto& x : th) x.join();
    std::vector<double> add_ms, build_ms;
    size_t hits = 0;
    for (int r = 0; r < reps + 1; ++r) {   // fresh object per build, as database.cpp:255 does; repetition 0 is the cold one
        auto ix = std::make_unique<string_index>();
        double t = now_ms();
        for (size_t i = 0; i < docs; ++i) ix->add((int64_t)i, values[i]);
        add_ms.push_back(now_ms() - t);
        t = now_ms();
        static_cast<index*>(ix.get())->build();
        build_ms.push_back(now_ms() - t);
        // (on text with bytes >= 0x80 the reference's own bisection misses keywords that occur — index.h:66-73, SURVEY Q2 —,
        //  and the default reference_compat reproduces that: ask until a keyword answers)
        for (size_t k = 0; k < 64 && (k == 0 || (utf8 && !hits)); ++k)
            hits += static_cast<index*>(ix.get())->query(ascii_keyword(values[(docs / 2 + 7919 * k) % docs], utf8 ? 3 : 8)).size();
    }
    size_t total = 0;
    for (auto& v : values) total += v.size();
    // (UTF-8 text: zero rows IS the reference's answer — its bisection compares unsigned bytes on an array whose root radix node
    //  lays the bytes 0x80..0xFF out first, index.h:66-73, so it walks away from every ASCII keyword; reference_compat = 1, the
    //  default, reproduces that bit for bit)
    if (!hits && !utf8) { std::printf("{\"error\": \"the built index did not answer\"}\n"); return 1; }
    std::printf("{\"docs\": %zu, \"doclen\": %zu, \"bytes\": %zu, \"utf8\": %s, \"first_build_ms\": %.2f, \"build_ms\": [", docs, doclen, total, utf8 ? "true" : "false", build_ms[0]);
    double best = 1e30;
    for (int r = 1; r <= reps; ++r) {
        std::printf("%s%.2f", r > 1 ? ", " : "", build_ms[r]);
        best = std::min(best, build_ms[r]);
    }
    std::printf("], \"add_ms\": %.2f, \"build_views_GiB_per_s\": %.3f, \"rows_of_the_probe_keywords\": %zu, \"note\": \"string_index::add of %zu separately allocated std::strings + "
                "string_index::build() = cdb_build_views (gather into pinned chunks + upload + device build), C++ caller, fresh object per build\"}\n",
                add_ms.back(), (double)total / (1ull << 30) / (best * 1e-3), hits, docs);
    return 0;
}
static int bench_cold(size_t bytes, bool reserve) {
    const size_t doclen = 1024, docs = bytes / doclen;
    std::vector<std::string> values(docs);
    const unsigned T = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    double reserve_call_ms = 0;
    if (reserve) {
        // what a CoffeeDB start-up does with the shim: the size of the raw directory is known before init() reads it
        // (server.cpp:43-44), so the first build's device memory is mapped on a helper thread WHILE the data is loaded
        // (here: while the strings are generated)
        std::string sample;
        fill_utf8(sample, 4096, 12345);
        const double tr = now_ms();
        string_index::reserve(bytes, sample);
        reserve_call_ms = now_ms() - tr;
    }
    const double tg = now_ms();
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([&, t] { for (size_t k = t; k < docs; k += T) fill_utf8(values[k], doclen, k); });
    for (auto& x : th) x.join();
    const double gen_ms = now_ms() - tg;
    size_t total = 0;
    for (auto& v : values) total += v.size();
    const double t0 = now_ms();
    auto ix = std::make_unique<string_index>();   // (device + stream creation: part of a cold start)
    const double t1 = now_ms();
    for (size_t i = 0; i < docs; ++i) ix->add((int64_t)i, values[i]);
    const double t2 = now_ms();
    static_cast<index*>(ix.get())->build();
    const double t3 = now_ms();
    size_t rows = 0, asked = 0;  // (reference_compat reproduces the reference's misses on bytes >= 0x80: ask until a keyword answers)
    for (; asked < 64 && !rows; ++asked) rows = static_cast<index*>(ix.get())->query(ascii_keyword(values[(docs / 2 + 7919 * asked) % docs], 3)).size();
    const double t4 = now_ms();
    // database.cpp:276-280 rebuilds when it is told to — seconds to hours after start-up.  Behind build() the library proves the
    // array's order and maps the memory the NEXT generation will ask for (helper thread); a rebuild that arrives before that is
    // through simply waits for it.  Here: wait first, and report how long that was.
    const bool proved = ix->settle();
    const double t4b = now_ms();
    auto again = std::make_unique<string_index>();
    for (size_t i = 0; i < docs; ++i) again->add((int64_t)i, values[i]);
    const double t5 = now_ms();
    static_cast<index*>(again.get())->build();
    const double t6 = now_ms();
    std::printf("{\"bytes\": %zu, \"docs\": %zu, \"reserve\": %s, \"reserve_call_ms\": %.2f, \"generate_ms\": %.0f, \"create_ms\": %.1f, \"add_ms\": %.1f, \"first_build_ms\": %.1f, "
                "\"first_query_ms\": %.2f, \"first_query_rows\": %zu, \"first_query_keywords_asked\": %zu, \"background_ms\": %.1f, \"order_proved\": %s, \"second_build_ms\": %.1f, \"cold_over_warm\": %.2f, "
                "\"first_build_GiB_per_s\": %.3f, \"note\": \"fresh process, no torch, no warm block cache: string_index over %zu separately "
                "allocated strings of valid UTF-8, build() = gather + upload + device build incl. every first-use allocation (server.cpp:44); "
                "background_ms = waiting for the order proof and the pre-mapping of the next generation's memory behind the first build (helper thread); "
                "second_build = a new object beside the first (database.cpp:276-280) after that; first_query_rows = 0 is the "
                "reference's own answer on text with bytes >= 0x80 (its bisection misses ASCII keywords there, SURVEY Q2; reproduced by the default "
                "reference_compat = 1)\"}\n",
                total, docs, reserve ? "true" : "false", reserve_call_ms, gen_ms, t1 - t0, t2 - t1, t3 - t2, t4 - t3, rows, asked, t4b - t4, proved ? "true" : "false", t6 - t5, (t3 - t2) / (t6 - t5),
                (double)total / (1ull << 30) / ((t3 - t2) * 1e-3), docs);
    (void)rows;  // (under reference_compat a UTF-8 keyword may find nothing: the reference's own behaviour on bytes >= 0x80)
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "views")
        return bench_views(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : (1u << 20), argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1024,
                           argc > 4 ? std::atoi(argv[4]) : 3, argc > 5 && std::string(argv[5]) == "utf8");
    if (argc > 1 && std::string(argv[1]) == "cold")
        return bench_cold(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : (4ull << 30), !(argc > 3 && std::string(argv[3]) == "noreserve"));
    const bool all = argc > 1 && std::string(argv[1]) == "all";
    numeric();
    if (all) gpu_string();
    std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
