// Exercises the drop-in index.h / index.cpp the way the reference's database.cpp does
// (make_unique<string_index>, dynamic_cast, add, build through index*, query through index*).
// usage: test_index_shim numeric   — CPU-only parts (no GPU needed)
//        test_index_shim all       — also the GPU string index (README known answers)
#include <cstdio>
#include <map>
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

int main(int argc, char** argv) {
    const bool all = argc > 1 && std::string(argv[1]) == "all";
    numeric();
    if (all) gpu_string();
    std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
