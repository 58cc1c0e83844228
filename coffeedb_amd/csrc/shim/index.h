// index.h — drop-in replacement for CoffeeDB's src/index.h (reference /root/reference/src/index.h:1-87).
//
// Same five classes, same public members, same `number` type tags (index.h:29,37,47,76 — they are the
// on-disk type bytes read by database.cpp:200-245), so the reference's database.cpp compiles and runs
// unchanged against this header:
//     std::make_unique<string_index>()              database.cpp:255, :303
//     dynamic_cast<string_index*>(ptr)->add(id, v)  database.cpp:257-264, :147
//     indices[key]->build() / ->query(range)        database.cpp:277, :392  (through `index*`)
// Only string_index changed: it no longer owns a CPU suffix array but a handle of the MI355X library
// (include/coffeedb_gpu.h).  bool/integer/double indexes keep the reference's behaviour.
#ifndef INDEX_GUARD
#define INDEX_GUARD
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

struct cdb_index;   // opaque GPU index (include/coffeedb_gpu.h)
struct cdb_shards;  // ... spread over several GPUs (COFFEEDB_GPUS)

class index {
public:
    using result_type = std::vector<std::pair<int64_t, int64_t>>;
    index() = default;
    index(const index&) = delete;
    index(index&&) = delete;
    index& operator=(const index&) = delete;
    index& operator=(index&&) = delete;
    virtual result_type query(const std::string& range) const;  // throws std::logic_error (index.h:16-18)
    virtual void build();                                        // throws std::logic_error (index.h:19-21)
    virtual ~index() = default;
};

class bool_index : public index {
public:
    using value_type = bool;
    static constexpr int8_t number = 0;
    void add(int64_t id, bool value);
    void build() override;
    result_type query(const std::string& range) const override;

private:
    std::array<std::vector<int64_t>, 2> data;
};

// integer and double columns share one implementation: (value, id) pairs sorted at build(), answered with
// two lower_bound calls over the parsed range (reference index.cpp:63-74, 154-173)
template <typename T, int8_t Tag>
class numeric_index : public index {
public:
    using value_type = T;
    static constexpr int8_t number = Tag;
    void add(int64_t id, T value) { rows.emplace_back(value, id); }
    void build() override;
    result_type query(const std::string& range) const override;

protected:
    std::vector<std::pair<T, int64_t>> rows;
};
class integer_index : public numeric_index<int64_t, 1> {};
class double_index : public numeric_index<double, 2> {};

class string_index : public index {
public:
    using value_type = std::string;
    static constexpr int8_t number = 3;
    string_index();
    ~string_index() override;
    // index.cpp:174-177.  As in the reference the index keeps a VIEW: the caller's string must stay valid until
    // build() has returned (the reference needs it for the index's whole life; here the text lives on the GPU afterwards).
    void add(int64_t id, std::string_view value);
    // index.cpp:178-236: suffix-array construction, now on the GPU.  Throws std::runtime_error with
    // the reference's messages for the capacity limits (index.cpp:196,199).
    void build() override;
    // index.cpp:237-326: (object id, occurrence count) per matching document, ascending insertion
    // order.  Throws std::runtime_error("Empty keywords are not allowed") for "" (index.cpp:239-241).
    result_type query(const std::string& keyword) const override;
    // Batched form used by callers that resolve many keywords at once (interface.cpp:79-113 loops
    // query() per keyword): one result list per keyword, same contents as query().
    std::vector<result_type> query_batch(const std::vector<std::string>& keywords) const;
    // OR over the keywords of this key — what filter() computes per key before the AND across keys
    // (interface.cpp:78-113): union by object id, counts summed, ascending id.  One GPU call.
    result_type query_any(const std::vector<std::string>& keywords) const;
    // The same union with the $correlation range filter and the ranking done on the GPU (interface.cpp:137-146
    // for a query on this one key): corr_lo <= count < corr_hi, descending count, ties ascending by object id
    // (the reference's unstable sort leaves that order open), at most `limit` rows (0 = all).
    result_type query_ranked(const std::vector<std::string>& keywords, int64_t corr_lo, int64_t corr_hi,
                             uint64_t limit = 0) const;
    // Highlight spans of every document that contains one of the keywords: (object id, [begin, end]
    // byte ranges, end inclusive) with the merge rule of ac_automaton::render (database.cpp:58-76).
    // Render with cdb_shim::render_spans (highlight.h).
    std::vector<std::pair<int64_t, std::vector<std::pair<uint64_t, uint64_t>>>> highlight_spans(
        const std::vector<std::string>& keywords) const;
    // NEW (no counterpart in the reference): announce a string column of roughly `bytes` bytes BEFORE the data is loaded —
    // start_server() calls init() and then build() (server.cpp:43-44); called at the start of init() with the size of the raw
    // directory (and any document as a sample of the alphabet) it lets the GPU map the first build's working set on a helper
    // thread while init() reads the files.  Returns at once; build() waits for it.  Purely an optimisation (cdb_reserve).
    static void reserve(uint64_t bytes, std::string_view sample = {});
    // NEW: waits for what the library does BEHIND build() on a helper thread — the order proof of the published array (every
    // adjacent pair against the text) and the mapping of the device memory a rebuild beside this index will ask for — and says
    // whether the order is proved.  Nothing needs to call it: queries are answered meanwhile, a later build() joins it (cdb_proof_wait).
    bool settle() const;

private:
    std::vector<int64_t> ids;        // index.h:58-59: ids and (non-owning) views of the documents, in add() order
    std::vector<const char*> ptrs;
    std::vector<uint64_t> lens;
    cdb_index* handle = nullptr;   // one GPU
    cdb_shards* shards = nullptr;  // several GPUs (environment COFFEEDB_GPUS; the library shards a column only when it
                                   // exceeds one GPU's share)
};
#endif
