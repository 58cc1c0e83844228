"""Pins the CPU oracle (oracle/cpu_ref.cpp) against the reference's own known answers
 before anything else is allowed to trust it."""
import json
import os

import numpy as np
import pytest

from coffeedb_amd import workloads as W
from oracle import OracleIndex, brute_count


def _load(golden_dir):
    with open(os.path.join(golden_dir, "reference_kat.json")) as f:
        return json.load(f)


def _mk(docs, ids):
    ix = OracleIndex()
    for i, d in zip(ids, docs):
        ix.add(i, d.encode() if isinstance(d, str) else d)
    ix.build()
    return ix


def test_reference_known_answers(golden_dir):
    g = _load(golden_dir)
    for case in g["cases"]:
        ix = _mk(case["docs"], case["ids"])
        assert (ix.bits, ix.mask, ix.size, ix.sa_width) == (case["bits"], case["mask"], case["size"], case["width"]), case["name"]
        if case["sa_off_doc"] is not None:
            ix.canonicalize()
            sa = ix.sa()
            got = [[int(e >> ix.bits), int(e & ix.mask)] for e in sa]
            assert got == case["sa_off_doc"], case["name"]
        for kw, want in case["queries"].items():
            assert ix.query(kw.encode()) == [tuple(r) for r in want], (case["name"], kw)


def test_readme_published_counts(golden_dir):
    # /root/reference/README.md:80-92 — "010" occurs 2x in "3010103" and 1x in "301022"
    g = _load(golden_dir)["readme_published"]
    ix = _mk(g["docs"], [1, 2])
    assert [c for _, c in ix.query(g["keyword"].encode())] == g["correlation"]


def test_empty_keyword_error(golden_dir):
    ix = _mk(["abc"], [1])
    with pytest.raises(RuntimeError, match=_load(golden_dir)["empty_keyword_error"]):
        ix.query(b"")


def test_width_rule():
    # Q4: mask grows while mask < count -> 2^k docs need k+1 bits; length 256 needs 9 offset bits.
    ix = _mk(["a" * 256] * 4, range(4))
    assert (ix.bits, ix.sa_width) == (3, 4)
    blob, ds = W.ascii_corpus(1 << 10, 1 << 10, seed=3)  # 11 + 11 bits -> u32
    ix = OracleIndex(); ix.add_bulk(np.arange(1 << 10), blob, ds); ix.build()
    assert (ix.bits, ix.sa_width) == (11, 4)


def _check_against_brute(blob, ds, ids, patterns, nthreads=0):
    ix = OracleIndex()
    ix.add_bulk(ids, blob, ds)
    ix.build(nthreads)
    assert ix.inversions() == 0
    for kw in patterns:
        want = brute_count(blob, ds, kw)
        got = dict(ix.query(kw))
        nz = np.nonzero(want)[0]
        assert got == {int(ids[d]): int(want[d]) for d in nz}, kw
    return ix


def test_property_test_string_shape():
    # test/test-string.py shape scaled down: a-z docs, random 3-char keywords, overlapping counts.
    blob, ds = W.ascii_corpus(200, 500, seed=11, lo=0x61, hi=0x7A)
    ids = np.arange(200, dtype=np.int64) * 7 + 3
    pats = [bytes(W.random_bytes(3, 100 + i, 0x61, 0x7A)) for i in range(40)] + [b"a", b"zz"]
    ix = _check_against_brute(blob, ds, ids, pats)
    assert ix.size == 200 * 500 and ix.size > 4096  # radix path exercised (Q5)


def test_property_c0_sample():
    # BASELINE config 0 shape (10k x 256 printable ASCII) with sampled + missing patterns.
    blob, ds = W.ascii_corpus(10000, 256, seed=12345)
    pb, po = W.sample_patterns(blob, ds, 60, 4, 16, seed=5)
    pats = [bytes(pb[int(po[i]):int(po[i + 1])]) for i in range(60)]
    ids = np.arange(10000, dtype=np.int64)
    _check_against_brute(blob, ds, ids, pats + [b" ", b"~~"])


def test_ragged_and_empty_docs_and_threads():
    blob, ds = W.ragged_corpus(3000, 40, seed=5, empty_every=7)
    ids = np.arange(3000, dtype=np.int64)[::-1].copy()
    pats = [b"ab", b"q", b"xyz", b"aaaa"]
    a = _check_against_brute(blob, ds, ids, pats, nthreads=1)
    b = _check_against_brute(blob, ds, ids, pats, nthreads=4)
    a.canonicalize(); b.canonicalize()
    assert np.array_equal(a.sa(), b.sa())


def test_duplicate_documents_tie_runs():
    # identical documents -> every suffix ties across docs; canonical order is by doc index.
    docs = [b"banana", b"banana", b"ban", b"banana"]
    ix = _mk(docs, [5, 6, 7, 8])
    runs = ix.canonicalize()
    assert runs > 0
    sa = ix.sa()
    assert ix.query(b"ana") == [(5, 2), (6, 2), (8, 2)]
    assert ix.query(b"ban") == [(5, 1), (6, 1), (7, 1), (8, 1)]
    # within a run of equal suffixes docs ascend
    text = [d for d in docs]
    suf = [text[int(e & ix.mask)][int(e >> ix.bits):] for e in sa]
    for i in range(1, len(sa)):
        assert suf[i - 1] <= suf[i]
        if suf[i - 1] == suf[i]:
            assert (sa[i - 1] & ix.mask) < (sa[i] & ix.mask)


def test_high_bytes_expose_signed_quirk():
    # Q2: radix nodes bucket in signed-char order, leaves/binary search compare unsigned.  With
    # bytes >= 0x80 and n > chuck_size the reference SA is not globally sorted.
    blob, ds = W.ascii_corpus(400, 64, seed=21, lo=0x00, hi=0xFF)
    ix = OracleIndex(); ix.add_bulk(np.arange(400), blob, ds); ix.build()
    assert ix.inversions() > 0
    # below the radix threshold (n <= 4096) the whole array is one comparison-sorted leaf
    blob, ds = W.ascii_corpus(40, 64, seed=21, lo=0x00, hi=0xFF)
    ix = OracleIndex(); ix.add_bulk(np.arange(40), blob, ds); ix.build()
    assert ix.inversions() == 0


def test_batch_matches_single():
    blob, ds = W.ascii_corpus(2000, 100, seed=8)
    ix = OracleIndex(); ix.add_bulk(np.arange(2000) + 10, blob, ds); ix.build()
    pb, po = W.sample_patterns(blob, ds, 200, 1, 6, seed=6)
    rp, ids, cnt, hits = ix.query_batch(pb, po, nthreads=3)
    for j in range(200):
        kw = bytes(pb[int(po[j]):int(po[j + 1])])
        want = ix.query(kw)
        got = list(zip(ids[int(rp[j]):int(rp[j + 1])].tolist(), cnt[int(rp[j]):int(rp[j + 1])].tolist()))
        assert got == want
    assert hits == int(cnt.sum())


def _model_cases():
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_model_fixtures", os.path.join(here, "golden", "make_model_fixtures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_oracle_matches_the_second_restatement_fixtures(golden_dir):
    # tests/golden/model_cases.json: canonical suffix arrays (SHA-256) and keyword rows of the fixture classes (3)-(6) of
    # SURVEY.md §8(c) — radix nodes over several levels, ragged documents, bytes >= 0x80 (signed child order, unsigned
    # leaves: the array is not globally sorted and some counts are "wrong" exactly like the reference's), the u32 / u64
    # width boundary — computed by tests/ref_model.py, a pure-Python reading of index.cpp that shares no code with
    # oracle/cpu_ref.cpp.
    mod = _model_cases()
    with open(os.path.join(golden_dir, "model_cases.json")) as f:
        want = json.load(f)["cases"]
    for name, (blob, ds, ids) in mod.cases().items():
        w = want[name]
        ix = OracleIndex()
        ix.add_bulk(ids, blob, ds)
        ix.build()
        ix.canonicalize()
        assert (ix.size, ix.bits, ix.mask, ix.sa_width) == (w["size"], w["bits"], w["mask"], w["width"]), name
        assert mod.sa_hash(ix.sa(), ix.sa_width) == w["sa_sha256"], name
        for kw, rows in w["queries"].items():
            assert ix.query(bytes.fromhex(kw)) == [tuple(r) for r in rows], (name, kw)
    assert want["width_32_bits_u32"]["width"] == 4 and want["width_33_bits_u64"]["width"] == 8


def test_second_restatement_agrees_on_random_small_inputs():
    # the model itself, run live on small random inputs (several alphabets, empty documents, duplicate documents, a
    # thread count sweep on the oracle side): entry-for-entry equal arrays and equal rows
    from ref_model import RefModel
    rng = np.random.default_rng(7)
    for trial in range(12):
        nd = int(rng.integers(1, 40))
        lo, hi = [(0x61, 0x62), (0x61, 0x7A), (0x00, 0xFF), (0x7E, 0x81)][trial % 4]
        docs = [bytes(rng.integers(lo, hi + 1, size=int(rng.integers(0, 400)), dtype=np.uint8)) for _ in range(nd)]
        if trial % 3 == 0 and nd > 2:
            docs[1] = docs[0]
        if sum(len(d) for d in docs) == 0:
            continue
        ids = (rng.permutation(nd) * 5 - 7).tolist()
        m = RefModel(ids, docs)
        ix = _mk(docs, ids)
        ix.canonicalize()
        assert (ix.size, ix.bits, ix.mask, ix.sa_width) == (m.size, m.bits, m.mask, m.width)
        assert ix.sa().tolist() == m.sa
        text = b"".join(docs)
        for _ in range(25):
            a = int(rng.integers(0, len(text)))
            kw = text[a:a + int(rng.integers(1, 6))]
            assert ix.query(kw) == m.query(kw), kw
