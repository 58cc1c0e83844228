"""BASELINE.json configs[1] at FULL size (2^20 docs x 1024 B = 1 GiB, 100k patterns) on the GPU, checked
through size-independent properties because no CPU oracle finishes at this size:
  * the suffix array is a sorted permutation with canonical tie order (cdb_debug_verify: adjacent-suffix
    comparison on the GPU by code independent of the build),
  * per-pattern rows are consistent (counts sum to the hit total, ids ascend, sampled patterns hit),
  * a sample of patterns is compared with an independent brute-force scan of the whole text (torch ops)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _give_memory_back():
    """The big corpora live in torch tensors and the library's block cache: hand both back after every test."""
    yield
    import gc
    import torch
    from coffeedb_amd import capi
    gc.collect()
    torch.cuda.empty_cache()
    capi.load_library().cdb_release_cached_memory()


def _brute_counts(torch, text, pat, doclen):
    """{doc: overlapping occurrences} of `pat` by a full scan (matches must not cross documents)."""
    m = len(pat)
    n = text.numel()
    ok = text[: n - m + 1] == pat[0]
    for k in range(1, m):
        ok &= text[k: n - m + 1 + k] == pat[k]
    pos = torch.nonzero(ok).flatten()
    pos = pos[(pos % doclen) + m <= doclen]
    docs, cnt = torch.unique(pos // doclen, return_counts=True)
    return dict(zip(docs.tolist(), cnt.tolist()))


def test_c1_full_size_properties():
    import torch
    from coffeedb_amd import capi, workloads as W
    nd, dl, npat = 1 << 20, 1024, 100_000
    text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
    ds = W.uniform_docs(nd, dl)
    ids = np.arange(nd, dtype=np.int64) * 3 + 7
    # patterns sampled from a host copy of the first 64 MiB (guaranteed hits) + 10 % perturbed
    host = text[: 1 << 26].cpu().numpy()
    pb, po = W.sample_patterns(host, W.uniform_docs(1 << 16, dl), npat, 4, 16, seed=99)
    torch.cuda.synchronize()
    g = capi.GpuStringIndex()
    g.build_device(text.data_ptr(), ds, ids)
    assert (g.size, g.bits, g.sa_width) == (nd * dl, 21, 4)      # 21 + 11 bits = 32 -> u32 edge (Q4)
    assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0
    v = g.verify()
    assert v["inversions"] == 0 and v["tie_violations"] == 0 and v["invalid_entries"] == 0
    assert v["entry_sum"] == v["expected_entry_sum"]

    rp, ri, rc, hits = g.query_batch(pb, po)
    assert len(rp) == npat + 1 and rp[0] == 0 and rp[-1] == len(ri) == len(rc)
    assert int(rc.sum()) == hits and (rc > 0).all()
    rows = np.diff(rp.astype(np.int64))
    assert (rows > 0).mean() > 0.88                                # the unperturbed ~90 % must hit
    inner = np.ones(len(ri), dtype=bool)
    inner[rp[1:-1].astype(np.int64)[rows[1:] > 0]] = False         # first row of each non-empty pattern
    inner[0] = False
    assert (np.diff(ri)[inner[1:]] > 0).all()                      # ids ascend within a pattern (doc order)

    rng = np.random.default_rng(5)
    for j in rng.choice(npat, 40, replace=False).tolist() + [int(np.argmin(np.diff(po.astype(np.int64))))]:
        kw = pb[int(po[j]):int(po[j + 1])]
        want = _brute_counts(torch, text, torch.from_numpy(kw.copy()).cuda(), dl)
        a, b = int(rp[j]), int(rp[j + 1])
        got = {(int(i) - 7) // 3: int(c) for i, c in zip(ri[a:b], rc[a:b])}
        assert got == want, (j, bytes(kw))
    assert g.proof_wait(300_000) == 2 and g.stat("self_check_fallbacks") == 0   # order proof behind the build: every adjacent pair
    g.close()


def test_beyond_4gib_on_one_gpu():
    """A corpus of more than 2^32 bytes (README benchmark shape, reference README.md:226-232: documents of
    128 KiB of a-z text; here 36864 of them = 4.5 GiB) on ONE GPU: 8-byte entries, 64-bit ranks, bucket-wise
    initial sort.  Checked by the GPU-side verifier and brute-force scans of sampled keywords."""
    import torch
    from coffeedb_amd import capi, workloads as W
    nd, dl = 36864, 131072
    n = nd * dl
    assert n > 1 << 32
    text = W.random_bytes_torch(n, 777, 0x61, 0x7A, device="cuda")
    ds = W.uniform_docs(nd, dl)
    ids = np.arange(nd, dtype=np.int64) + 1_000_000
    torch.cuda.synchronize()
    g = capi.GpuStringIndex()
    g.build_device(text.data_ptr(), ds, ids)
    assert (g.size, g.sa_width, g.bits) == (n, 8, 16) and g.stat("bucketed") == 1   # 16 + 18 bits -> u64 entries
    assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0
    v = g.verify()
    assert v["inversions"] == 0 and v["tie_violations"] == 0 and v["invalid_entries"] == 0
    assert v["entry_sum"] == v["expected_entry_sum"]
    pb = W.random_bytes(5 * 2000, 4242, 0x61, 0x7A)                       # test/benchmark.py: 5-char keywords
    po = (np.arange(2001) * 5).astype(np.uint64)
    rp, ri, rc, hits = g.query_batch(pb, po)
    assert int(rc.sum()) == hits and hits > 2000 * 300                    # ~ n / 26^5 = 407 hits per keyword
    for j in (0, 1234):
        kw = torch.from_numpy(pb[5 * j:5 * j + 5].copy()).cuda()
        per_doc = {}
        step = 1 << 30
        for s0 in range(0, n, step):
            seg = text[s0:min(n, s0 + step + 4)]
            ok = seg[: len(seg) - 4] == kw[0]
            for k in range(1, 5):
                ok &= seg[k: len(seg) - 4 + k] == kw[k]
            pos = torch.nonzero(ok).flatten() + s0
            pos = pos[((pos % dl) + 5 <= dl) & (pos < s0 + step)]
            d, c = torch.unique(pos // dl, return_counts=True)
            for a, b in zip(d.tolist(), c.tolist()):
                per_doc[a] = per_doc.get(a, 0) + b
        a, b = int(rp[j]), int(rp[j + 1])
        assert {int(i) - 1_000_000: int(c) for i, c in zip(ri[a:b], rc[a:b])} == per_doc
    assert g.proof_wait(300_000) == 2 and g.stat("self_check_fallbacks") == 0   # order proof behind the build: every adjacent pair
    g.close()
    capi.load_library().cdb_release_cached_memory()


def _scan_occurrences(torch, text, kw, chunk=1 << 30):
    """All start positions of `kw` in `text` (device tensors) by a chunked scan: the first three bytes are compared
    densely, the rest only at the surviving positions."""
    m, n = len(kw), text.numel()
    out = []
    for s0 in range(0, n, chunk):
        seg = text[s0:min(n, s0 + chunk + m - 1)]
        L = seg.numel() - m + 1
        if L <= 0:
            break
        ok = seg[:L] == kw[0]
        for k in range(1, min(m, 3)):
            ok &= seg[k:L + k] == kw[k]
        pos = torch.nonzero(ok).flatten()
        for k in range(3, m):
            pos = pos[seg[pos + k] == kw[k]]
        out.append(pos + s0)
    return torch.cat(out) if out else torch.empty(0, dtype=torch.int64, device=text.device)


def test_c2_full_size_zipf_one_million_patterns_with_offsets():
    """BASELINE.json configs[2] at FULL size: 2^23 docs x 1024 B of Zipf(64) text = 8 GiB (8-byte entries, bucket-wise
    initial sort, text-extension rounds), one batch of 10^6 patterns with occurrence offsets.  Checked through
    size-independent properties and brute-force scans of sampled patterns incl. their offsets."""
    import torch
    from coffeedb_amd import capi, workloads as W
    nd, dl, npat = 1 << 23, 1024, 1_000_000
    n = nd * dl
    text = W.zipf_bytes_torch(n, seed=2, device="cuda")
    ds = W.uniform_docs(nd, dl)
    ids = np.arange(nd, dtype=np.int64) * 2 + 1
    d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
    d_blob, d_offs, nbytes = W.sample_patterns_torch(text, d_ds, npat, 6, 16, seed=99, miss_byte=0x7F)
    pb = d_blob[:nbytes].cpu().numpy()
    po = d_offs.cpu().numpy().astype(np.uint64)
    del d_blob, d_offs
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    g = capi.GpuStringIndex()
    g.build_device(text.data_ptr(), ds, ids)
    assert (g.size, g.bits, g.sa_width) == (n, 24, 8) and g.stat("bucketed") == 1       # 24 + 11 bits -> u64 entries
    assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0
    # skewed text: variable-length keys (vl_code.h), and their unresolved share as the cost model promised
    assert g.stat("vl_key_bits") in (40, 48) and g.stat("unresolved_after_initial") < n / 24
    v = g.verify()
    assert v["inversions"] == 0 and v["tie_violations"] == 0 and v["invalid_entries"] == 0
    assert v["entry_sum"] == v["expected_entry_sum"]

    rp, ri, rc, hp, off = g.query_batch_offsets(pb, po)
    hits = int(hp[-1])
    assert len(rp) == npat + 1 and rp[0] == 0 and rp[-1] == len(ri) == len(rc) == len(hp) - 1
    assert int(rc.sum()) == hits == len(off) and (rc > 0).all()
    assert np.array_equal(np.diff(hp.astype(np.int64)), rc)                              # a row owns `count` offsets
    rows = np.diff(rp.astype(np.int64))
    assert (rows > 0).mean() > 0.88
    # offsets ascend inside a row and every occurrence fits its document
    inner = np.ones(len(off), dtype=bool)
    inner[hp[:-1].astype(np.int64)] = False
    assert (np.diff(off.astype(np.int64))[inner[1:]] > 0).all() and int(off.max()) < dl
    rng = np.random.default_rng(11)
    lens = np.diff(po.astype(np.int64))
    picks = rng.choice(npat, 10, replace=False).tolist() + [int(np.argmax(rows)), int(np.argmin(lens))]
    for j in picks:
        kw = pb[int(po[j]):int(po[j + 1])]
        pos = _scan_occurrences(torch, text, torch.from_numpy(kw.copy()).cuda())
        pos = pos[(pos % dl) + len(kw) <= dl]                                            # matches never cross documents
        want_docs, want_cnt = torch.unique(pos // dl, return_counts=True)
        a, b = int(rp[j]), int(rp[j + 1])
        assert np.array_equal((ri[a:b] - 1) // 2, want_docs.cpu().numpy()), (j, bytes(kw))
        assert np.array_equal(rc[a:b], want_cnt.cpu().numpy()), (j, bytes(kw))
        got_off = off[int(hp[a]):int(hp[b])]
        assert np.array_equal(got_off, (pos % dl).cpu().numpy().astype(np.uint64)), (j, bytes(kw))   # (doc, offset) ascending
    del rp, ri, rc, hp, off

    # ---- the short-pattern tail of SURVEY §8(d)'s C2 (m from 2): keywords of 2..5 bytes match 10^4 .. 4 x 10^8 suffixes
    # each on this text — the regime where the reference switches to its 17-bit radix (index.cpp:288-315) and where the
    # batch is resolved in chunks of patterns under the hit budget (2^31 hits of sort scratch).  Results stay in HBM
    # (3 x 10^9 hits with offsets are ~50 GB); rows and offsets of sampled keywords are checked by brute-force scans.
    nshort = 200
    s_blob, s_offs, s_bytes = W.sample_patterns_torch(text, d_ds, nshort, 2, 5, seed=5, miss_byte=0x7F)
    torch.cuda.synchronize()
    g.set_option("query_hit_budget", 1 << 30)                                            # several chunks at this size
    r, hx = g.query_batch_offsets_device(s_blob.data_ptr(), s_offs.data_ptr(), nshort, s_bytes)

    def dev(ptr, cnt):
        class A:
            __cuda_array_interface__ = {"shape": (int(cnt),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(A(), device="cuda")
    nrows, nhits = int(r.nrows), int(r.nhits)
    assert nhits > 1 << 30 and nrows > 100_000_000, (nhits, nrows)                       # far beyond one chunk
    rp_d, rc_d, ri_d = dev(r.d_row_ptr, nshort + 1), dev(r.d_counts, nrows), dev(r.d_ids, nrows)
    hp_d, off_d = dev(hx.d_hit_ptr, nrows + 1), dev(hx.d_offsets, nhits)
    assert int(rp_d[0]) == 0 and int(rp_d[-1]) == nrows and int(hp_d[-1]) == nhits
    assert int(rc_d.sum()) == nhits and bool((rc_d > 0).all())
    assert bool((hp_d[1:] - hp_d[:-1] == rc_d).all())                                    # a row owns `count` offsets
    srows = (rp_d[1:] - rp_d[:-1]).cpu().numpy()
    so = s_offs.cpu().numpy()
    sb = s_blob[:s_bytes].cpu().numpy()
    for j in [int(np.argmax(srows)), int(np.argmax(np.diff(so))), 17]:
        kw = sb[int(so[j]):int(so[j + 1])]
        pos = _scan_occurrences(torch, text, torch.from_numpy(kw.copy()).cuda())
        pos = pos[(pos % dl) + len(kw) <= dl]
        want_docs, want_cnt = torch.unique(pos // dl, return_counts=True)
        a, b = int(rp_d[j]), int(rp_d[j + 1])
        assert b - a == want_docs.numel(), (j, bytes(kw), b - a, want_docs.numel())
        assert bool(((ri_d[a:b] - 1) // 2 == want_docs).all()) and bool((rc_d[a:b] == want_cnt).all()), (j, bytes(kw))
        assert bool((off_d[int(hp_d[a]):int(hp_d[b])] == pos % dl).all()), (j, bytes(kw))
        del pos, want_docs, want_cnt
    assert g.proof_wait(300_000) == 2 and g.stat("self_check_fallbacks") == 0   # order proof behind the build: every adjacent pair
    g.close()
    capi.load_library().cdb_release_cached_memory()


def test_utf8_4gib_reference_order_and_true_order():
    """north_star target shape: 4 GiB of valid UTF-8 on one GPU.  reference_compat = 1 must give the REFERENCE's order
    (signed child order inside radix nodes, index.h:66-73 / index.cpp:96-126; checked by cdb_debug_verify_reference),
    reference_compat = 0 the plain unsigned order with true counts (brute-force scans)."""
    import torch
    from coffeedb_amd import capi, workloads as W
    text, ds = W.utf8_bytes_torch(4 << 30, seed=4, device="cuda")
    nd, n = len(ds) - 1, int(ds[-1])
    assert n > (4 << 30) - (64 << 20) and n % 16 == 0
    text[: 1 << 20].cpu().numpy().tobytes()[: int(ds[64])].decode("utf-8")             # valid UTF-8, cut at code points
    ids = np.arange(nd, dtype=np.int64)
    d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
    d_blob, d_offs, nbytes = W.sample_patterns_torch(text, d_ds, 100_000, 4, 16, seed=99, miss_byte=0xFF, utf8=True)
    pb = d_blob[:nbytes].cpu().numpy()
    po = d_offs.cpu().numpy().astype(np.uint64)
    del d_blob, d_offs
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    g = capi.GpuStringIndex()
    g.build_device(text.data_ptr(), ds, ids)                                              # reference_compat = 1 (default)
    assert g.size == n and g.sa_width == 8 and g.stat("compat_rotations") >= 1
    assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0
    v = g.verify()
    assert v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"] and v["tie_violations"] == 0
    assert v["inversions"] == g.stat("compat_rotations")        # not globally sorted: one inversion per rotated node
    r = g.verify_reference()
    assert r["violations"] == 0 and r["tie_violations"] == 0, r
    assert r["radix_node_pairs"] == g.stat("compat_rotations") and r["mixed_pairs"] > r["radix_node_pairs"], r
    # the lone-keyword kernel walks the reference's own bisections on this array; the batch must agree with it
    rp, ri, rc, hits = g.query_batch(pb[: int(po[2000])], po[:2001])
    for j in range(0, 2000, 97):
        a, b = int(rp[j]), int(rp[j + 1])
        assert g.query(bytes(pb[int(po[j]):int(po[j + 1])])) == list(zip(ri[a:b].tolist(), rc[a:b].tolist())), j
    assert g.proof_wait(300_000) == 2 and g.stat("self_check_fallbacks") == 0   # order proof behind the build: every adjacent pair
    g.close()

    g = capi.GpuStringIndex()
    g.set_option("reference_compat", 0)
    g.build_device(text.data_ptr(), ds, ids)
    assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0
    v = g.verify()
    assert v["inversions"] == 0 and v["tie_violations"] == 0 and v["invalid_entries"] == 0
    assert v["entry_sum"] == v["expected_entry_sum"]
    rp, ri, rc, hits = g.query_batch(pb, po)
    assert int(rc.sum()) == hits and (np.diff(rp.astype(np.int64)) > 0).mean() > 0.88
    for j in np.random.default_rng(3).choice(100_000, 8, replace=False).tolist():
        kw = pb[int(po[j]):int(po[j + 1])]
        pos = _scan_occurrences(torch, text, torch.from_numpy(kw.copy()).cuda())
        doc = torch.searchsorted(d_ds, pos, right=True) - 1
        pos = pos[pos + len(kw) <= d_ds[doc + 1]]
        doc = torch.searchsorted(d_ds, pos, right=True) - 1
        wd, wc = torch.unique(doc, return_counts=True)
        a, b = int(rp[j]), int(rp[j + 1])
        assert np.array_equal(ri[a:b], wd.cpu().numpy()) and np.array_equal(rc[a:b], wc.cpu().numpy()), (j, bytes(kw))
    assert g.proof_wait(300_000) == 2 and g.stat("self_check_fallbacks") == 0   # order proof behind the build: every adjacent pair
    g.close()
    capi.load_library().cdb_release_cached_memory()


def test_c4_shard_16gib_utf8_ten_million_patterns():
    """One GPU's share of BASELINE.json configs[4]: 16 GiB of valid UTF-8 (8-byte entries, reference order), a batch
    of 10^7 patterns, $correlation ranking of a keyword list."""
    import torch
    from coffeedb_amd import capi, workloads as W
    free, _ = torch.cuda.mem_get_info()
    if free < (250 << 30):
        pytest.skip("needs a whole 288 GB MI355X")
    text, ds = W.utf8_bytes_torch(16 << 30, seed=5, device="cuda")
    nd, n = len(ds) - 1, int(ds[-1])
    ids = np.arange(nd, dtype=np.int64) + 10
    d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
    npat = 10_000_000
    d_blob, d_offs, nbytes = W.sample_patterns_torch(text, d_ds, npat, 4, 16, seed=7, miss_byte=0xFF, utf8=True)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    g = capi.GpuStringIndex()
    g.build_device(text.data_ptr(), ds, ids)
    assert g.size == n and g.sa_width == 8 and g.stat("bucketed") == 1
    assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0
    assert g.stat("vl_key_bits") == 0 and g.stat("unresolved_after_initial") < n / 24
    v = g.verify()
    assert v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"] and v["tie_violations"] == 0
    r = g.verify_reference()
    assert r["violations"] == 0 and r["radix_node_pairs"] >= 1, r
    res = g.query_batch_device(d_blob.data_ptr(), d_offs.data_ptr(), npat, nbytes)
    assert int(res.npat) == npat and int(res.nrows) > 0.85 * npat and int(res.nhits) >= int(res.nrows)
    # the first 20 000 patterns again through the host API with the plain reference-sequence search: same rows
    hb = d_blob[: int(d_offs[20_000].item())].cpu().numpy()
    ho = d_offs[:20_001].cpu().numpy().astype(np.uint64)
    fast = g.query_batch(hb, ho)
    g.set_option("fast_search", 0)
    g.set_option("wave_rows", 0)
    slow = g.query_batch(hb, ho)
    assert fast[3] == slow[3] and all(np.array_equal(a, b) for a, b in zip(fast[:3], slow[:3]))
    # device rows of those patterns equal the host API's
    rp_d = torch.as_tensor(_Dev(res.d_row_ptr, npat + 1), device="cuda")[:20_001].cpu().numpy()
    assert np.array_equal(rp_d.astype(np.uint64), fast[0])
    # $correlation ranking of the union over 1000 keywords: descending counts, filter respected, top rows consistent
    kws = [bytes(hb[int(ho[j]):int(ho[j + 1])]) for j in range(1000)]
    g.set_option("fast_search", 1)
    ranked = g.query_ranked(kws, lo=1, limit=50)
    union = dict(g.query_or(kws))
    assert len(ranked) == min(50, len(union)) and all(union[i] == c for i, c in ranked)
    assert [c for _, c in ranked] == sorted(union.values(), reverse=True)[: len(ranked)]
    assert g.proof_wait(300_000) == 2 and g.stat("self_check_fallbacks") == 0   # order proof behind the build: every adjacent pair
    g.close()
    capi.load_library().cdb_release_cached_memory()


class _Dev:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def test_reference_test_string_at_its_real_size():
    """test/test-string.py:25-56 as the reference runs it: 5000 docs x 5000 random a-z characters, 100 random 3-character
    keywords, and for EVERY document $correlation == overlapping brute-force count (absent when 0)."""
    from coffeedb_amd import capi, workloads as W
    nd, dl = 5000, 5000
    blob, ds = W.ascii_corpus(nd, dl, seed=2024, lo=0x61, hi=0x7A)
    ids = np.arange(nd, dtype=np.int64)
    g = capi.GpuStringIndex()
    g.add_bulk(ids, blob, ds)
    g.build()
    mat = blob.reshape(nd, dl)
    for i in range(100):
        kw = W.random_bytes(3, 9000 + i, 0x61, 0x7A)
        ok = (mat[:, :-2] == kw[0]) & (mat[:, 1:-1] == kw[1]) & (mat[:, 2:] == kw[2])    # count(): overlapping occurrences
        want = ok.sum(axis=1)
        got = dict(g.query(bytes(kw)))
        assert got == {int(d): int(want[d]) for d in np.nonzero(want)[0]}, bytes(kw)
    g.close()


def test_reference_test_highlight_sequential_replace():
    """test/test-highlight.py:31-60 restated: 5 keywords of 4 characters over disjoint letters of a shuffled alphabet; the
    result set must be the documents `str.replace` changes, and every rendered document must equal the sequential
    replace(k, "<b>" + k + "</b>") — here from cdb_query_spans + the shim's span rendering."""
    from coffeedb_amd import capi, workloads as W
    nd, dl = 5000, 5000
    blob, ds = W.ascii_corpus(nd, dl, seed=77, lo=0x61, hi=0x7A)
    letters = [chr(c) for c in range(0x61, 0x7B)]
    np.random.default_rng(5).shuffle(letters)
    kws = ["".join(letters[4 * i:4 * i + 4]).encode() for i in range(5)]
    # plant some occurrences (random text holds ~55 of each; make runs and document edges certain)
    planted = blob.copy()
    planted[0:4] = np.frombuffer(kws[0], dtype=np.uint8)
    planted[dl - 4:dl] = np.frombuffer(kws[1], dtype=np.uint8)
    planted[3 * dl + 10:3 * dl + 14] = np.frombuffer(kws[2], dtype=np.uint8)
    planted[3 * dl + 14:3 * dl + 18] = np.frombuffer(kws[3], dtype=np.uint8)              # adjacent occurrences: two spans
    ids = np.arange(nd, dtype=np.int64) + 1000
    g = capi.GpuStringIndex()
    g.add_bulk(ids, planted, ds)
    g.build()
    spans = dict(g.query_spans(kws))
    docs = [planted[i * dl:(i + 1) * dl].tobytes() for i in range(nd)]
    changed = set()
    for i, text in enumerate(docs):
        want = text
        for k in kws:
            want = want.replace(k, b"<b>" + k + b"</b>")
        if want != text:
            changed.add(1000 + i)
            out, last = [], 0
            for b, e in spans[1000 + i]:                                                  # shim/highlight.h render_spans
                out += [text[last:b], b"<b>", text[b:e + 1], b"</b>"]
                last = e + 1
            out.append(text[last:])
            assert b"".join(out) == want, i
    assert set(spans) == changed and len(changed) > 200
    assert g.proof_wait(300_000) == 2 and g.stat("self_check_fallbacks") == 0   # order proof behind the build: every adjacent pair
    g.close()


def test_c3_shard_8gib_ascii():
    """One GPU's share of BASELINE.json configs[3] (32 GiB over 4 GPUs): 2^23 docs x 1024 B of printable ASCII = 8 GiB,
    8-byte entries, bucket-wise build; sorted permutation + brute-force scans of sampled patterns."""
    import torch
    from coffeedb_amd import capi, workloads as W
    nd, dl = 1 << 23, 1024
    n = nd * dl
    text = W.random_bytes_torch(n, 12345, 0x20, 0x7E, stream=3, device="cuda")
    ds = W.uniform_docs(nd, dl)
    ids = np.arange(nd, dtype=np.int64)
    d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
    d_blob, d_offs, nbytes = W.sample_patterns_torch(text, d_ds, 100_000, 4, 16, seed=99, miss_byte=0x7F)
    pb = d_blob[:nbytes].cpu().numpy()
    po = d_offs.cpu().numpy().astype(np.uint64)
    del d_blob, d_offs
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    g = capi.GpuStringIndex()
    g.build_device(text.data_ptr(), ds, ids)
    assert (g.size, g.bits, g.sa_width) == (n, 24, 8) and g.stat("bucketed") == 1
    assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0
    # a flat alphabet keeps the dense keys (an alphabetic code cannot beat log2(96) bits per symbol), and few suffixes stay open
    assert g.stat("vl_key_bits") == 0 and g.stat("unresolved_after_initial") < n / 64
    v = g.verify()
    assert v["inversions"] == 0 and v["tie_violations"] == 0 and v["invalid_entries"] == 0
    assert v["entry_sum"] == v["expected_entry_sum"]
    rp, ri, rc, hits = g.query_batch(pb, po)
    assert int(rc.sum()) == hits and (np.diff(rp.astype(np.int64)) > 0).mean() > 0.88
    for j in np.random.default_rng(2).choice(100_000, 6, replace=False).tolist():
        kw = pb[int(po[j]):int(po[j + 1])]
        pos = _scan_occurrences(torch, text, torch.from_numpy(kw.copy()).cuda())
        pos = pos[(pos % dl) + len(kw) <= dl]
        wd, wc = torch.unique(pos // dl, return_counts=True)
        a, b = int(rp[j]), int(rp[j + 1])
        assert np.array_equal(ri[a:b], wd.cpu().numpy()) and np.array_equal(rc[a:b], wc.cpu().numpy()), (j, bytes(kw))
    assert g.proof_wait(300_000) == 2 and g.stat("self_check_fallbacks") == 0   # order proof behind the build: every adjacent pair
    g.close()


def test_rebuild_8gib_column_beside_the_serving_index():
    """database.cpp:276-280: build() prepares the NEW index while the old one keeps answering, and frees the old one only
    afterwards — two generations of an 8 GiB column (8-byte entries: 64 GiB of suffix array each) and the build's scratch
    must fit the 288 GB together.  The build sizes its bucket groups from the memory that is really free
    (sa_build.hip: hipMemGetInfo + the block cache), so beside a serving index it takes several groups instead of failing.
    Checked: the new generation verifies, the old one answered every batch meanwhile with its own rows, both are resident
    at the end; the memory figures go to stdout (pytest -s) and into DESIGN §3."""
    import threading
    import torch
    from coffeedb_amd import capi, workloads as W
    nd, dl = 1 << 23, 1024
    n = nd * dl
    ds = W.uniform_docs(nd, dl)
    d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
    d_ids = torch.arange(nd, dtype=torch.int64, device="cuda")
    gens = []
    capi.load_library().cdb_release_cached_memory()
    capi.memory_reset_peak()
    text_a = W.random_bytes_torch(n, 77, device="cuda")
    pa_blob, pa_offs, pa_bytes = W.sample_patterns_torch(text_a, d_ds, 20_000, 6, 14, seed=5, miss_byte=0x7F)
    torch.cuda.synchronize()
    old = capi.GpuStringIndex()
    old.build_resident(text_a.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), nd)
    assert old.sa_width == 8 and old.stat("self_check_fallbacks") == 0
    gens.append(dict(zip(("in_use", "peak", "cached"), capi.memory_stats()), groups=old.stat("bucket_groups"), fused=old.stat("fused_records")))
    want = old.query_batch_device(pa_blob.data_ptr(), pa_offs.data_ptr(), 20_000, pa_bytes)
    want_rows, want_hits = int(want.nrows), int(want.nhits)
    assert want_hits >= 18_000
    capi.load_library().cdb_release_cached_memory()          # (what a long-running server's cache may or may not hold: start clean)
    capi.memory_reset_peak()

    stop, errors, served = threading.Event(), [], [0]

    def serve():
        try:
            while not stop.is_set():
                r = old.query_batch_device(pa_blob.data_ptr(), pa_offs.data_ptr(), 20_000, pa_bytes)
                assert (int(r.nrows), int(r.nhits)) == (want_rows, want_hits)
                served[0] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = threading.Thread(target=serve)
    th.start()
    text_b = W.random_bytes_torch(n, 78, device="cuda")        # the new generation of the column (database.cpp:171-172)
    torch.cuda.synchronize()
    new = capi.GpuStringIndex()
    new.build_resident(text_b.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), nd)
    b_ms = new.stat("build_ms")
    stop.set()
    th.join()
    assert not errors, errors[:2]
    assert served[0] >= 1
    assert new.stat("self_check_fallbacks") == 0 and new.stat("group_fallbacks") == 0
    v = new.verify()
    assert v["inversions"] == 0 and v["tie_violations"] == 0 and v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"]
    gens.append(dict(zip(("in_use", "peak", "cached"), capi.memory_stats()), groups=new.stat("bucket_groups"), fused=new.stat("fused_records"), build_ms=b_ms,
                     served_batches=served[0]))
    free, total = torch.cuda.mem_get_info()
    print("\nrebuild beside a serving index, 8 GiB columns:", gens, f"device free {free / 2**30:.1f} of {total / 2**30:.1f} GiB at the end")
    # both generations answer; then the old one goes (database.cpp:280)
    r = old.query_batch_device(pa_blob.data_ptr(), pa_offs.data_ptr(), 20_000, pa_bytes)
    assert int(r.nrows) == want_rows
    old.close()
    pb_blob, pb_offs, pb_bytes = W.sample_patterns_torch(text_b, d_ds, 20_000, 6, 14, seed=6, miss_byte=0x7F)
    torch.cuda.synchronize()
    r = new.query_batch_device(pb_blob.data_ptr(), pb_offs.data_ptr(), 20_000, pb_bytes)
    assert int(r.nhits) >= 18_000
    new.close()


def test_c3_32gib_as_four_co_resident_shards_merged_on_one_gpu():
    """BASELINE.json configs[3] AS WORDED except for the xGMI hop: 2^25 docs x 1024 B of printable ASCII = 32 GiB, cut by
    byte range into four doc-aligned 8 GiB shards (index.h:61-65: suffixes never cross documents; index.cpp:317-321: a row
    belongs to one document), every shard with its own suffix array — all four CO-RESIDENT on one MI355X (packed 5-byte
    entries: ~44.5 GB of index per shard beside its 8 GiB of text).  The later shards build beside the resident ones, so
    their bucket groups adapt to what is free.  Global object ids are rank * ndocs + i; every shard answers the whole
    100 k-pattern batch; the per-shard results meet through cdb_comm (one host thread per rank, ranks sharing the device
    exchange by device copies) in BOTH forms: the counts-only merge and the full all-gatherv merge.  Checked: every
    shard's array by the GPU verifier, merged row_ptr = sum of the shards' counts, merged rows = shard rows in shard
    order, and brute-force scans over the whole 32 GiB for sampled patterns incl. short ones whose rows lie in all four
    shards."""
    import json
    import os
    import threading
    import torch
    from coffeedb_amd import capi, workloads as W
    free, total = torch.cuda.mem_get_info()
    if free < (270 << 30):
        pytest.skip("needs a whole 288 GB MI355X")
    G, nd, dl, npat = 4, 1 << 23, 1024, 100_000
    n = nd * dl
    ds = W.uniform_docs(nd, dl)
    d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
    capi.load_library().cdb_release_cached_memory()
    capi.memory_reset_peak()
    texts, shards, report = [], [], []
    blobs, offs, base = [], [], 0
    for r in range(G):
        text = W.random_bytes_torch(n, 12345, 0x20, 0x7E, stream=20 + r, device="cuda")
        # a quarter of the batch is drawn from every shard's documents (so every shard owns guaranteed hits)
        b_, o_, nb = W.sample_patterns_torch(text, d_ds, npat // G, 4, 16, seed=99 + r, miss_byte=0x7F)
        blobs.append(b_[:nb].clone())
        offs.append(o_[:-1] + base)
        base += nb
        del b_, o_
        d_ids = torch.arange(nd, dtype=torch.int64, device="cuda") + r * nd          # global object ids
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        g = capi.GpuStringIndex()
        g.build_resident(text.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), nd)
        assert (g.size, g.bits, g.sa_width) == (n, 24, 8) and g.stat("bucketed") == 1
        assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0
        in_use, peak, cached = capi.memory_stats()
        report.append({"shard": r, "build_ms": round(g.stat("build_ms"), 1), "bucket_groups": int(g.stat("bucket_groups")),
                       "fused_records": int(g.stat("fused_records")), "library_in_use_bytes": in_use, "library_peak_bytes": peak,
                       "device_free_bytes_after": int(torch.cuda.mem_get_info()[0])})
        capi.load_library().cdb_release_cached_memory()       # the next shard sizes its groups from what is really free
        texts.append((text, d_ids))
        shards.append(g)
    for r, g in enumerate(shards):
        v = g.verify()
        what = (r, v, report[r], {k: g.stat(k) for k in ("key_symbols", "partial_levels", "vl_key_bits", "unresolved_after_initial", "rounds", "sweep_records", "segmented", "fused_records", "bucket_groups", "gen_prebased", "sort_passes")})
        assert v["inversions"] == 0 and v["tie_violations"] == 0 and v["invalid_entries"] == 0, what
        assert v["entry_sum"] == v["expected_entry_sum"], what
    d_blob = torch.cat(blobs + [torch.zeros(16, dtype=torch.uint8, device="cuda")])
    d_offs = torch.cat(offs + [torch.tensor([base], dtype=torch.int64, device="cuda")])
    del blobs, offs
    torch.cuda.synchronize()
    local = [g.query_batch_device(d_blob.data_ptr(), d_offs.data_ptr(), npat, base) for g in shards]
    comms = capi.ShardComm.group([0] * G)
    assert all(c.world == G and c.transport == "device copies" for c in comms)

    def collective(fn):
        out, err = [None] * G, []

        def run(r):
            try:
                out[r] = fn(r)
            except Exception as e:  # noqa: BLE001
                err.append(repr(e))
        th = [threading.Thread(target=run, args=(r,)) for r in range(G)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not err, err[:2]
        return out

    def dev(ptr, cnt):
        if cnt == 0:
            return torch.zeros(0, dtype=torch.int64, device="cuda")
        return torch.as_tensor(_Dev(ptr, cnt), device="cuda")

    lrp = [dev(r.d_row_ptr, npat + 1) for r in local]
    lrows = [int(r.nrows) for r in local]
    assert all(int(r.nhits) >= int(r.nrows) > 0 for r in local)
    want_rp = torch.zeros(npat + 1, dtype=torch.int64, device="cuda")
    for p in lrp:
        want_rp += p                                           # merged row_ptr = sum of the shards' row_ptrs
    # ---- counts-only merge: merged row_ptr + every rank's row bases
    slices = collective(lambda r: comms[r].merge_counts(local[r]))
    for r, sl in enumerate(slices):
        assert int(sl.nrows_total) == sum(lrows) and int(sl.nrows_local) == lrows[r]
        assert torch.equal(dev(sl.d_row_ptr, npat + 1), want_rp)
        want_base = want_rp[:-1].clone()
        for q in range(r):
            want_base += lrp[q][1:] - lrp[q][:-1]
        assert torch.equal(dev(sl.d_row_base, npat), want_base)
    # ---- full merge: every rank holds the merged CSR (shard rows in shard order = ascending global document)
    merged = collective(lambda r: comms[r].merge(local[r]))
    total_rows = sum(lrows)
    m0 = merged[0]
    assert int(m0.nrows) == total_rows and torch.equal(dev(m0.d_row_ptr, npat + 1), want_rp)
    m_ids, m_cnt = dev(m0.d_ids, total_rows), dev(m0.d_counts, total_rows)
    for r in (1, G - 1):
        assert torch.equal(dev(merged[r].d_ids, total_rows), m_ids) and torch.equal(dev(merged[r].d_counts, total_rows), m_cnt)
    assert int(m_cnt.sum()) == sum(int(r.nhits) for r in local)
    rows = (want_rp[1:] - want_rp[:-1])
    inner = torch.ones(total_rows, dtype=torch.bool, device="cuda")
    inner[want_rp[:-1][rows > 0]] = False
    assert bool(((m_ids[1:] - m_ids[:-1])[inner[1:]] > 0).all())     # ids ascend inside every pattern, across the shard seams
    assert float((rows > 0).float().mean()) > 0.88
    # ---- brute force over the whole 32 GiB
    ho = d_offs.cpu().numpy()
    hb = d_blob.cpu().numpy()
    lens = np.diff(ho)
    rng = np.random.default_rng(8)
    short = np.nonzero(lens == 4)[0]
    picks = rng.choice(npat, 16, replace=False).tolist() + rng.choice(short, 5, replace=False).tolist() + [int(np.argmax(rows.cpu().numpy()))]
    straddle = 0
    hrp = want_rp.cpu().numpy()
    for j in picks:
        kw = hb[int(ho[j]):int(ho[j + 1])]
        d_kw = torch.from_numpy(kw.copy()).cuda()
        wd, wc = [], []
        for r in range(G):
            pos = _scan_occurrences(torch, texts[r][0], d_kw)
            pos = pos[(pos % dl) + len(kw) <= dl]
            d, c = torch.unique(pos // dl, return_counts=True)
            wd.append(d + r * nd)
            wc.append(c)
        straddle += int(sum(1 for d in wd if d.numel()) > 1)
        a, b = int(hrp[j]), int(hrp[j + 1])
        assert torch.equal(m_ids[a:b], torch.cat(wd)) and torch.equal(m_cnt[a:b], torch.cat(wc)), (j, bytes(kw))
    assert straddle >= 5                                              # patterns whose rows lie on both sides of a shard boundary
    free_end = int(torch.cuda.mem_get_info()[0])
    summary = {"config": "C3: 2^25 docs x 1024 B = 32 GiB as 4 doc-aligned shards on ONE MI355X", "shards": report,
               "merged_rows": total_rows, "patterns": npat, "patterns_checked_by_brute_force": len(picks), "straddling": straddle,
               "device_total_bytes": int(total), "device_free_bytes_all_resident": free_end, "transport": comms[0].transport}
    print("\n" + json.dumps(summary))
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/c3_one_gpu.json", "w") as f:
            json.dump(summary, f, indent=1)
    except OSError:
        pass
    for c in comms:
        c.close()
    for g in shards:
        g.close()
    capi.load_library().cdb_release_cached_memory()


def test_rebuild_16gib_utf8_shard_beside_the_serving_index_and_with_8gb_held():
    """database.cpp:276-280 at C4's per-GPU share: the serving index of a 16 GiB UTF-8 shard keeps answering while the next
    generation is built beside it.  Two packed generations (5-byte entries) + both texts are ~212 GB; the build sizes its bucket
    groups from what is left (records come from one text sweep per group, so no partitioned entry array exists).  The FIRST
    build runs with 8 GB of device memory held by somebody else — the torch context + RCCL buffers a rank carries at N = 8 —
    and must complete as well.  Memory figures go to stdout and gpurun_out/rebuild_16g.json."""
    import json
    import os
    import threading
    import torch
    from coffeedb_amd import capi, workloads as W
    free, total = torch.cuda.mem_get_info()
    if free < (270 << 30):
        pytest.skip("needs a whole 288 GB MI355X")
    capi.load_library().cdb_release_cached_memory()
    capi.memory_reset_peak()
    report = {}
    held = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")          # (somebody else's 8 GB)
    text_a, ds_a = W.utf8_bytes_torch(16 << 30, seed=5, device="cuda")
    nd_a = len(ds_a) - 1
    d_ds_a = torch.from_numpy(ds_a.astype(np.int64)).cuda()
    d_ids_a = torch.arange(nd_a, dtype=torch.int64, device="cuda")
    pa_blob, pa_offs, pa_bytes = W.sample_patterns_torch(text_a, d_ds_a, 20_000, 6, 14, seed=5, miss_byte=0xFF, utf8=True)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    old = capi.GpuStringIndex()
    old.build_resident(text_a.data_ptr(), d_ds_a.data_ptr(), d_ids_a.data_ptr(), nd_a)
    assert old.sa_width == 8 and old.stat("self_check_fallbacks") == 0 and old.stat("group_fallbacks") == 0
    v = old.verify()
    assert v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"] and v["tie_violations"] == 0
    report["first_build_with_8GB_held"] = dict(zip(("in_use", "peak", "cached"), capi.memory_stats()), groups=int(old.stat("bucket_groups")),
                                               build_ms=round(old.stat("build_ms"), 1), device_free=int(torch.cuda.mem_get_info()[0]))
    del held
    want = old.query_batch_device(pa_blob.data_ptr(), pa_offs.data_ptr(), 20_000, pa_bytes)
    want_rows, want_hits = int(want.nrows), int(want.nhits)
    capi.load_library().cdb_release_cached_memory()
    torch.cuda.empty_cache()
    capi.memory_reset_peak()

    stop, errors, served = threading.Event(), [], [0]

    def serve():
        try:
            while not stop.is_set():
                r = old.query_batch_device(pa_blob.data_ptr(), pa_offs.data_ptr(), 20_000, pa_bytes)
                assert (int(r.nrows), int(r.nhits)) == (want_rows, want_hits)
                served[0] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = threading.Thread(target=serve)
    th.start()
    outcome = "built"
    try:
        text_b, ds_b = W.utf8_bytes_torch(16 << 30, seed=6, device="cuda")   # the next generation of the column
        nd_b = len(ds_b) - 1
        d_ds_b = torch.from_numpy(ds_b.astype(np.int64)).cuda()
        d_ids_b = torch.arange(nd_b, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        report["device_free_before_rebuild"] = int(torch.cuda.mem_get_info()[0])
        new = capi.GpuStringIndex()
        try:
            new.build_resident(text_b.data_ptr(), d_ds_b.data_ptr(), d_ids_b.data_ptr(), nd_b)
        except RuntimeError as e:
            outcome = "cannot: " + str(e)[:300]
    finally:
        stop.set()
        th.join()
    assert not errors, errors[:2]
    report["outcome"] = outcome
    report["served_batches_meanwhile"] = served[0]
    if outcome == "built":
        assert new.stat("self_check_fallbacks") == 0 and new.stat("group_fallbacks") == 0
        v = new.verify()
        assert v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"] and v["tie_violations"] == 0
        r = new.verify_reference()
        assert r["violations"] == 0, r
        report["rebuild"] = dict(zip(("in_use", "peak", "cached"), capi.memory_stats()), groups=int(new.stat("bucket_groups")),
                                 build_ms=round(new.stat("build_ms"), 1), device_free=int(torch.cuda.mem_get_info()[0]))
        r = old.query_batch_device(pa_blob.data_ptr(), pa_offs.data_ptr(), 20_000, pa_bytes)      # both generations answer
        assert int(r.nrows) == want_rows
        new.close()
    print("\n" + json.dumps(report))
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/rebuild_16g.json", "w") as f:
            json.dump(report, f, indent=1)
    except OSError:
        pass
    old.close()
    capi.load_library().cdb_release_cached_memory()
    assert outcome == "built", outcome


def test_c4_shape_eight_ranks_on_one_gpu_ten_million_patterns():
    """Dress rehearsal of BASELINE.json configs[4]'s 8-rank code path on ONE device: the first 8-GPU run is the
    driver's, so everything but the xGMI hop runs here.  8 GiB of printable ASCII = 8 doc-aligned shards of 1 GiB (index.h:61-65:
    suffixes never cross documents), one suffix array per shard, eight ranks of cdb_comm_create_group on device 0 (one host thread
    each), 10^7 patterns answered by every shard, the counts-only merge (8 x 10^7 u32 row counts all-gathered, 64-bit row bases) —
    and the merged rows compared, ROW FOR ROW, with ONE index over the whole 8 GiB (index.cpp:317-321: rows ascend by document, so
    the single index's rows of a pattern are the shards' rows in shard order)."""
    import threading
    import torch
    from coffeedb_amd import capi, workloads as W
    free, _total = torch.cuda.mem_get_info()
    if free < (250 << 30):
        pytest.skip("needs a whole 288 GB MI355X")
    G, nd, dl, npat = 8, 1 << 20, 1024, 10_000_000
    n = nd * dl
    text = W.random_bytes_torch(G * n, 4321, 0x20, 0x7E, device="cuda")
    ds_all = W.uniform_docs(G * nd, dl)
    d_ds_all = torch.from_numpy(ds_all.astype(np.int64)).cuda()
    d_blob, d_offs, nb = W.sample_patterns_torch(text, d_ds_all, npat, 4, 16, seed=17, miss_byte=0x7F)
    torch.cuda.synchronize()

    def dev(ptr, cnt):
        if cnt == 0:
            return torch.zeros(0, dtype=torch.int64, device="cuda")
        return torch.as_tensor(_Dev(ptr, cnt), device="cuda")

    # ---- the referee: one index over the whole corpus (8-byte entries), its rows copied out, then released
    d_ids_all = torch.arange(G * nd, dtype=torch.int64, device="cuda")
    one = capi.GpuStringIndex()
    one.build_resident(text.data_ptr(), d_ds_all.data_ptr(), d_ids_all.data_ptr(), G * nd)
    assert (one.size, one.sa_width) == (G * n, 8)
    r1 = one.query_batch_device(d_blob.data_ptr(), d_offs.data_ptr(), npat, nb)
    total_rows = int(r1.nrows)
    want_rp = dev(r1.d_row_ptr, npat + 1).clone()
    want_ids = dev(r1.d_ids, total_rows).clone()
    want_cnt = dev(r1.d_counts, total_rows).clone()
    assert total_rows > npat and int(want_cnt.sum()) == int(r1.nhits)
    assert one.proof_wait(120_000) == 2                              # (the order proof behind the build: every pair)
    one.close()
    del d_ids_all
    torch.cuda.empty_cache()
    capi.load_library().cdb_release_cached_memory()
    # ---- eight shards of the same text, global object ids
    d_ds = torch.from_numpy(W.uniform_docs(nd, dl).astype(np.int64)).cuda()
    shards, keep = [], []
    for r in range(G):
        d_ids = torch.arange(nd, dtype=torch.int64, device="cuda") + r * nd
        g = capi.GpuStringIndex()
        g.build_resident(text[r * n:].data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), nd)
        assert (g.size, g.sa_width) == (n, 4) and g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0
        shards.append(g)
        keep.append(d_ids)
    local = [g.query_batch_device(d_blob.data_ptr(), d_offs.data_ptr(), npat, nb) for g in shards]
    comms = capi.ShardComm.group([0] * G)
    assert all(c.world == G for c in comms)
    out, err = [None] * G, []

    def run(r):
        try:
            out[r] = comms[r].merge_counts(local[r])
        except Exception as e:  # noqa: BLE001
            err.append(repr(e))
    th = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not err, err[:2]
    assert sum(int(r.nrows) for r in local) == total_rows
    pat_of = None
    for r in range(G):
        sl, lr = out[r], local[r]
        assert int(sl.nrows_total) == total_rows and int(sl.nrows_local) == int(lr.nrows)
        assert torch.equal(dev(sl.d_row_ptr, npat + 1), want_rp)     # merged row_ptr == the single index's
        k = int(lr.nrows)
        lrp = dev(lr.d_row_ptr, npat + 1)
        base = dev(sl.d_row_base, npat)
        cnt = lrp[1:] - lrp[:-1]
        pat_of = torch.repeat_interleave(torch.arange(npat, device="cuda"), cnt)           # pattern of every local row
        dest = base[pat_of] + (torch.arange(k, device="cuda") - lrp[:-1][pat_of])          # its slot in the merged CSR
        assert torch.equal(want_ids[dest], dev(lr.d_ids, k)) and torch.equal(want_cnt[dest], dev(lr.d_counts, k)), r
        assert int(dev(lr.d_ids, k).min()) >= r * nd and int(dev(lr.d_ids, k).max()) < (r + 1) * nd
        del lrp, base, cnt, dest
    for g in shards:
        assert g.proof_wait(60_000) == 2
    for c in comms:
        c.close()
    for g in shards:
        g.close()
    capi.load_library().cdb_release_cached_memory()
