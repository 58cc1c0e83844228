"""Seeded fuzz parity: random corpus shapes x build options against the CPU oracle (suffix array
bit-exact after tie canonicalisation, batched query rows, OR-merge and highlight spans); every build is also awaited through
its order proof (self_check = 3: every adjacent pair against the text, verify.hip)."""
import numpy as np
import pytest

from coffeedb_amd import workloads as W

pytestmark = pytest.mark.gpu


def _corpus(rng):
    kind = rng.integers(0, 6)
    nd = int(rng.integers(1, 4000))
    if kind == 0:      # uniform, random alphabet range (sometimes with bytes >= 0x80)
        lo = int(rng.integers(0, 200)); hi = int(min(255, lo + rng.integers(0, 120)))
        blob, ds = W.ragged_corpus(nd, int(rng.integers(1, 120)), seed=int(rng.integers(1 << 30)), lo=lo, hi=hi,
                                   empty_every=int(rng.integers(0, 9)))
    elif kind == 1:    # tiny alphabet -> deep repeats
        blob, ds = W.ragged_corpus(nd, int(rng.integers(1, 200)), seed=int(rng.integers(1 << 30)), lo=0x61,
                                   hi=0x61 + int(rng.integers(0, 3)))
    elif kind == 2:    # duplicated documents
        base, _ = W.ascii_corpus(1, int(rng.integers(1, 300)), seed=int(rng.integers(1 << 30)), lo=0x61, hi=0x66)
        reps = int(rng.integers(2, 40))
        blob = np.concatenate([base] * reps)
        ds = (np.arange(reps + 1) * len(base)).astype(np.uint64)
    elif kind == 3:    # zipf
        blob, ds = W.zipf_corpus(max(1, nd // 4), int(rng.integers(8, 300)), seed=int(rng.integers(1 << 30)))
    elif kind == 4:    # valid UTF-8
        blob, ds = W.utf8_corpus(max(1, nd // 20), int(rng.integers(10, 200)), seed=int(rng.integers(1 << 30)))
    else:              # one long document among short ones (u64 entries when long enough)
        lens = rng.integers(0, 6, size=nd).astype(np.uint64)
        lens[int(rng.integers(0, nd))] = int(rng.integers(1000, 90000))
        ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        blob = W.random_bytes(int(ds[-1]), int(rng.integers(1 << 30)), 0x41, 0x44)
    return blob, ds


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CDB_FUZZ_N", "40"))))
def test_fuzz_parity(seed):
    from coffeedb_amd import capi
    from oracle import OracleIndex
    rng = np.random.default_rng(1000 + seed)
    blob, ds = _corpus(rng)
    nd = len(ds) - 1
    if int(ds[-1]) == 0:
        pytest.skip("empty corpus drawn")
    ids = rng.permutation(nd).astype(np.int64) * 3 - 50
    opts = {}
    if rng.random() < 0.3: opts["force_doubling"] = 1
    if rng.random() < 0.3: opts["fuse_keygen"] = 0
    if rng.random() < 0.25:
        opts["force_big_path"] = 1
        if rng.random() < 0.5: opts["bucket_group_limit"] = int(rng.integers(1, 5000))
    if rng.random() < 0.3: opts["initial_passes"] = int(rng.integers(1, 8))
    elif rng.random() < 0.5: opts["key_coding"] = int(rng.choice([1, 2]))
    if rng.random() < 0.3: opts["sort_variant"] = int(rng.choice([1, 21, 26, 31, 36, 32]))
    if rng.random() < 0.2: opts["keep_keys"] = 0
    if rng.random() < 0.3: opts["narrow_keys"] = 0
    if rng.random() < 0.2: opts["fast_search"] = 0
    if rng.random() < 0.2: opts["wave_rows"] = 0
    rng2 = np.random.default_rng(50_000 + seed)   # (its own stream: the draws above keep their seeds' meaning)
    if rng2.random() < 0.35:                       # the lone-keyword path through the resident workgroup
        opts["resident_query"] = 1
    if rng2.random() < 0.35 and "initial_passes" not in opts:
        # the 16 Ki-tile sorts with a forced key width: keys of 33..40 bits take the MSD-first sort (pair form for 6 symbols)
        opts["sort_variant"] = int(rng2.choice([31, 33]))
        opts["key_coding"] = 2
        opts["key_symbols"] = int(rng2.integers(3, 14))
        opts.pop("fuse_keygen", None)
        if rng2.random() < 0.3: opts["msd_first"] = 0
        if rng2.random() < 0.3: opts["msd_pair"] = 0
    if rng2.random() < 0.2: opts["key_directory"] = 0
    if rng2.random() < 0.2: opts["overlap_paircount"] = 0
    o = OracleIndex(); o.add_bulk(ids, blob, ds); o.build(2); o.canonicalize()
    g = capi.GpuStringIndex()
    for k, v in opts.items():
        g.set_option(k, v)
    g.add_bulk(ids, blob, ds)
    g.build()   # (no configuration is refused any more: the bucket-wise path takes all 256 byte values too)
    assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0, opts   # (a fallback would mask a wrong array)
    assert (g.size, g.bits, g.mask, g.sa_width) == (o.size, o.bits, o.mask, o.sa_width), (seed, opts)
    assert np.array_equal(g.sa(), o.sa()), (seed, opts)
    npat = int(rng.integers(1, 400))
    pb, po = W.sample_patterns(blob, ds, npat, 1, int(rng.integers(1, 24)), seed=seed, miss_frac=0.2, miss_byte=int(blob[0]))
    rp, gi, gc, hits = g.query_batch(pb, po)
    orp, oi, oc, ohits = o.query_batch(pb, po, nthreads=2)
    assert hits == ohits and np.array_equal(rp, orp) and np.array_equal(gi, oi) and np.array_equal(gc, oc), (seed, opts)
    kws = [bytes(pb[int(po[j]):int(po[j + 1])]) for j in range(min(npat, 12))]
    if len(set(ids.tolist())) == nd:
        assert g.query_or(kws) == o.filter_or(kws), (seed, opts)
    assert g.query_spans(kws) == o.highlight_spans(kws, ids), (seed, opts)
    for kw in kws[:6]:   # a lone keyword takes the one-wavefront kernel (<= 4096 hits) or hands over to the batch path
        assert g.query(kw) == o.query(kw), (seed, opts, kw)
    assert g.proof_wait(60_000) == 2 and g.stat("self_check_fallbacks") == 0   # the order proof behind the build: every adjacent pair
    g.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CDB_FUZZ_SEG_N", "28"))))
def test_fuzz_segmented_bucket_wise_parity(seed):
    """The >= 2^32 code path with 8-byte entries (packed records, segmented passes, entries + flags from the last pass,
    reference order folded into the sort) on seeded corpora small enough for the oracle: tens of thousands of tiny
    documents and one long one make bits + offset bits exceed 32; alphabets from 2 symbols to all 256 byte values."""
    from coffeedb_amd import capi
    from oracle import OracleIndex
    rng = np.random.default_rng(7000 + seed)
    nd = int(rng.integers(33000, 60000))
    lens = rng.integers(0, 6, size=nd).astype(np.uint64)
    lens[int(rng.integers(0, nd))] = int(rng.integers(66000, 120000))
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(ds[-1])
    kind = int(rng.integers(0, 5))
    if kind == 0:      # a handful of symbols on both sides of 0x80: radix nodes several levels deep
        syms = sorted(set(int(x) for x in rng.choice([0x00, 0x10, 0x41, 0x42, 0x7F, 0x80, 0xA9, 0xC3, 0xE2, 0xFF], size=int(rng.integers(2, 6)))))
        blob = np.asarray(syms, dtype=np.uint8)[W.random_bytes(n, int(rng.integers(1 << 30)), 0, len(syms) - 1)]
    elif kind == 1:    # any byte range
        lo = int(rng.integers(0, 200)); hi = int(min(255, lo + rng.integers(1, 200)))
        blob = W.random_bytes(n, int(rng.integers(1 << 30)), lo, hi)
    elif kind == 2:    # skewed
        blob = W.zipf_corpus(1, n, seed=int(rng.integers(1 << 30)))[0][:n]
    elif kind == 3:    # all 256 byte values
        blob = W.random_bytes(n, int(rng.integers(1 << 30)), 0, 255)
    else:              # repeats: equal suffixes from different documents, groups that never resolve
        blob = W.random_bytes(n, int(rng.integers(1 << 30)), 0x61, 0x63)
        half = n // 3
        blob[half:2 * half] = blob[:half]
    ids = rng.permutation(nd).astype(np.int64) * 2 + 9
    opts = {"force_big_path": 1}
    if rng.random() < 0.5: opts["bucket_group_limit"] = int(rng.integers(1, 80000))
    if rng.random() < 0.25: opts["segmented_sort"] = 0
    if rng.random() < 0.3: opts["fuse_records"] = 0      # (records by partition + gather although they all fit at once)
    if rng.random() < 0.25: opts["fold_root"] = 0
    if rng.random() < 0.25: opts["fold_depth1"] = 0
    if rng.random() < 0.3: opts["plain_tile_order"] = 1
    if rng.random() < 0.3: opts["force_doubling"] = 1
    if rng.random() < 0.2: opts["reference_compat"] = 0
    if rng.random() < 0.3: opts["initial_passes"] = int(rng.integers(1, 8))
    if rng.random() < 0.3: opts["pack_sa"] = 0          # (plain 8-byte storage; default: packed 5-byte storage)
    # (drawn last, so that the seeds of earlier rounds keep their corpora and options) records by sweeps over the text — the default —
    # against the generated records pass / partition + gather; counted tile bases against the chained scan
    if rng.random() < 0.3: opts["sweep_records"] = 0
    if rng.random() < 0.2: opts["gen_prebased"] = 0
    # (round 5, its own stream: earlier seeds keep their draws) variable-length keys: forced at a drawn width, where the sweep form
    # applies; where it does not (255 / 256 symbols, sweep_records = 0, ...) the build must fall back to the dense keys by itself
    rng5 = np.random.default_rng(91000 + seed)
    if rng5.random() < 0.5: opts["vl_keys"] = int(rng5.choice([1, 16, 24, 32, 40, 48, 56]))
    elif rng5.random() < 0.3: opts["vl_keys"] = 0
    if rng5.random() < 0.25: opts["group_sort"] = 0      # (the general sort for the refinement rounds; default: one pass inside the groups)
    if rng5.random() < 0.25: opts["list_rounds"] = 0     # (every round compacts from the flag array; default: later rounds from the previous list)
    if rng5.random() < 0.25: opts["fuse_pairclass"] = 0  # (next-byte classes by their own sweep; default: counted beside the bytes)
    g = capi.GpuStringIndex()
    for k, v in opts.items():
        g.set_option(k, v)
    g.add_bulk(ids, blob, ds)
    g.build()
    assert g.sa_width == 8 and g.stat("bucketed") == 1, opts
    assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0, (seed, kind, opts)   # (a fallback would mask a wrong array)
    if opts.get("reference_compat", 1):
        o = OracleIndex(); o.add_bulk(ids, blob, ds); o.build(2); o.canonicalize()
        assert np.array_equal(g.sa(), o.sa()), (seed, kind, opts)
        pb, po = W.sample_patterns(blob, ds, 120, 1, 9, seed=seed, miss_frac=0.1, miss_byte=int(blob[0]) ^ 0x55)
        got, want = g.query_batch(pb, po), o.query_batch(pb, po)
        assert got[3] == want[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3])), (seed, kind, opts)
    else:              # plain unsigned order: a sorted permutation (the oracle restates the reference's order)
        v = g.verify()
        assert v["inversions"] == v["tie_violations"] == v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"], (seed, kind, opts)
    assert g.proof_wait(60_000) == 2 and g.stat("self_check_fallbacks") == 0   # the order proof behind the build: every adjacent pair
    g.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CDB_FUZZ_MID_N", "32"))))
def test_fuzz_bucket_wise_with_documents_of_real_size(seed):
    """The >= 2^32 code path on corpora whose documents are hundreds of bytes long — what the full-size configurations look like
    and what the tiny-document fuzz above never reaches: tiles whose document table fits the LDS take the FAST phase B of the
    sweeps (records_sweep.h), and with several bucket groups a sweep keeps an arbitrary share of every tile.  (Round 5 shipped a
    loop that was wrong for tiles with 56-99 % kept for half a day: only the 32 GiB test saw it.)  2-7 MB per corpus: seconds in
    the oracle."""
    from coffeedb_amd import capi
    from oracle import OracleIndex
    rng = np.random.default_rng(52000 + seed)
    nd = int(rng.integers(3000, 20000))
    mean = int(rng.integers(60, 900))
    lens = rng.integers(mean // 2, mean * 3 // 2 + 2, size=nd).astype(np.uint64)
    if rng.random() < 0.3: lens[rng.integers(0, nd, size=nd // 50)] = 0                       # some empty documents (generic path tiles)
    if rng.random() < 0.6: lens[int(rng.integers(0, nd))] = int(rng.integers(1 << 20, 1 << 21))   # (mostly) 8-byte entries
    while int(lens.sum()) > 7_000_000: lens = lens[: len(lens) * 3 // 4]
    nd = len(lens)
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(ds[-1])
    kind = int(rng.integers(0, 5))
    if kind == 0:
        lo = int(rng.integers(0x20, 0x60)); hi = int(min(0x7E, lo + rng.integers(1, 95)))
        blob = W.random_bytes(n, int(rng.integers(1 << 30)), lo, hi)
    elif kind == 1:
        blob = W.zipf_corpus(1, n, seed=int(rng.integers(1 << 30)), nsym=int(rng.integers(8, 90)))[0][:n]
    elif kind == 2:    # valid UTF-8 (bytes on both sides of 0x80: the reference's signed child order)
        blob = W.utf8_corpus(1, n + n // 8 + 64, seed=int(rng.integers(1 << 30)))[0][:n].copy()   # (about 1.7 bytes per code point: cut from a longer one)
    elif kind == 3:
        lo = int(rng.integers(0, 120)); hi = int(min(255, lo + rng.integers(20, 253)))
        blob = W.random_bytes(n, int(rng.integers(1 << 30)), lo, hi)
    else:              # long repeats: groups the key cannot resolve
        blob = W.random_bytes(n, int(rng.integers(1 << 30)), 0x61, 0x7A)
        q = n // 4
        blob[2 * q:3 * q] = blob[:q]
    ids = rng.permutation(nd).astype(np.int64) * 3 + 1
    opts = {"force_big_path": 1}
    if rng.random() < 0.8: opts["bucket_group_limit"] = max(1, int(n * rng.uniform(0.08, 1.0)))
    if rng.random() < 0.3: opts["vl_keys"] = int(rng.choice([0, 1, 24, 32, 40]))
    if rng.random() < 0.25: opts["partial_symbol"] = 0
    if rng.random() < 0.2: opts["pack_sa"] = 0
    if rng.random() < 0.15: opts["sweep_records"] = 0
    if rng.random() < 0.15: opts["plain_tile_order"] = 1
    if rng.random() < 0.15: opts["force_doubling"] = 1
    if rng.random() < 0.2: opts["initial_passes"] = int(rng.integers(2, 8))
    if rng.random() < 0.2: opts["group_sort"] = 0
    if rng.random() < 0.2: opts["list_rounds"] = 0
    if rng.random() < 0.2: opts["fuse_pairclass"] = 0
    g = capi.GpuStringIndex()
    for k, v in opts.items():
        g.set_option(k, v)
    g.add_bulk(ids, blob, ds)
    g.build()
    info = (seed, kind, nd, n, opts, g.sa_width, g.stat("bucket_groups"), g.stat("sweep_records"))
    assert g.stat("bucketed") == 1, info
    assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0, info
    o = OracleIndex(); o.add_bulk(ids, blob, ds); o.build(4); o.canonicalize()
    assert np.array_equal(g.sa(), o.sa()), info
    pb, po = W.sample_patterns(blob, ds, 100, 1, 9, seed=seed, miss_frac=0.1, miss_byte=int(blob[0]) ^ 0x55)
    got, want = g.query_batch(pb, po), o.query_batch(pb, po)
    assert got[3] == want[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3])), info
    assert g.proof_wait(60_000) == 2 and g.stat("self_check_fallbacks") == 0   # the order proof behind the build: every adjacent pair
    g.close()
