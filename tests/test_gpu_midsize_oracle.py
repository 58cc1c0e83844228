"""Literal oracle parity on the bucket-wise (>= 2^32-style) build path at a size with REAL tiles.

The toy-size parity tests take the generic phases of the sweeps and the full-size tests have no oracle: round 5 shipped wrong
suffix arrays for half a day in exactly that gap (tiles of which a bucket group kept 56-99 %).  Here: 256 MiB of valid UTF-8 and
256 MiB of Zipf-64 text in documents of ~1 KiB (plus one long document, so that entries are 8 bytes wide and the packed /
segmented / swept forms run as they do at 4-16 GiB), `force_big_path = 1`, bucket groups capped so that the build runs in 2 and
3+ groups of uneven shares, default key forms AND the other one (Zipf: variable-length keys off; UTF-8: no partial symbol), reference_compat = 1:

    cdb_sa_copy == oracle array (ties canonicalised), element for element          (index.h:66-73, index.cpp:86-126)
    cdb_query_batch rows == oracle rows for 10 000 patterns                            (index.cpp:237-326)

The oracle (oracle/cpu_ref.cpp) builds each corpus ONCE (~25 s on the GPU box's 32 threads); every GPU build is compared with it.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_BYTES = int(os.environ.get("CDB_MIDSIZE_BYTES", str(256 << 20)))
N_PATTERNS = 10_000


def _make(kind):
    import torch
    from coffeedb_amd import workloads as W
    if kind == "utf8":
        text, ds = W.utf8_bytes_torch(N_BYTES, seed=41, device="cuda")
        blob = text.cpu().numpy()
        del text
    else:
        blob = W.zipf_bytes_torch(N_BYTES, seed=43, device="cuda").cpu().numpy()
        ds = W.uniform_docs(N_BYTES // 1024, 1024)
    torch.cuda.empty_cache()
    # one long document (64 consecutive ones merged: ~64 KiB -> 17 offset bits; with 2^18 documents the entries need 35 bits)
    mid = len(ds) // 3
    ds = np.concatenate([ds[:mid + 1], ds[mid + 64:]]).astype(np.uint64)
    return blob, ds


@pytest.fixture(scope="module", params=["utf8", "zipf"])
def corpus(request):
    """(kind, blob, doc_start, ids, oracle array, patterns, oracle rows) — the oracle runs once per kind."""
    from coffeedb_amd import workloads as W
    from oracle import OracleIndex
    kind = request.param
    blob, ds = _make(kind)
    nd = len(ds) - 1
    ids = (np.arange(nd, dtype=np.int64) * 5 + 3)[::-1].copy()      # descending ids: rows are in DOCUMENT order, not id order
    threads = min(32, os.cpu_count() or 1)
    o = OracleIndex()
    o.add_bulk(ids, blob, ds)
    o.build(threads)
    o.canonicalize(threads)
    assert o.sa_width == 8
    pb, po = W.sample_patterns(blob, ds, N_PATTERNS, 2, 14, seed=7, miss_frac=0.1, miss_byte=0xFF if kind == "utf8" else 0x7F)
    rows = o.query_batch(pb, po, nthreads=threads)
    osa = o.sa_view().copy()
    meta = (o.size, o.bits, o.mask, o.sa_width)
    del o
    yield kind, blob, ds, ids, osa, meta, pb, po, rows
    del osa


# share of all suffixes a bucket group may hold -> 2 groups (one large, one small), 2 uneven, 3 or more
@pytest.mark.parametrize("share,min_groups", [(0.90, 2), (0.62, 2), (0.37, 3)])
@pytest.mark.parametrize("other_keys", [False, True])
def test_bucket_wise_build_equals_the_oracle_at_256_mib(corpus, share, min_groups, other_keys):
    from coffeedb_amd import capi
    kind, blob, ds, ids, osa, meta, pb, po, rows = corpus
    n = int(ds[-1])
    opts = {"force_big_path": 1, "bucket_group_limit": int(n * share)}
    # Zipf-64: at 8 GiB (BASELINE config 2) the cost model picks 40-bit variable-length keys; at 256 MiB it would not, so the
    # form is forced here — and, as the other form, the dense keys.  UTF-8 (some 190 byte values: the code stream form stops at 127)
    # runs on dense keys with a partial next symbol in the leftover bits — and, as the other form, without it
    if kind == "zipf":
        opts["vl_keys"] = 0 if other_keys else 40
    elif other_keys:
        opts["partial_symbol"] = 0
    g = capi.GpuStringIndex()
    try:
        for k, v in opts.items():
            g.set_option(k, v)
        g.add_bulk(ids, blob, ds)
        g.build()
        info = (kind, opts, {k: g.stat(k) for k in ("bucket_groups", "vl_key_bits", "partial_levels", "sweep_records", "segmented", "fused_records", "key_symbols", "alphabet", "bucket_low_digits", "rounds", "unresolved_after_initial")})
        print(info)
        assert g.stat("bucketed") == 1 and g.sa_width == 8, info
        assert g.stat("bucket_groups") >= min_groups or g.stat("sweep_records") == 0, info
        assert g.stat("self_check_fallbacks") == 0 and g.stat("group_fallbacks") == 0, info    # (a fallback would mask a wrong array)
        assert g.stat("dense_key_retries") == 0, info           # (the key form chosen must be one the records path can take: no second prologue)
        if kind == "zipf":
            assert g.stat("vl_key_bits") == (0 if other_keys else 40), info
            assert other_keys or g.stat("sweep_records") == 1, info
        elif other_keys:
            assert g.stat("partial_levels") == 0, info
        assert (g.size, g.bits, g.mask, g.sa_width) == meta, info
        gsa = g.sa()
        assert gsa.dtype == osa.dtype and gsa.shape == osa.shape, info
        if not np.array_equal(gsa, osa):
            bad = np.flatnonzero(gsa != osa)
            raise AssertionError((info, "entries differ", int(bad.size), "first at", int(bad[0]), "last at", int(bad[-1])))
        del gsa
        got = g.query_batch(pb, po)
        assert got[3] == rows[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], rows[:3])), info
    finally:
        g.close()
        capi.load_library().cdb_release_cached_memory()
