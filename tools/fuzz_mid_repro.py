"""Rebuilds corpora of tests/test_gpu_fuzz.py::test_fuzz_bucket_wise_with_documents_of_real_size in ONE process, in the given order,
and compares the GPU's arrays with the oracle's.  usage: python tools/fuzz_mid_repro.py <seed>[:opt=val,...] ...
(options given replace the drawn ones; "seed:" alone = force_big_path only).  Test-side tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coffeedb_amd import capi, workloads as W
from oracle import OracleIndex


def corpus(seed):
    rng = np.random.default_rng(52000 + seed)
    nd = int(rng.integers(3000, 20000))
    mean = int(rng.integers(60, 900))
    lens = rng.integers(mean // 2, mean * 3 // 2 + 2, size=nd).astype(np.uint64)
    if rng.random() < 0.3: lens[rng.integers(0, nd, size=nd // 50)] = 0
    if rng.random() < 0.6: lens[int(rng.integers(0, nd))] = int(rng.integers(1 << 20, 1 << 21))
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
    elif kind == 2:
        blob = W.utf8_corpus(1, n + n // 8 + 64, seed=int(rng.integers(1 << 30)))[0][:n].copy()   # (about 1.7 bytes per code point: cut from a longer one)
    elif kind == 3:
        lo = int(rng.integers(0, 120)); hi = int(min(255, lo + rng.integers(20, 253)))
        blob = W.random_bytes(n, int(rng.integers(1 << 30)), lo, hi)
    else:
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
    return kind, ids, blob, ds, opts


cache = {}
for spec in sys.argv[1:]:
    seed, _, o_ = spec.partition(":")
    seed = int(seed)
    if seed not in cache:
        kind, ids, blob, ds, opts = corpus(seed)
        o = OracleIndex(); o.add_bulk(ids, blob, ds); o.build(4); o.canonicalize()
        cache[seed] = (kind, ids, blob, ds, opts, o.sa())
    kind, ids, blob, ds, opts, want = cache[seed]
    if _:
        opts = {"force_big_path": 1}
        for kv in filter(None, o_.split(",")):
            k, v = kv.split("="); opts[k] = int(v)
    g = capi.GpuStringIndex()
    for k, v in opts.items(): g.set_option(k, v)
    g.add_bulk(ids, blob, ds); g.build()
    got = g.sa()
    bad = np.flatnonzero(got != want)
    v = g.verify()
    print("seed", seed, "kind", kind, "nd", len(ds) - 1, "n", int(ds[-1]), opts, "width", g.sa_width, "groups", g.stat("bucket_groups"), "sweep", g.stat("sweep_records"),
          "vl", g.stat("vl_key_bits"), "partial", g.stat("partial_levels"), "unres", g.stat("unresolved_after_initial"), "rounds", g.stat("rounds"), "scfb", g.stat("self_check_fallbacks"),
          "MISMATCHES", len(bad), "inv", v["inversions"], "ties", v["tie_violations"], "invalid", v["invalid_entries"], "sum_ok", v["entry_sum"] == v["expected_entry_sum"])
    if len(bad):
        i = int(bad[0]); j = int(bad[-1])
        print("   first", i, "last", j, "got", [hex(int(x)) for x in got[i:i + 3]], "want", [hex(int(x)) for x in want[i:i + 3]])
    g.close()
