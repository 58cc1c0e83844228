"""Re-runs one seed of tests/test_gpu_fuzz.py::test_fuzz_segmented_bucket_wise_parity with option variations and says where
the suffix array first differs from the oracle's."""
import sys
import numpy as np
sys.path.insert(0, ".")
from coffeedb_amd import capi, workloads as W
from oracle import OracleIndex
seed = int(sys.argv[1])
rng = np.random.default_rng(7000 + seed)
nd = int(rng.integers(33000, 60000))
lens = rng.integers(0, 6, size=nd).astype(np.uint64)
lens[int(rng.integers(0, nd))] = int(rng.integers(66000, 120000))
ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
n = int(ds[-1])
kind = int(rng.integers(0, 5))
assert kind == 0, kind
syms = sorted(set(int(x) for x in rng.choice([0x00, 0x10, 0x41, 0x42, 0x7F, 0x80, 0xA9, 0xC3, 0xE2, 0xFF], size=int(rng.integers(2, 6)))))
blob = np.asarray(syms, dtype=np.uint8)[W.random_bytes(n, int(rng.integers(1 << 30)), 0, len(syms) - 1)]
ids = rng.permutation(nd).astype(np.int64) * 2 + 9
print("syms", [hex(x) for x in syms], "n", n, "nd", nd)
o = OracleIndex(); o.add_bulk(ids, blob, ds); o.build(2); o.canonicalize()
osa = o.sa()
for opts in (dict(segmented_sort=0, bucket_group_limit=68338), dict(segmented_sort=0), dict(segmented_sort=0, fold_root=0),
             dict(segmented_sort=0, pack_entries=0), dict(bucket_group_limit=68338), dict()):
    g = capi.GpuStringIndex()
    g.set_option("force_big_path", 1)
    for k, v in opts.items():
        g.set_option(k, v)
    g.add_bulk(ids, blob, ds)
    g.build()
    sa = g.sa()
    bad = np.nonzero(sa != osa)[0]
    print(opts, "mismatches", len(bad), "first", bad[:5], {k: g.stat(k) for k in ("segmented", "bucket_groups", "root_folded", "compat_rotations", "key_symbols", "bucket_low_digits", "rounds")})
    if len(bad):
        i = int(bad[0]); bits = g.bits
        for name, a in (("gpu", sa), ("ora", osa)):
            e = int(a[i]); d = e & int(g.mask); off = e >> bits
            print("   ", name, "slot", i, "doc", d, "off", off, bytes(blob[int(ds[d]) + off:int(ds[d + 1])][:12]))
    g.close()
