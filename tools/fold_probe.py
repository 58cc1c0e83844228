"""Which configuration of the folded reference order fails its spot check?  Builds the corpora of
tests/test_gpu_parity.py::test_reference_order_folded_into_the_bucket_wise_build with the self-check off (nothing masks a
wrong array) and compares with the oracle."""
import sys
import numpy as np
sys.path.insert(0, ".")
from coffeedb_amd import capi, workloads as W
from oracle import OracleIndex
lens = np.full(40000, 4, dtype=np.uint64); lens[40000 // 3] = 70000
ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
n = int(ds[-1])
ids = np.arange(40000, dtype=np.int64) * 3 + 1
if len(sys.argv) < 2:
    for seed, syms in ((5, [0x41, 0x42, 0xC3, 0xA9]), (6, [0x10, 0x7F, 0x80, 0xF0, 0x41]), (7, [0xC3, 0xA9, 0xE2]), (8, list(range(0x60, 0xA0)))):
        blob = np.asarray(syms, dtype=np.uint8)[W.random_bytes(n, seed, 0, len(syms) - 1)]
        o = OracleIndex(); o.add_bulk(ids, blob, ds); o.build(2); o.canonicalize()
        osa = o.sa()
        for opts in (dict(), dict(fold_depth1=0), dict(fold_root=0), dict(segmented_sort=0)):
            for sc in (0, 1):
                g = capi.GpuStringIndex()
                g.set_option("force_big_path", 1); g.set_option("self_check", sc)
                for k, v in opts.items():
                    g.set_option(k, v)
                g.add_bulk(ids, blob, ds)
                g.build()
                bad = int((g.sa() != osa).sum())
                print([hex(x) for x in syms][:5], opts, "self_check", sc, "mismatches", bad, "fallbacks", g.stat("self_check_fallbacks"),
                      "segmented", g.stat("segmented"), "rot", g.stat("compat_rotations"), flush=True)
                g.close()

    sys.exit(0)

# details of the first failing configuration
syms = [0x41, 0x42, 0xC3, 0xA9]
blob = np.asarray(syms, dtype=np.uint8)[W.random_bytes(n, 5, 0, 3)]
o = OracleIndex(); o.add_bulk(ids, blob, ds); o.build(2); o.canonicalize()
osa = o.sa()
g = capi.GpuStringIndex()
g.set_option("force_big_path", 1); g.set_option("self_check", 0)
g.add_bulk(ids, blob, ds); g.build()
sa = g.sa()
bad = np.nonzero(sa != osa)[0]
print("mismatches", len(bad), "range", bad[:3], bad[-3:], "bits", g.bits)
def show(a, i):
    e = int(a[i]); d = e & int(g.mask); off = e >> g.bits
    return (d, off, bytes(blob[int(ds[d]) + off:int(ds[d + 1])][:6]).hex())
for i in list(bad[:6]) + list(bad[len(bad)//2:len(bad)//2+3]) + list(bad[-3:]):
    print(i, "gpu", show(sa, int(i)), "ora", show(osa, int(i)))
# is it a permutation problem or a local order problem?
print("same multiset in mismatch range:", np.array_equal(np.sort(sa[bad[0]:bad[-1]+1]), np.sort(osa[bad[0]:bad[-1]+1])))
