"""Repro harness for the segmented bucket-wise build (small inputs through force_big_path)."""
import sys, numpy as np
sys.path.insert(0, ".")
from coffeedb_amd import capi, workloads as W
from oracle import OracleIndex

def run(force_doubling, seg, group_limit=0, alphabet=(0x61, 0x63), seed=17):
    lens = np.full(40000, 3, dtype=np.uint64); lens[123] = 70000
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blob = W.random_bytes(int(ds[-1]), seed, *alphabet)
    ids = np.arange(len(ds) - 1, dtype=np.int64)
    g = capi.GpuStringIndex(device=0)
    for k, v in dict(force_big_path=1, force_doubling=force_doubling, segmented_sort=seg, bucket_group_limit=group_limit).items():
        g.set_option(k, v)
    g.add_bulk(ids, blob, ds)
    print("build fd=%d seg=%d gl=%d" % (force_doubling, seg, group_limit), flush=True)
    g.build()
    print("  stats", {k: g.stat(k) for k in ("segmented", "bucket_groups", "rounds", "unresolved_after_initial", "key_symbols", "bucket_low_digits")}, flush=True)
    o = OracleIndex(); o.add_bulk(ids, blob, ds); o.build(); o.canonicalize()
    print("  parity", bool(np.array_equal(g.sa(), o.sa())), g.verify(), flush=True)
    g.close()

if __name__ == "__main__":
    for args in [(0, 0), (0, 1), (1, 0), (1, 1), (1, 1, 30000), (0, 1, 30000)]:
        run(*args)
