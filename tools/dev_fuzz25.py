import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coffeedb_amd import capi, workloads as W
from oracle import OracleIndex
sys.path.insert(0, "tests")
from test_gpu_fuzz import _corpus
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
blob, ds = _corpus(rng)
nd = len(ds) - 1
ids = rng.permutation(nd).astype(np.int64) * 3 - 50
print("nd", nd, "n", int(ds[-1]), "alphabet", len(np.unique(blob)), "maxlen", int((ds[1:]-ds[:-1]).max()))
o = OracleIndex(); o.add_bulk(ids, blob, ds); o.build(2); o.canonicalize()
for opts in ({}, {"force_big_path": 1}, {"force_big_path": 1, "force_doubling": 1}, {"wave_rows": 0}, {"fast_search": 0}):
    g = capi.GpuStringIndex()
    for k, v in opts.items(): g.set_option(k, v)
    g.add_bulk(ids, blob, ds); g.build()
    for _ in range(9): rng2 = None
    rng3 = np.random.default_rng(1000 + seed); _corpus(rng3); rng3.permutation(nd)
    for _ in range(8): rng3.random()
    npat = int(rng3.integers(1, 400)); mm = int(rng3.integers(1, 24))
    pb, po = W.sample_patterns(blob, ds, npat, 1, mm, seed=seed, miss_frac=0.2, miss_byte=int(blob[0]))
    kws = [bytes(pb[int(po[j]):int(po[j + 1])]) for j in range(min(npat, 12))]
    a = g.query_spans(kws); b = o.highlight_spans(kws, ids)
    print(opts, "width", g.sa_width, "bits", g.bits, "spans equal:", a == b, len(a), len(b), "sa equal", np.array_equal(g.sa(), o.sa()))
    if a != b:
        for kw in kws:
            x = g.query_spans([kw]); y = o.highlight_spans([kw], ids)
            if x != y: print("   kw", kw, len(x), len(y), x[:2], y[:2]); break
