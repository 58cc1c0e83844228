import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coffeedb_amd import capi, workloads as W
from oracle import OracleIndex
nd, dl, passes, variant = (int(x) for x in sys.argv[1:5])
blob, ds = W.ascii_corpus(nd, dl, seed=5)
ids = np.arange(nd, dtype=np.int64)
o = OracleIndex(); o.add_bulk(ids, blob, ds); o.build(); o.canonicalize()
g = capi.GpuStringIndex(); g.set_option("initial_passes", passes); g.set_option("sort_variant", variant)
g.add_bulk(ids, blob, ds); g.build()
print(nd, dl, passes, variant, "nsym", g.stat("key_symbols"), "fused", g.stat("fused_keygen"), "equal", np.array_equal(g.sa(), o.sa()), flush=True)
