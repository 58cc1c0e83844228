import sys, threading, time
sys.path.insert(0, "/root/repo")
import numpy as np
from coffeedb_amd import capi, workloads as W
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 8
corp = [W.ascii_corpus(16384, 1024, seed=70 + k) for k in range(NT)]
res = [None] * NT
def work(k):
    blob, ds = corp[k]
    g = capi.GpuStringIndex(); g.set_option("sort_variant", 31)
    g.add_bulk(np.arange(len(ds) - 1, dtype=np.int64), blob, ds)
    t = time.time()
    for _ in range(4):
        g.build()
    v = g.verify()
    res[k] = (time.time() - t, g.stat("group_fallbacks"), v["inversions"], v["invalid_entries"], v["entry_sum"] == v["expected_entry_sum"])
ths = [threading.Thread(target=work, args=(k,)) for k in range(NT)]
t0 = time.time()
for t in ths: t.start()
for t in ths: t.join()
print("wall", round(time.time() - t0, 2), res)
