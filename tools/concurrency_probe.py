"""Single-keyword cdb_query from T host threads at once (the reference's serving pattern): queries/s."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coffeedb_amd import capi, workloads as W
nd, dl = 1 << 16, 1024
blob, ds = W.ascii_corpus(nd, dl, seed=12345)
g = capi.GpuStringIndex(); g.add_bulk(np.arange(nd, dtype=np.int64), blob, ds); g.build()
pb, po = W.sample_patterns(blob, ds, 4096, 4, 16, seed=9)
kws = [bytes(pb[int(po[j]):int(po[j + 1])]) for j in range(4096)]
for coalesce in (1, 0):
    g.set_option("coalesce_queries", coalesce)
    for T in (1, 4, 16, 64):
        per = 4096 // T if T > 1 else 512
        def run(t):
            for j in range(per): g.query(kws[(t * per + j) % 4096])
        th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
        t0 = time.time(); [x.start() for x in th]; [x.join() for x in th]; dt = time.time() - t0
        print(f"coalesce={coalesce} threads={T:3d}: {T*per/dt/1e3:8.1f} k queries/s ({dt/(T*per)*1e6:.1f} us per query overall)", flush=True)
