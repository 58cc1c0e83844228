"""PCIe-inclusive build: cdb_add_bulk (host staging) + cdb_build of the C1 corpus."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coffeedb_amd import capi, workloads as W
nd, dl = 1 << 20, 1024
blob, ds = W.ascii_corpus(nd, dl, seed=12345)
ids = np.arange(nd, dtype=np.int64)
g = capi.GpuStringIndex()
t = time.perf_counter(); g.add_bulk(ids, blob, ds); ta = time.perf_counter() - t
for rep in range(4):
    t = time.perf_counter(); g.build(); tb = time.perf_counter() - t
    print(f"rep {rep}: cdb_build from host staging {tb*1e3:.1f} ms = {nd*dl/2**30/tb:.2f} GiB/s (device part {g.stat('build_ms'):.1f} ms); add_bulk {ta*1e3:.0f} ms")
