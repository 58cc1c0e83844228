"""Latency of one cdb_query by hit-list size (one wavefront <= 64 hits, LDS sort <= 4096, batched path beyond)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd, dl = 1 << 18, 1024
text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
ds = W.uniform_docs(nd, dl); ids = np.arange(nd, dtype=np.int64)
torch.cuda.synchronize()
g = capi.GpuStringIndex(); g.build_device(text.data_ptr(), ds, ids)
host = text[:4096].cpu().numpy().tobytes()
for single in (1, 0):
    g.set_option("single_query", single)
    for m in (2, 3, 4, 8, 16):
        kw = host[100:100 + m]
        g.query(kw)
        t = time.perf_counter()
        for _ in range(200):
            r = g.query(kw)
        dt = (time.perf_counter() - t) / 200
        print(f"single_query={single} keyword of {m:2d} bytes: {sum(c for _, c in r):7d} hits in {len(r):7d} rows: {dt*1e6:7.1f} us (library {g.stat('query_ms')*1e3:.1f} us)")
