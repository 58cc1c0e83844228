import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd, dl = 1 << 20, 1024
text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
ds = W.uniform_docs(nd, dl); ids = np.arange(nd, dtype=np.int64)
torch.cuda.synchronize()
g = capi.GpuStringIndex(); g.build_device(text.data_ptr(), ds, ids)
host = text[: 1 << 26].cpu().numpy()
for npat in (100000, 1000000):
    pb, po = W.sample_patterns(host, W.uniform_docs(1 << 16, dl), npat, 4, 16, seed=99)
    for rep in range(4):
        r = None
        t = time.perf_counter(); r = g.query_batch(pb, po); dt = time.perf_counter() - t
    print(f"npat={npat}: wall {dt*1e3:.2f} ms; library {g.stat('query_ms'):.2f} = upload {g.stat('query_upload_ms'):.2f} + device {g.stat('query_device_ms'):.2f} + download {g.stat('query_download_ms'):.2f}")
