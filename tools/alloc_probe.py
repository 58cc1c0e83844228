import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd, dl = 1 << 20, 1024
text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
ds = W.uniform_docs(nd, dl); ids = np.arange(nd, dtype=np.int64)
g = capi.GpuStringIndex()
for i in range(6):
    t = time.time(); g.build_device(text.data_ptr(), ds, ids); w = time.time() - t
    print(f"build {i}: wall {w*1e3:.1f} ms lib {g.stat('build_ms'):.1f} alloc {g.stat('alloc_ms'):.1f} free {g.stat('free_ms'):.1f}", flush=True)
print(torch.cuda.mem_get_info())
