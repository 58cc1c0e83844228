import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd, dl = 1 << 20, 1024
text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
ds = W.uniform_docs(nd, dl); ids = np.arange(nd, dtype=np.int64)
torch.cuda.synchronize()
g = capi.GpuStringIndex(); g.build_device(text.data_ptr(), ds, ids)
host = text[:65536].cpu().numpy().tobytes()
kws = [host[p:p + 8] for p in range(1000, 1000 + 97 * 64, 97)]
print("C caller us:", np.sort(g.query_latency_us(kws, reps=32))[[6, 32, 57]])
g.set_option("resident_query", 1)
print("resident us:", np.sort(g.query_latency_us(kws, reps=32))[[6, 32, 57]], "directory cells", g.stat("key_directory_cells"))
g.set_option("key_directory", 0)
print("resident, no key directory us:", np.sort(g.query_latency_us(kws, reps=32))[[6, 32, 57]])
g.set_option("resident_query", 0)
print("launched, no key directory us:", np.sort(g.query_latency_us(kws, reps=32))[[6, 32, 57]])
