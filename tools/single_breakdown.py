"""Where a lone cdb_query's microseconds go on C1: (a) a keyword with a byte the text never holds (answered on the host:
pure call overhead), (b) launched kernel, (c) resident workgroup — all timed inside the library."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from coffeedb_amd import capi, workloads as W
nd, dl = 1 << 20, 1024
text = W.random_bytes_torch(nd * dl, 12345, 0x20, 0x7E, device="cuda")
ds = W.uniform_docs(nd, dl)
ids = np.arange(nd, dtype=np.int64)
host = text[: 1 << 24].cpu().numpy()
torch.cuda.synchronize()
g = capi.GpuStringIndex()
g.build_device(text.data_ptr(), ds, ids)
kws = [bytes(host[p:p + 8]) for p in range(1000, 1000 + 97 * 64, 97)]
absent = [b"\x01" + k[1:] for k in kws]
long_kws = [bytes(host[p:p + 24]) for p in range(1000, 1000 + 97 * 64, 97)]
short = [bytes(host[p:p + 5]) for p in range(1000, 1000 + 97 * 64, 97)]
def med(x):
    x = np.sort(x); return round(float(x[len(x) // 2]), 2)
for name, ks in (("absent byte (host only)", absent), ("8-byte keywords", kws), ("24-byte keywords", long_kws), ("5-byte keywords (decisive keys)", short)):
    g.set_option("resident_query", 0)
    a = med(g.query_latency_us(ks, reps=32))
    g.set_option("resident_query", 1)
    for k in ks[:4]:
        g.query(k)
    b = med(g.query_latency_us(ks, reps=32))
    print(f"{name:36s} launched {a:6.2f} us   resident {b:6.2f} us")
g.close()
