import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
n = 1 << 28
g_ = torch.Generator(device="cuda").manual_seed(4)
a = torch.randint(0x20, 0x7F, (n // 2,), device="cuda", generator=g_).to(torch.uint8)
b = torch.randint(0x20, 0x7F, (n // 2,), device="cuda", generator=g_).to(torch.uint8)
lead = torch.randint(0xC2, 0xE0, (n // 2,), device="cuda", generator=g_).to(torch.uint8)
cont = torch.randint(0x80, 0xC0, (n // 2,), device="cuda", generator=g_).to(torch.uint8)
two = torch.randint(0, 10, (n // 2,), device="cuda", generator=g_) >= 6
text = torch.stack([torch.where(two, lead, a), torch.where(two, cont, b)], 1).reshape(-1).contiguous()
nd = n // 1024
ds = W.uniform_docs(nd, 1024); ids = np.arange(nd, dtype=np.int64)
host = text[:4096].cpu().numpy().tobytes()
torch.cuda.synchronize()
for compat in (0, 1):
    g = capi.GpuStringIndex(); g.set_option("reference_compat", compat); g.build_device(text.data_ptr(), ds, ids)
    for m in (4, 8, 16):
        kw = host[100:100 + m]
        g.query(kw)
        t = time.perf_counter()
        for _ in range(200): r = g.query(kw)
        dt = (time.perf_counter() - t) / 200
        print(f"compat={compat} keyword {m:2d} B: {len(r)} rows {dt*1e6:.1f} us")
