"""SA build time by corpus size (printable ASCII, 1 KiB documents): where the small-column floor sits."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
dl = 1024
for lg in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20, 31, 2):
    nd = (1 << lg) // dl
    text = W.random_bytes_torch(nd * dl, 12345, 0x20, 0x7E, device="cuda")
    ds = W.uniform_docs(nd, dl)
    d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
    d_ids = torch.arange(nd, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    g = capi.GpuStringIndex()
    ms = []
    for i in range(6):
        t = time.perf_counter()
        g.build_resident(text.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), nd)
        ms.append((time.perf_counter() - t) * 1e3)
    g.set_option("profile", 1)
    g.build_resident(text.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), nd)
    p = g.profile()
    kern = sum(v["ms"] for v in p.values()); launches = sum(v["launches"] for v in p.values())
    print(json.dumps({"log2_n": lg, "build_ms": round(min(ms[1:]), 3), "GiB_per_s": round((1 << lg) / 2**30 / (min(ms[1:]) * 1e-3), 2),
                      "kernels_ms": round(kern, 3), "profiled_launch_groups": launches, "key_symbols": g.stat("key_symbols"), "passes": g.stat("sort_passes")}), flush=True)
    g.close()
