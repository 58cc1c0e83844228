"""Per-build kernel times over consecutive rebuilds of the C1 corpus (do buffer roles / addresses matter?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd, dl = 1 << 20, 1024
text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
ds = W.uniform_docs(nd, dl); ids = np.arange(nd, dtype=np.int64)
torch.cuda.synchronize()
g = capi.GpuStringIndex(); g.set_option("profile", 1)
for rep in range(8):
    g.profile_reset()
    g.build_device(text.data_ptr(), ds, ids)
    p = g.profile()
    keys = ["rs_onesweep_k32_v32_w8_t16384", "rs_onesweep_textgen_split_t16384", "sa_keyhist", "sa_initflags"]
    print(f"rep {rep}: build {g.stat('build_ms'):.2f} ms alloc {g.stat('alloc_ms'):.2f} | " + " ".join(f"{k.split('_t16384')[0][-14:]}={p[k]['ms']:.2f}" for k in keys if k in p))
