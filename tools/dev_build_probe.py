import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd = int(sys.argv[1]); dl = int(sys.argv[2]); variant = int(sys.argv[3]); fuse = int(sys.argv[4])
text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
ds = W.uniform_docs(nd, dl); ids = np.arange(nd, dtype=np.int64)
torch.cuda.synchronize()
g = capi.GpuStringIndex(); g.set_option("sort_variant", variant); g.set_option("fuse_keygen", fuse)
if len(sys.argv) > 5: g.set_option("initial_passes", int(sys.argv[5]))
for i in range(3):
    g.build_device(text.data_ptr(), ds, ids)
    print(nd, dl, variant, fuse, "build", i, "ok", g.stat("build_ms"), flush=True)
