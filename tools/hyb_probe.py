"""Hybrid sort probe on C1-shaped corpora: timing per kernel, correctness by the GPU verifier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dl = 1024
text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
ds = torch.from_numpy(W.uniform_docs(nd, dl).astype(np.int64)).cuda(); ids = torch.arange(nd, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
g = capi.GpuStringIndex(); g.set_option("profile", 1)
for i in range(3):
    if i == 1: g.profile_reset()
    g.build_resident(text.data_ptr(), ds.data_ptr(), ids.data_ptr(), nd)
p = g.profile()
print(f"hybrid={g.stat('hybrid')} passes_opt={os.environ.get('CDB_HYBRID_PASSES')} largest={g.stat('hybrid_largest_bucket')} est={g.stat('hybrid_estimate')} retries={g.stat('hybrid_retries')} build {g.stat('build_ms'):.1f} ms unresolved={g.stat('unresolved_after_initial')}")
for k, v in sorted(p.items(), key=lambda kv: -kv[1]["ms"])[:8]:
    print(f"   {k:36s} {v['ms'] / 2:8.2f} ms/build  x{v['launches'] // 2}")
v = g.verify(); print("  ", v)
