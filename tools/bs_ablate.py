"""Timing breakdown of the hybrid sort's bucket kernel on C1 (CDB_BS_ABLATE switches phases off; results are wrong then)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np, torch
    from coffeedb_amd import capi, workloads as W
    nd, dl = 1 << 20, 1024
    text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
    ds = torch.from_numpy(W.uniform_docs(nd, dl).astype(np.int64)).cuda(); ids = torch.arange(nd, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    g = capi.GpuStringIndex(); g.set_option("profile", 1)
    for i in range(3):
        if i == 1: g.profile_reset()
        try:
            g.build_resident(text.data_ptr(), ds.data_ptr(), ids.data_ptr(), nd)
        except RuntimeError as e:
            print("   build error:", e)
    p = g.profile()
    print(f"ablate={os.environ.get('CDB_BS_ABLATE', '0'):>2s}: sa_bucket_sort {p['sa_bucket_sort']['ms'] / p['sa_bucket_sort']['launches']:.2f} ms  build {g.stat('build_ms'):.1f} ms hybrid={g.stat('hybrid')}")
else:
    for a in (0, 1, 2, 4, 8, 3, 7, 15):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, CDB_BS_ABLATE=str(a)))
