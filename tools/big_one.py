"""One HBM-resident build of a bench workload (default utf8_4g), kernel table from the library's HIP-event profiler.
usage: python tools/big_one.py [workload] [reps]   (options through CDB_OPTIONS / ablation env vars)"""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from coffeedb_amd import capi, workloads as W
name = sys.argv[1] if len(sys.argv) > 1 else "utf8_4g"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = bench.WORKLOADS[name]
dev = torch.device("cuda", 0)
text, ds, n = bench.make_corpus(torch, W, cfg, 0, dev)
nd = len(ds) - 1
d_ds = torch.from_numpy(ds.astype(np.int64)).to(dev)
d_ids = torch.arange(nd, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
g = capi.GpuStringIndex(device=0)
g.set_option("profile", 1)
ms = []
try:
    for i in range(reps + 1):
        if i == 1:
            g.profile_reset()
        t = time.perf_counter()
        g.build_resident(text.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), nd)
        ms.append(round((time.perf_counter() - t) * 1e3, 2))
    prof = g.profile()
    print(name, "build_ms", ms, "kernels_ms", round(sum(v["ms"] for v in prof.values()) / reps, 2))
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:12]:
        print(f"   {k:34s} {v['ms'] / reps:9.3f} ms  x{v['launches'] // reps:<5d} {v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f} GB/s")
    print("   stats", {k: g.stat(k) for k in ("segmented", "bucket_groups", "rounds", "unresolved_after_initial")})
except Exception as e:  # noqa: BLE001
    print(name, "FAILED", repr(e)[:300], ms)
    prof = g.profile()
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:8]:
        print(f"   {k:34s} {v['ms']:9.3f} ms  x{v['launches']:<5d}")
