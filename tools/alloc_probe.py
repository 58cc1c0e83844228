"""Where a big build's wall time goes besides its kernels: allocation / release times and the pool's state, build after build.
usage: alloc_probe.py <workload> [builds]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from coffeedb_amd import capi, workloads as W
name = sys.argv[1] if len(sys.argv) > 1 else "c4shard"
builds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = bench.WORKLOADS[name]
dev = torch.device("cuda", 0)
text, ds, n = bench.make_corpus(torch, W, cfg, 0, dev)
ndocs = len(ds) - 1
d_ds = torch.from_numpy(ds.astype(np.int64)).to(dev)
d_ids = torch.arange(ndocs, dtype=torch.int64, device=dev)
torch.cuda.synchronize(); torch.cuda.empty_cache()
g = capi.GpuStringIndex(device=0)
g.set_option("profile", 1)
for kv in os.environ.get("CDB_OPTS", "").split(","):
    if kv: g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for i in range(builds):
    g.profile_reset()
    t = time.perf_counter()
    g.build_resident(text.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), ndocs)
    wall = (time.perf_counter() - t) * 1e3
    prof = g.profile()
    free, total = torch.cuda.mem_get_info()
    print(json.dumps({"build": i, "wall_ms": round(wall, 1), "build_ms": round(g.stat("build_ms"), 1), "alloc_ms": round(g.stat("alloc_ms"), 1),
                      "free_ms": round(g.stat("free_ms"), 1), "kernels_ms": round(sum(v["ms"] for v in prof.values()), 1),
                      "self_check_ms": round(g.stat("self_check_ms"), 2), "self_check_pairs": g.stat("self_check_pairs"), "groups": g.stat("bucket_groups"), "fused": g.stat("fused_records"), "packed": g.stat("sa_packed"),
                      "mem": dict(zip(("in_use", "peak", "cached"), [round(x / 2**30, 1) for x in capi.memory_stats()])),
                      "device_free_GiB": round(free / 2**30, 1)}), flush=True)
g.close()
