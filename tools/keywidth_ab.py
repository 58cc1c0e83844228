"""A/B of the initial sort's key width (option key_symbols) on a bench workload: is one radix pass fewer worth the larger
share of suffixes it leaves to the refinement rounds?  usage: keywidth_ab.py <workload> <sym,sym,...> [reps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from coffeedb_amd import capi, workloads as W

name = sys.argv[1] if len(sys.argv) > 1 else "c1"
syms = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = bench.WORKLOADS[name]
dev = torch.device("cuda", 0)
text, ds, n = bench.make_corpus(torch, W, cfg, 0, dev)
ndocs = len(ds) - 1
d_ds = torch.from_numpy(ds.astype(np.int64)).to(dev)
d_ids = torch.arange(ndocs, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
torch.cuda.empty_cache()
for k in syms:
    g = capi.GpuStringIndex(device=0)
    g.set_option("profile", 1)
    g.set_option("key_symbols", k)
    for kv in os.environ.get("CDB_OPTS", "").split(","):   # e.g. CDB_OPTS=sweep_records=0
        if "=" in kv:
            g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ms = []
    for i in range(reps + 1):
        if i == 1:
            g.profile_reset()
        g.build_resident(text.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), ndocs)
        ms.append(g.stat("build_ms"))
    prof = g.profile()
    v = g.verify_reference() if cfg["kind"] == "utf8" else g.verify()
    top = {kk: round(vv["ms"] / reps, 2) for kk, vv in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get('CDB_TOP', '7'))]}
    st = bench.build_stats(g)
    print(json.dumps({"workload": name, "key_symbols_option": k, "key_symbols": st.get("key_symbols"), "build_ms": [round(x, 2) for x in ms[1:]],
                      "unresolved_after_initial": st.get("unresolved_after_initial"), "unresolved_share": round(st.get("unresolved_after_initial", 0) / n, 5),
                      "sort_passes": st.get("sort_passes"), "bucket_groups": st.get("bucket_groups"), "sweep_records": st.get("sweep_records"), "ext_rounds": st.get("ext_rounds"), "dbl_rounds": st.get("dbl_rounds"),
                      "alg_bytes_per_suffix": round(sum(x["bytes"] for x in prof.values()) / reps / n, 1), "kernels_ms": top,
                      "verify": {kk: int(vv) for kk, vv in v.items() if kk in ("inversions", "tie_violations", "invalid_entries", "violations")}}), flush=True)
    g.close()
