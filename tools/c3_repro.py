"""repro of a wrong third shard in the C3 test: the 8 GiB ASCII shard (stream 22) built with various bucket-group limits"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd, dl = 1 << 23, 1024
n = nd * dl
ds = W.uniform_docs(nd, dl)
d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
stream = int(sys.argv[1]) if len(sys.argv) > 1 else 22
text = W.random_bytes_torch(n, 12345, 0x20, 0x7E, stream=stream, device="cuda")
d_ids = torch.arange(nd, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for extra in sys.argv[2:] or ["bucket_group_limit=0"]:
    g = capi.GpuStringIndex()
    for kv in extra.split(","):
        k, v = kv.split("=")
        g.set_option(k, int(v))
    g.set_option("self_check", 0)
    g.build_resident(text.data_ptr(), d_ds.data_ptr(), d_ids.data_ptr(), nd)
    v = g.verify()
    print(extra, "groups", g.stat("bucket_groups"), "fused", g.stat("fused_records"), "sweep", g.stat("sweep_records"), "inv", v["inversions"], "ties", v["tie_violations"],
          "scfb", g.stat("self_check_fallbacks"), "unres", g.stat("unresolved_after_initial"), "ms", round(g.stat("build_ms"), 1), flush=True)
    g.close()
