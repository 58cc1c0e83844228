"""Full-size builds under option variants, each verified pair by pair on the device (inversions / ties / invalid entries / entry sum).
Written as the repro of a wrong third shard in the C3 test: the 8 GiB ASCII shard (stream 22) built with various bucket-group
limits.  usage: python tools/c3_repro.py <stream | zipf | utf8> opt=val,... [...]      (test-side tool; needs a GPU)"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd, dl = 1 << 23, 1024
n = nd * dl
ds = W.uniform_docs(nd, dl)
d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
kind = sys.argv[1] if len(sys.argv) > 1 else "22"
if kind == "zipf":
    text = W.zipf_bytes_torch(n, seed=2, nsym=64, base=0x30, device="cuda")
elif kind == "utf8":
    text, ds = W.utf8_bytes_torch(n, seed=4, device="cuda")
    nd = len(ds) - 1
    d_ds = torch.from_numpy(ds.astype(np.int64)).cuda()
else:
    text = W.random_bytes_torch(n, 12345, 0x20, 0x7E, stream=int(kind), device="cuda")
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
    print(extra, "groups", g.stat("bucket_groups"), "fused", g.stat("fused_records"), "sweep", g.stat("sweep_records"), "vl", g.stat("vl_key_bits"), "partial", g.stat("partial_levels"), "inv", v["inversions"], "ties", v["tie_violations"],
          "invalid", v["invalid_entries"], "sum_ok", v["entry_sum"] == v["expected_entry_sum"], "rot", g.stat("compat_rotations"),
          "scfb", g.stat("self_check_fallbacks"), "unres", g.stat("unresolved_after_initial"), "ms", round(g.stat("build_ms"), 1), flush=True)
    g.close()
