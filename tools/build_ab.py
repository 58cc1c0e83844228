"""A/B of SA-build options on the C1 corpus (one process, interleaved): sort kernel variant, digit width."""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W

nd = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 3, 7, 11]
dbits_list = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 8]
codings = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]   # key_coding option
dl = 1024
text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
ds = W.uniform_docs(nd, dl); ids = np.arange(nd, dtype=np.int64)
torch.cuda.synchronize()
g = capi.GpuStringIndex(); g.set_option("profile", 1)
for kv in os.environ.get("CDB_OPTS", "").split(","):   # e.g. CDB_OPTS=fuse_keygen=0
    if kv: g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
res = {}
for rnd in range(3):
    for v, db, kc in itertools.product(variants, dbits_list, codings):
        g.set_option("sort_variant", v); g.set_option("digit_bits", db); g.set_option("key_coding", kc)
        g.profile_reset()
        g.build_device(text.data_ptr(), ds, ids)
        p = g.profile()
        names = [k for k in p if k.startswith("rs_onesweep_k64_v32")]
        os_ = {"ms": sum(p[k]["ms"] for k in names), "bytes": sum(p[k]["bytes"] for k in names)}
        tg = sum(p[k]["ms"] for k in p if k.startswith("rs_onesweep_textgen"))
        kh = sum(p[k]["ms"] for k in p if k.startswith("sa_keyhist") or k.startswith("sa_keygen") or k == "rs_hist")
        res.setdefault((v, db, kc), []).append((g.stat("build_ms"), os_["ms"], os_["bytes"], g.stat("sort_passes"), g.stat("digit_bits"), g.stat("unresolved_after_initial"), tg, kh))
for k, r in res.items():
    b = sorted(x[0] for x in r[1:])[0]
    o = min(r[1:], key=lambda x: x[1])
    print(f"variant {k[0]:2d} digit_bits {k[1]} key_coding {k[2]} (used {o[4]:.0f}): build {b:7.2f} ms = {nd*dl/2**30/(b*1e-3):6.2f} GiB/s | onesweep {o[1]:7.2f} ms {o[2]/(o[1]*1e-3)/1e9:6.0f} GB/s passes {o[3]:.0f} unres0 {o[5]:.0f} | textgen pass {min(x[6] for x in r[1:]):.2f} ms keyhist {min(x[7] for x in r[1:]):.2f} ms")
