"""A/B of onesweep kernel configurations inside the real C1 build (split records: u32 key, u32 entry, u8 digit).
usage: pass_ab.py <variants comma-separated> [docs]   — prints build ms and per-kernel average ms per variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W

variants = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [31]
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dl = 1024
text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
ds = W.uniform_docs(nd, dl); ids = np.arange(nd, dtype=np.int64)
torch.cuda.synchronize()
g = capi.GpuStringIndex(); g.set_option("profile", 1)
for kv in os.environ.get("CDB_OPTS", "").split(","):
    if kv: g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
res = {}
ok = {}
for rnd in range(4):
    for v in variants:
        g.set_option("sort_variant", v)
        g.profile_reset()
        try:
            g.build_device(text.data_ptr(), ds, ids)
        except Exception as e:
            ok[v] = f"FAILED: {e}"
            continue
        p = g.profile()
        if rnd == 0:
            r = g.verify()
            good = r["inversions"] == 0 and r["tie_violations"] == 0 and r["invalid_entries"] == 0 and r["entry_sum"] == r["expected_entry_sum"]
            ok[v] = "verified" if good else f"WRONG {r}"
            continue
        res.setdefault(v, []).append((g.stat("build_ms"), {k: (x["ms"] / max(x["launches"], 1), x["launches"]) for k, x in p.items() if k.startswith("rs_onesweep") or k in ("sa_keyhist", "sa_initflags")}))
for v in variants:
    if v not in res:
        print(f"variant {v}: {ok.get(v)}"); continue
    r = res[v]
    b = min(x[0] for x in r)
    names = sorted(r[0][1])
    parts = "  ".join(f"{k.replace('rs_onesweep_', '')} {min(x[1][k][0] for x in r if k in x[1]):.3f}x{r[0][1][k][1]}" for k in names)
    print(f"variant {v:2d} [{ok.get(v)}]: build {b:6.2f} ms | {parts}")
