"""Hybrid initial sort (option hybrid = 1) against the plain LSD sort on several 1 GiB-class corpora: build time, which
plan ran, GPU verification of the result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W

def corpora():
    nd, dl = 1 << 20, 1024
    yield "ascii95 1GiB (C1)", W.random_bytes_torch(nd * dl, 12345, device="cuda"), nd, dl
    yield "ascii95 256MiB", W.random_bytes_torch((nd // 4) * dl, 7, device="cuda"), nd // 4, dl
    yield "a-z 1GiB", W.random_bytes_torch(nd * dl, 5, lo=0x61, hi=0x7A, device="cuda"), nd, dl
    yield "acgt 1GiB", W.random_bytes_torch(nd * dl, 6, lo=0x61, hi=0x64, device="cuda"), nd, dl
    yield "zipf64 1GiB", W.zipf_bytes_torch(nd * dl, seed=2, device="cuda"), nd, dl
    yield "utf8 1GiB", None, None, None

for name, text, nd, dl in corpora():
    if nd is None:
        text, ds = W.utf8_bytes_torch(1 << 30, seed=4, device="cuda")
        ds = np.asarray(ds, dtype=np.uint64)
    else:
        ds = W.uniform_docs(nd, dl)
    ids = np.arange(len(ds) - 1, dtype=np.int64)
    torch.cuda.synchronize()
    line = f"{name:20s}"
    for hyb in (0, 1):
        g = capi.GpuStringIndex(); g.set_option("hybrid", hyb)
        best = 1e9
        for _ in range(3):
            g.build_device(text.data_ptr(), ds, ids)
            best = min(best, g.stat("build_ms"))
        if g.stat("alphabet") > 127 and name.startswith("utf8"):
            g.set_option("reference_compat", 0); g.build_device(text.data_ptr(), ds, ids)   # (plain order for the verifier)
        v = g.verify()
        ok = v["inversions"] == 0 and v["tie_violations"] == 0 and v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"]
        line += f" | hybrid={hyb}: {best:7.2f} ms plan={g.stat('hybrid'):.0f} retries={g.stat('hybrid_retries'):.0f} nsym={g.stat('key_symbols'):.0f} {'ok' if ok else 'WRONG ' + str(v)}"
        g.close()
    print(line, flush=True)
    del text
    torch.cuda.empty_cache()
