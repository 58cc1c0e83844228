import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd, dl = 1 << 20, 1024
text = W.random_bytes_torch(nd * dl, 12345, device="cuda")
ds = W.uniform_docs(nd, dl); ids = np.arange(nd, dtype=np.int64)
host = text[: 1 << 28].cpu().numpy()
torch.cuda.synchronize()
g = capi.GpuStringIndex(); g.set_option("profile", 1)
g.build_device(text.data_ptr(), ds, ids)
for kv in os.environ.get("CDB_OPTS", "").split(","):   # e.g. CDB_OPTS=interp_search=0
    if kv: g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for npat, mmin, mmax in ((1000, 4, 16), (100_000, 4, 16), (1_000_000, 4, 16), (100_000, 2, 3), (100, 1, 1)):
    pb, po = W.sample_patterns(host, W.uniform_docs(1 << 18, dl), npat, mmin, mmax, seed=99)
    d_blob = torch.from_numpy(pb).cuda(); d_offs = torch.from_numpy(po.astype(np.int64)).cuda(); torch.cuda.synchronize()
    for rep in range(3):
        g.profile_reset()
        t = time.time(); r = g.query_batch_device(d_blob.data_ptr(), d_offs.data_ptr(), npat, len(pb)); td = time.time() - t
    t = time.time(); rp, ri, rc, hits = g.query_batch(pb, po); th = time.time() - t
    print(f"npat={npat} len {mmin}-{mmax}: device-resident {td*1e3:.3f} ms ({npat/td/1e6:.1f} M/s), host round trip {th*1e3:.3f} ms ({npat/th/1e6:.1f} M/s), hits={hits} rows={len(ri)}")
    print("    ", {k: round(v["ms"], 3) for k, v in sorted(g.profile().items(), key=lambda kv: -kv[1]["ms"])[:6]})
# single-keyword latency through cdb_query
kw = bytes(pb[:1]) + b"abc"
t = time.time()
for _ in range(200): g.query(b"hello")
print(f"single cdb_query latency: {(time.time()-t)/200*1e6:.1f} us")
