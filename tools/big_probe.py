import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd = int(sys.argv[1]); dl = int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "ascii"
n = nd * dl
if kind == "zipf":
    w = 1.0 / torch.arange(1, 65, dtype=torch.float64, device="cuda"); cdf = torch.cumsum(w / w.sum(), 0)
    text = torch.empty(n, dtype=torch.uint8, device="cuda")
    step = 1 << 28
    gen = torch.Generator(device="cuda").manual_seed(2)
    for s in range(0, n, step):
        e = min(n, s + step)
        u = torch.rand(e - s, dtype=torch.float64, device="cuda", generator=gen)
        text[s:e] = (0x30 + torch.searchsorted(cdf, u).clamp_(0, 63)).to(torch.uint8)
else:
    text = W.random_bytes_torch(n, 12345, device="cuda")
ds = W.uniform_docs(nd, dl); ids = np.arange(nd, dtype=np.int64)
torch.cuda.synchronize(); torch.cuda.empty_cache()
g = capi.GpuStringIndex(); g.set_option("profile", 1)
for kv in os.environ.get("CDB_OPTS", "").split(","):   # e.g. CDB_OPTS=streamed_build=1
    if kv: g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for i in range(2):
    g.profile_reset(); t = time.time(); g.build_device(text.data_ptr(), ds, ids); w_ = time.time() - t
    print(f"{kind} n={n/2**30:.2f} GiB width={g.sa_width} build {w_*1e3:.1f} ms ({n/2**30/w_:.2f} GiB/s) rounds={g.stat('rounds'):.0f} ext={g.stat('ext_rounds'):.0f} dbl={g.stat('dbl_rounds'):.0f} "
          f"unres0={g.stat('unresolved_after_initial'):.0f} passes={g.stat('sort_passes'):.0f} nsym={g.stat('key_symbols'):.0f} symbits={g.stat('symbol_bits'):.0f} fused={g.stat('fused_keygen'):.0f} depth={g.stat('final_depth'):.0f}", flush=True)
for k, v in sorted(g.profile().items(), key=lambda kv: -kv[1]["ms"])[:16]:
    print(f"   {k:32s} {v['ms']:9.3f} ms x{v['launches']}")
t = time.time(); print(g.verify(), f"verify {time.time()-t:.2f}s")
print("free/total GiB:", [x / 2**30 for x in torch.cuda.mem_get_info()])
if len(sys.argv) > 4:   # query stage: N patterns of length lo-hi sampled from a 256 MiB host prefix, with offsets
    npat, lo, hi = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    host = text[: 1 << 28].cpu().numpy()
    pb, po = W.sample_patterns(host, W.uniform_docs((1 << 28) // dl, dl), npat, lo, hi, seed=99)
    for rep in range(3):   # results are released before the next batch, whose pinned blocks are then reused
        rp = ri = rc = None
        t = time.time(); rp, ri, rc, hits = g.query_batch(pb, po); tq = time.time() - t
    print(f"query_batch {npat} patterns len {lo}-{hi}: {tq*1e3:.1f} ms ({npat/tq/1e6:.1f} M/s) hits={hits} rows={len(ri)} library {g.stat('query_ms'):.1f} ms = upload {g.stat('query_upload_ms'):.1f} + device {g.stat('query_device_ms'):.1f} + download {g.stat('query_download_ms'):.1f}")
    g.profile_reset(); g.query_batch(pb, po)
    for k, v in sorted(g.profile().items(), key=lambda kv: -kv[1]["ms"])[:10]:
        print(f"   {k:32s} {v['ms']:9.3f} ms x{v['launches']}")
    t = time.time(); rp2, ri2, rc2, hp, off = g.query_batch_offsets(pb, po); tq = time.time() - t
    assert np.array_equal(ri, ri2) and np.array_equal(rc, rc2) and len(off) == hits
    # spot-check a few rows against the text
    for r in np.random.default_rng(1).choice(len(ri2), 20, replace=False):
        j = int(np.searchsorted(rp2, r, side="right") - 1)
        kw = pb[int(po[j]):int(po[j + 1])]
        d = int(ri2[r])
        doc = text[d * dl:(d + 1) * dl].cpu().numpy()
        for o_ in off[int(hp[r]):int(hp[r + 1])]:
            assert np.array_equal(doc[int(o_):int(o_) + len(kw)], kw)
    print(f"query_batch_offsets: {tq*1e3:.1f} ms, {len(off)} occurrence offsets, spot checks ok")
