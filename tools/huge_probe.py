"""README benchmark shape at full size: 65536 docs x 131072 B = 8 GiB of a-z text (reference README.md:226-232,
test/benchmark.py:18-47) on ONE MI355X: build (bucket-wise >= 2^32 path), GPU-side verification, queries."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dl = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
n = nd * dl
t = time.time(); text = W.random_bytes_torch(n, 777, 0x61, 0x7A, device="cuda"); torch.cuda.synchronize()
print(f"text {n/2**30:.1f} GiB generated in {time.time()-t:.1f}s", flush=True)
ds = W.uniform_docs(nd, dl); ids = np.arange(nd, dtype=np.int64) + 1_000_000
g = capi.GpuStringIndex(); g.set_option("profile", 1)
for rep in range(2):
    g.profile_reset(); t = time.time(); g.build_device(text.data_ptr(), ds, ids); w = time.time() - t
    print(f"build {rep}: {w:.3f} s = {n/2**30/w:.2f} GiB/s width={g.sa_width} bits={g.bits} bucketed={g.stat('bucketed'):.0f} nsym={g.stat('key_symbols'):.0f} symbits={g.stat('symbol_bits'):.0f} "
          f"rounds={g.stat('rounds'):.0f} ext={g.stat('ext_rounds'):.0f} dbl={g.stat('dbl_rounds'):.0f} unres0={g.stat('unresolved_after_initial'):.0f} passes={g.stat('sort_passes'):.0f}", flush=True)
for k, v in sorted(g.profile().items(), key=lambda kv: -kv[1]["ms"])[:8]:
    print(f"   {k:34s} {v['ms']:10.2f} ms x{v['launches']}  {v['bytes']/(v['ms']*1e-3)/1e9 if v['ms'] else 0:7.0f} GB/s")
print("free/total GiB:", [round(x / 2**30, 1) for x in torch.cuda.mem_get_info()], flush=True)
t = time.time(); v = g.verify(); print(v, f"verify {time.time()-t:.2f}s", flush=True)
assert v["inversions"] == 0 and v["tie_violations"] == 0 and v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"]
# benchmark.py shape: 10000 random 5-char keywords
pb = W.random_bytes(50000, 4242, 0x61, 0x7A); po = (np.arange(10001) * 5).astype(np.uint64)
for rep in range(2):
    t = time.time(); rp, ri, rc, hits = g.query_batch(pb, po); w = time.time() - t
print(f"10000 5-char keywords: {w*1e3:.1f} ms ({w/10000*1e6:.1f} us per keyword), hits={hits} rows={len(ri)}")
assert int(rc.sum()) == hits
# brute-force check of three keywords over the whole 8 GiB
for j in (0, 17, 9999):
    kw = torch.from_numpy(pb[5 * j:5 * j + 5].copy()).cuda()
    cnt = 0
    step = 1 << 30
    per_doc = {}
    for s0 in range(0, n, step):
        e0 = min(n, s0 + step + 4)
        seg = text[s0:e0]
        ok = seg[: len(seg) - 4] == kw[0]
        for k in range(1, 5):
            ok &= seg[k: len(seg) - 4 + k] == kw[k]
        pos = torch.nonzero(ok).flatten() + s0
        pos = pos[(pos % dl) + 5 <= dl]
        if s0 + step < n:
            pos = pos[pos < s0 + step]
        d, c = torch.unique(pos // dl, return_counts=True)
        for a, b in zip(d.tolist(), c.tolist()): per_doc[a] = per_doc.get(a, 0) + b
    a, b = int(rp[j]), int(rp[j + 1])
    got = {int(i) - 1_000_000: int(c) for i, c in zip(ri[a:b], rc[a:b])}
    assert got == per_doc, (j, len(got), len(per_doc))
print("brute-force spot checks ok")
