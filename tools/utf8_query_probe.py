"""Query throughput on a UTF-8 corpus with and without reference_compat (kept keys / sorted search vs the
reference probe sequence on the rotated array)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi, workloads as W
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
n = int(gib * 2**30) // 16 * 16
g_ = torch.Generator(device="cuda").manual_seed(4)
# cheap UTF-8-like text: 2-byte sequences (lead 0xC2..0xDF, continuation 0x80..0xBF) mixed with ASCII
cls = torch.randint(0, 10, (n // 2,), device="cuda", generator=g_)
a = torch.randint(0x20, 0x7F, (n // 2,), device="cuda", generator=g_).to(torch.uint8)
b = torch.randint(0x20, 0x7F, (n // 2,), device="cuda", generator=g_).to(torch.uint8)
lead = torch.randint(0xC2, 0xE0, (n // 2,), device="cuda", generator=g_).to(torch.uint8)
cont = torch.randint(0x80, 0xC0, (n // 2,), device="cuda", generator=g_).to(torch.uint8)
two = cls >= 6
text = torch.stack([torch.where(two, lead, a), torch.where(two, cont, b)], 1).reshape(-1).contiguous()
nd = n // 1024
ds = W.uniform_docs(nd, 1024); ids = np.arange(nd, dtype=np.int64)
host = text[: 1 << 26].cpu().numpy()
pb, po = W.sample_patterns(host, W.uniform_docs(1 << 16, 1024), npat, 4, 16, seed=9)
d_blob = torch.from_numpy(pb).cuda(); d_offs = torch.from_numpy(po.astype(np.int64)).cuda()
torch.cuda.synchronize()
for compat in (0, 1):
    g = capi.GpuStringIndex(); g.set_option("reference_compat", compat); g.set_option("profile", 1)
    g.build_device(text.data_ptr(), ds, ids); g.build_device(text.data_ptr(), ds, ids)
    for _ in range(2):
        g.profile_reset(); t = time.time()
        r = g.query_batch_device(d_blob.data_ptr(), d_offs.data_ptr(), npat, len(pb)); tq = time.time() - t
    p = g.profile()
    print(f"compat={compat}: build {g.stat('build_ms'):.1f} ms, {npat} patterns in {tq*1e3:.2f} ms ({npat/tq/1e6:.0f} M/s), hits={int(r.nhits)} "
          f"q_search {p['q_search']['ms']:.2f} ms")
