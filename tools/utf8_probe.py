"""BASELINE north-star shape: synthetic valid UTF-8 corpus (50 % 1-byte, 30 % 2-byte, 20 % 3-byte code points),
documents of ~1 KiB cut at code-point boundaries, generated on the GPU; SA build with and without
reference_compat, GPU-side verification (compat = 0: globally sorted)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coffeedb_amd import capi

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
ncp = int(gib * 2**30 / 1.7)
g_ = torch.Generator(device="cuda").manual_seed(4)
cuts = []
step = 1 << 27
total = 0
text_buf = torch.empty(int(ncp * 1.72) + (1 << 20), dtype=torch.uint8, device="cuda")
for s in range(0, ncp, step):
    m = min(step, ncp - s)
    cls = torch.randint(0, 10, (m,), device="cuda", generator=g_)
    val = torch.randint(0, 1 << 30, (m,), device="cuda", generator=g_)
    cp = torch.where(cls < 5, 0x20 + val % 0x5F, torch.where(cls < 8, 0x80 + val % 0x780, 0x800 + val % 0xD000))
    ln = torch.where(cp < 0x80, 1, torch.where(cp < 0x800, 2, 3))
    off = torch.cumsum(ln, 0) - ln
    nb = int((off[-1] + ln[-1]).item())
    out = text_buf[total:total + nb]
    out.zero_()
    one, two, three = cp < 0x80, (cp >= 0x80) & (cp < 0x800), cp >= 0x800
    out[off[one]] = cp[one].to(torch.uint8)
    out[off[two]] = (0xC0 | (cp[two] >> 6)).to(torch.uint8); out[off[two] + 1] = (0x80 | (cp[two] & 0x3F)).to(torch.uint8)
    out[off[three]] = (0xE0 | (cp[three] >> 12)).to(torch.uint8); out[off[three] + 1] = (0x80 | ((cp[three] >> 6) & 0x3F)).to(torch.uint8)
    out[off[three] + 2] = (0x80 | (cp[three] & 0x3F)).to(torch.uint8)
    # document cuts at code-point boundaries, one about every 1024 bytes (first code point at or after k * 1024)
    first = (total + 1023) // 1024 * 1024
    tg = torch.arange(first, total + nb, 1024, device="cuda") - total
    idx = torch.searchsorted(off, tg).clamp_(max=m - 1)
    cuts.append((off[idx] + total).cpu().numpy())
    total += nb
    del cls, val, cp, ln, off, one, two, three, tg, idx
n16 = (total // 16) * 16
text = text_buf[:n16]
cuts = np.concatenate(cuts); cuts = cuts[cuts < n16]
ds = np.unique(np.concatenate([[0], cuts, [n16]])).astype(np.uint64)
torch.cuda.empty_cache()
nd = len(ds) - 1
n = int(ds[-1])
print(f"UTF-8 corpus: {n/2**30:.2f} GiB, {nd} docs, max doc {int((ds[1:]-ds[:-1]).max())} B", flush=True)
text[:64].cpu().numpy().tobytes().decode("utf-8", errors="strict") if False else None
ids = np.arange(nd, dtype=np.int64)
torch.cuda.synchronize()
for compat in (0, 1):
    g = capi.GpuStringIndex(); g.set_option("profile", 1); g.set_option("reference_compat", compat)
    for rep in range(4):
        g.profile_reset(); t = time.time(); g.build_device(text.data_ptr(), ds, ids); w = time.time() - t
        print(f"   rep {rep}: {w*1e3:.1f} ms", flush=True)
    print(f"compat={compat}: build {w*1e3:.1f} ms = {n/2**30/w:.2f} GiB/s width={g.sa_width} nsym={g.stat('key_symbols'):.0f} symbits={g.stat('symbol_bits'):.0f} alphabet={g.stat('alphabet'):.0f} "
          f"fused={g.stat('fused_keygen'):.0f} bucketed={g.stat('bucketed'):.0f} rounds={g.stat('rounds'):.0f} ext={g.stat('ext_rounds'):.0f} dbl={g.stat('dbl_rounds'):.0f} unres0={g.stat('unresolved_after_initial'):.0f} "
          f"passes={g.stat('sort_passes'):.0f} rotations={g.stat('compat_rotations'):.0f} depth={g.stat('compat_depth'):.0f}", flush=True)
    for k, v in sorted(g.profile().items(), key=lambda kv: -kv[1]["ms"])[:6]:
        print(f"   {k:34s} {v['ms']:10.2f} ms x{v['launches']}  {v['bytes']/(v['ms']*1e-3)/1e9 if v['ms'] else 0:7.0f} GB/s")
    v = g.verify(); print("  ", v)
    if compat == 0:
        assert v["inversions"] == 0 and v["tie_violations"] == 0
    assert v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"]
    g.close()
