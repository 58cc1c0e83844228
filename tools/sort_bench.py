"""A/B of the onesweep kernel configurations on random keys (interleaved rounds, one process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coffeedb_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 48
variants = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 3, 4, 5]
g = torch.Generator(device="cuda").manual_seed(1)
keys0 = torch.randint(0, (1 << bits) - 1, (n,), dtype=torch.int64, device="cuda", generator=g)
vals0 = torch.arange(n, dtype=torch.int32, device="cuda")
res = {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        k, x = keys0.clone(), vals0.clone()
        torch.cuda.synchronize()
        ms, passes = capi.debug_radix_sort(k.data_ptr(), x.data_ptr(), n, 4, bits, v)
        res[v].append(ms / passes)
for v in variants:
    r = sorted(res[v][1:])
    gbs = n * 24 / (r[len(r) // 2] * 1e-3) / 1e9
    print(f"variant {v}: per-pass ms min {r[0]:.3f} med {r[len(r)//2]:.3f} -> {gbs:.0f} GB/s algorithmic (k64,v32)")
