"""Practical HBM ceiling on this box: device-to-device copies (read + write) of the size one radix pass moves."""
import torch, time
for gib in (1, 4, 12):
    n = gib << 30
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    a.zero_(); b.zero_()
    for _ in range(2): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"copy {gib} GiB: {ms:.3f} ms -> {2 * n / ms / 1e9:.2f} TB/s (read + write)")
    # write-only
    e0.record()
    for _ in range(5): b.fill_(1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"fill {gib} GiB: {ms:.3f} ms -> {n / ms / 1e9:.2f} TB/s (write)")
    del a, b
