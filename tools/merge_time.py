import sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from coffeedb_amd import shard
class FakeDist:
    @staticmethod
    def all_gather(lst, t):
        for x in lst: x.copy_(t)
world, npat = 8, 100000
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
cnt = (torch.rand(npat, device=dev, generator=g) < 0.9).long() * torch.randint(1, 4, (npat,), device=dev, generator=g)
row_ptr = torch.zeros(npat + 1, dtype=torch.int64, device=dev); row_ptr[1:] = torch.cumsum(cnt, 0)
nrows = int(row_ptr[-1])
ids = torch.arange(nrows, device=dev); counts = torch.ones(nrows, dtype=torch.int64, device=dev)
for _ in range(3): shard.merge_shard_results(torch, FakeDist, row_ptr, ids, counts, world)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): shard.merge_shard_results(torch, FakeDist, row_ptr, ids, counts, world)
torch.cuda.synchronize(); print(f"merge world={world} npat={npat} rows/rank={nrows}: {(time.perf_counter()-t)/10*1e3:.2f} ms")
