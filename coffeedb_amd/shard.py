"""Multi-GPU sharding of the string index.

Suffixes never cross document boundaries (reference src/index.h:61-65) and a result row belongs to
exactly one document (src/index.cpp:317-321), so the corpus is split into doc-aligned byte ranges, one
independent suffix array per GPU.  Every shard answers the whole (broadcast) pattern batch for its
documents; the per-shard CSR match lists are merged with all-gathers over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Concatenating the rows of a
pattern in shard order is already ascending in global document index — no reduction is needed.
"""
import numpy as np


def shard_bounds(doc_start, world):
    """Doc-aligned split into `world` contiguous ranges balanced by bytes.
    Returns doc indices b[0..world] with shard r = docs [b[r], b[r+1])."""
    doc_start = np.asarray(doc_start, dtype=np.uint64)
    nd = len(doc_start) - 1
    total = int(doc_start[-1])
    b = [0]
    for r in range(1, world):
        target = total * r // world
        d = int(np.searchsorted(doc_start, np.uint64(target), side="left"))
        b.append(min(max(d, b[-1]), nd))
    b.append(nd)
    return b


def merge_shard_results(torch, dist, row_ptr, ids, counts, world):
    """All-gather merge of per-shard CSR results.

    row_ptr: int64[npat+1], ids/counts: int64[nrows] of THIS shard (torch tensors on the collective's
    device).  Returns (global_row_ptr int64[npat+1], global_ids, global_counts), identical on every rank.
    """
    device = row_ptr.device
    npat = row_ptr.numel() - 1
    nrows = int(ids.numel())
    cnt = (row_ptr[1:] - row_ptr[:-1]).contiguous()
    cnt_list = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(cnt_list, cnt)
    all_cnt = torch.stack(cnt_list)                      # [world, npat]
    rows_per_rank = all_cnt.sum(1)
    maxrows = max(int(rows_per_rank.max().item()), 1)
    pad = torch.zeros(2, maxrows, dtype=torch.int64, device=device)
    if nrows:
        pad[0, :nrows] = ids
        pad[1, :nrows] = counts
    pad_list = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(pad_list, pad)                       # all-gatherv via padding to the longest list
    # one vectorised placement for all shards: rows are enumerated shard-major (shard q, its rows in
    # order); row i of shard q belongs to pattern pat[i] and lands behind the rows that earlier shards
    # contribute to the same pattern
    total = all_cnt.sum(0)
    g_row_ptr = torch.zeros(npat + 1, dtype=torch.int64, device=device)
    torch.cumsum(total, 0, out=g_row_ptr[1:])
    before = torch.cumsum(all_cnt, 0) - all_cnt            # [world, npat] rows of earlier shards per pattern
    flat_cnt = all_cnt.reshape(-1)                         # (shard, pattern) group sizes, shard-major
    nrows_total = int(g_row_ptr[-1].item())
    out = torch.empty(2, nrows_total, dtype=torch.int64, device=device)
    if nrows_total:
        group = torch.repeat_interleave(torch.arange(world * npat, device=device), flat_cnt)
        group_start = torch.cumsum(flat_cnt, 0) - flat_cnt
        pos = torch.arange(nrows_total, device=device)
        within = pos - group_start[group]
        q = group // npat
        pat = group - q * npat
        dest = g_row_ptr[pat] + before.reshape(-1)[group] + within
        rank_start = torch.cumsum(rows_per_rank, 0) - rows_per_rank
        src = q * maxrows + (pos - rank_start[q])
        allrows = torch.stack(pad_list)                    # [world, 2, maxrows]
        out[0, dest] = allrows[:, 0, :].reshape(-1)[src]
        out[1, dest] = allrows[:, 1, :].reshape(-1)[src]
    return g_row_ptr, out[0], out[1]


def merge_shard_counts(torch, dist, row_ptr, world, rank):
    """Counts-only merge (the consumer is not on the GPU — every shard keeps its own slice): all-gather of
    the per-pattern row counts only.  Returns (global_row_ptr int64[npat+1], row_base int64[npat]): this rank's rows of
    pattern j are the merged rows [row_base[j], row_base[j] + local count of j).  The torch restatement of
    cdb_comm_merge_counts (shards.hip), used by the gloo test mode."""
    cnt = (row_ptr[1:] - row_ptr[:-1]).contiguous()
    cnt_list = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(cnt_list, cnt)
    all_cnt = torch.stack(cnt_list)                      # [world, npat]
    g_row_ptr = torch.zeros(cnt.numel() + 1, dtype=torch.int64, device=row_ptr.device)
    torch.cumsum(all_cnt.sum(0), 0, out=g_row_ptr[1:])
    base = g_row_ptr[:-1] + all_cnt[:rank].sum(0)
    return g_row_ptr, base


class _DevArr:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class ShardMerger:
    """bench.py's N > 1 step: merges this rank's device-resident CSR with the other shards'.  Over RCCL the merge is the
    library's own (cdb_comm_* of the C ABI: all-gather of the row counts, all-gatherv of the rows, HIP placement kernel);
    `--backend gloo --share-gpu` (a one-GPU box, testing only) goes through torch.distributed instead."""

    def __init__(self, capi, index, dist, rank, world, coll_device, device, mode="counts"):
        import torch
        self.torch, self.dist, self.index, self.world, self.device, self.coll_device = torch, dist, index, world, device, coll_device
        self.rank, self.mode = rank, mode   # counts: cdb_comm_merge_counts (every rank keeps its slice); full: all-gatherv
        self.comm = None
        self.note = "torch.distributed all-gathers (gloo test mode)"
        if coll_device == device and hasattr(capi, "ShardComm"):
            uid = torch.zeros(128, dtype=torch.uint8, device=device)
            if rank == 0:
                uid.copy_(torch.from_numpy(capi.ShardComm.unique_id()).to(device))
            if world > 1:
                dist.broadcast(uid, 0)
            try:
                self.comm = capi.ShardComm(uid.cpu().numpy(), rank, world, device.index or 0)
                ok = 1
            except RuntimeError as e:   # reported in the bench line, never silent
                self.note = f"torch.distributed all-gathers (cdb_comm_create failed on rank {rank}: {e})"
                ok = 0
            if world > 1:   # all ranks use the same path
                t = torch.tensor([ok], dtype=torch.int32, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                if int(t.item()) == 0 and self.comm is not None:
                    self.comm.close()
                    self.comm = None
                    self.note = "torch.distributed all-gathers (cdb_comm_create failed on another rank)"
            if self.comm is not None:
                self.note = ("cdb_comm_merge_counts (RCCL all-gather of the row counts; every rank keeps its own rows and learns "
                             "where they sit in the merged CSR)" if mode == "counts" else
                             "cdb_comm_merge (RCCL all-gather + grouped broadcasts + placement kernel)")
                self.note += f"; {self.comm.world} RCCL ranks, transport {self.comm.transport}"

    def merge(self, r, npat):
        if self.comm is not None:
            return self.comm.merge_counts(r) if self.mode == "counts" else self.comm.merge(r)
        torch = self.torch
        nrows = int(r.nrows)
        row_ptr = torch.as_tensor(_DevArr(r.d_row_ptr, npat + 1, "<i8"), device=self.device)
        if self.mode == "counts":
            return merge_shard_counts(torch, self.dist, row_ptr.to(self.coll_device), self.world, self.rank)
        if nrows:
            ids = torch.as_tensor(_DevArr(r.d_ids, nrows, "<i8"), device=self.device)
            cnt = torch.as_tensor(_DevArr(r.d_counts, nrows, "<i8"), device=self.device)
        else:
            ids = cnt = torch.empty(0, dtype=torch.int64, device=self.device)
        return merge_shard_results(torch, self.dist, row_ptr.to(self.coll_device), ids.to(self.coll_device),
                                   cnt.to(self.coll_device), self.world)

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
