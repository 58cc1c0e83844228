"""N > 1 path on CPU: two processes (gloo), doc-aligned shards, all-gather merge of per-shard match
lists.  The per-shard index here is the CPU oracle (test infrastructure) — what is under test is the
sharding arithmetic and the collective merge that bench.py runs over RCCL on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from coffeedb_amd import shard, workloads as W


def test_shard_bounds_are_doc_aligned_and_balanced():
    _, ds = W.ragged_corpus(1000, 50, seed=3, empty_every=9)
    b = shard.shard_bounds(ds, 4)
    assert b[0] == 0 and b[-1] == 1000 and all(b[i] <= b[i + 1] for i in range(4))
    sizes = [int(ds[b[i + 1]] - ds[b[i]]) for i in range(4)]
    assert max(sizes) - min(sizes) <= 2 * 50
    assert shard.shard_bounds(W.uniform_docs(8, 16), 8) == list(range(9))
    assert shard.shard_bounds(np.array([0, 0, 0], dtype=np.uint64), 2) == [0, 0, 2]


def _worker(rank, world, port, q):
    from oracle import OracleIndex
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blob, ds = W.ragged_corpus(600, 80, seed=21, lo=0x61, hi=0x64, empty_every=11)
        ids = np.arange(600, dtype=np.int64) * 5 + 2
        pb, po = W.sample_patterns(blob, ds, 150, 1, 4, seed=3, miss_byte=0x7A)
        b = shard.shard_bounds(ds, world)
        lo, hi = b[rank], b[rank + 1]
        o = OracleIndex()
        o.add_bulk(ids[lo:hi], blob, ds[lo:hi + 1])   # this rank's doc-aligned byte range
        o.build(1)
        rp, ri, rc, _ = o.query_batch(pb, po)
        g_rp, g_ids, g_cnt = shard.merge_shard_results(
            torch, dist, torch.from_numpy(rp.astype(np.int64)), torch.from_numpy(ri), torch.from_numpy(rc), world)
        full = OracleIndex()
        full.add_bulk(ids, blob, ds)
        full.build(1)
        frp, fi, fc, _ = full.query_batch(pb, po)
        ok = (np.array_equal(g_rp.numpy(), frp.astype(np.int64)) and np.array_equal(g_ids.numpy(), fi)
              and np.array_equal(g_cnt.numpy(), fc))
        # counts-only merge: merged row_ptr + where this rank's own rows sit in the merged row stream
        c_rp, base = shard.merge_shard_counts(torch, dist, torch.from_numpy(rp.astype(np.int64)), world, rank)
        ok = ok and np.array_equal(c_rp.numpy(), frp.astype(np.int64))
        for j in range(len(po) - 1):
            k = int(rp[j + 1] - rp[j])
            ok = ok and np.array_equal(fi[int(base[j]):int(base[j]) + k], ri[int(rp[j]):int(rp[j + 1])])
        q.put((rank, bool(ok), int(g_rp[-1])))
    finally:
        dist.destroy_process_group()


def test_two_rank_merge_matches_single_index():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert [r[1] for r in res] == [True, True], res
    assert res[0][2] == res[1][2] > 0
