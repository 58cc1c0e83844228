"""Worker of tests/test_gpu_shards.py::test_two_process_rccl_merge (run under torch.distributed.run, one rank per GPU):
every rank builds its document-aligned shard on its own GPU through the C ABI, answers the whole pattern batch, and the
ranks merge over cdb_comm (RCCL): full all-gatherv merge and the counts-only merge, both checked against the CPU oracle
over the whole column."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from coffeedb_amd import capi, shard, workloads as W
    from oracle import OracleIndex
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    blob, ds = W.ragged_corpus(8000, 100, seed=77, lo=0x61, hi=0x66, empty_every=13)
    ids = np.arange(8000, dtype=np.int64) * 3 + 5
    b = shard.shard_bounds(ds, world)
    lo, hi = b[rank], b[rank + 1]
    g = capi.GpuStringIndex(device=local)
    g.build_view(ids[lo:hi], blob, ds[lo:hi + 1])
    pb, po = W.sample_patterns(blob, ds, 2000, 1, 6, seed=9, miss_byte=0x7A)
    npat = len(po) - 1
    d_blob = torch.from_numpy(np.concatenate([pb, np.zeros(16, dtype=np.uint8)])).to(dev)
    d_offs = torch.from_numpy(po.astype(np.int64)).to(dev)
    torch.cuda.synchronize()
    r = g.query_batch_device(d_blob.data_ptr(), d_offs.data_ptr(), npat, len(pb))
    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid.copy_(torch.from_numpy(capi.ShardComm.unique_id()).to(dev))
    dist.broadcast(uid, 0)
    comm = capi.ShardComm(uid.cpu().numpy(), rank, world, local)
    assert comm.world == world and comm.transport == "rccl"

    def arr(ptr, n):
        if n == 0:
            return np.zeros(0, dtype=np.int64)

        class A:
            __cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(A(), device=dev).cpu().numpy()

    o = OracleIndex()
    o.add_bulk(ids, blob, ds)
    o.build()
    wrp, wid, wcn, _ = o.query_batch(pb, po)
    m = comm.merge(r)                                             # every rank ends up with the merged CSR
    assert int(m.nrows) == len(wid)
    assert np.array_equal(arr(m.d_row_ptr, npat + 1), wrp.astype(np.int64))
    assert np.array_equal(arr(m.d_ids, len(wid)), wid) and np.array_equal(arr(m.d_counts, len(wid)), wcn)
    sl = comm.merge_counts(r)                                     # counts only: this rank's slice placed by row_base
    assert int(sl.nrows_total) == len(wid) and int(sl.nrows_local) == int(r.nrows)
    assert np.array_equal(arr(sl.d_row_ptr, npat + 1), wrp.astype(np.int64))
    base, lrp = arr(sl.d_row_base, npat), arr(r.d_row_ptr, npat + 1)
    lid, lcn = arr(r.d_ids, int(r.nrows)), arr(r.d_counts, int(r.nrows))
    for j in range(npat):
        k = int(lrp[j + 1] - lrp[j])
        assert np.array_equal(wid[base[j]:base[j] + k], lid[lrp[j]:lrp[j + 1]])
        assert np.array_equal(wcn[base[j]:base[j] + k], lcn[lrp[j]:lrp[j + 1]])
    comm.close()
    g.close()
    dist.barrier()
    if rank == 0:
        print("MP_COMM_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
