"""Several shards behind the C ABI (cdb_shards_* / cdb_comm_*), per-shard index = the HIP path.  A one-GPU box holds all
shards on device 0, where they exchange through device-to-device copies (RCCL refuses two ranks on one GPU); the RCCL
transport itself is exercised with a one-rank communicator (all-gather and grouped broadcasts to itself)."""
import ctypes as C

import numpy as np
import pytest

from coffeedb_amd import workloads as W

pytestmark = pytest.mark.gpu


def _oracle(blob, ds, ids):
    from oracle import OracleIndex
    o = OracleIndex()
    o.add_bulk(ids, blob, ds)
    o.build()
    o.canonicalize()
    return o


@pytest.mark.parametrize("nshards", [2, 3, 5])
def test_sharded_index_matches_single_index(nshards):
    from coffeedb_amd import capi
    blob, ds = W.ragged_corpus(4000, 90, seed=21, lo=0x61, hi=0x64, empty_every=11)
    ids = np.arange(4000, dtype=np.int64) * 5 + 2
    full = _oracle(blob, ds, ids)
    sh = capi.GpuShards([0] * nshards)
    sh.set_option("use_all_devices", 1)
    sh.add_bulk(ids[:1000], blob, ds[:1001])
    for d in range(1000, 1010):                                   # (both add forms)
        sh.add(int(ids[d]), bytes(blob[int(ds[d]):int(ds[d + 1])]))
    sh.add_bulk(ids[1010:], blob, ds[1010:])
    sh.build()
    assert sh.count == nshards and sh.transport == "device copies"
    # per-shard parity: the suffix array of shard i is the one of its documents alone
    bounds = [sh.first_doc(i) for i in range(nshards + 1)]
    assert bounds[0] == 0 and bounds[-1] == 4000 and bounds == sorted(bounds)
    sizes = [int(ds[bounds[i + 1]] - ds[bounds[i]]) for i in range(nshards)]
    assert max(sizes) - min(sizes) <= 2 * 90                       # balanced by bytes, cut at document boundaries
    for i in range(nshards):
        lo, hi = bounds[i], bounds[i + 1]
        o = _oracle(blob, ds[lo:hi + 1], ids[lo:hi])
        g = sh.shard(i)
        assert (g.size, g.bits, g.mask, g.sa_width) == (o.size, o.bits, o.mask, o.sa_width)
        assert np.array_equal(g.sa(), o.sa())
    # global results: identical rows to ONE index over the whole column
    pb, po = W.sample_patterns(blob, ds, 400, 1, 5, seed=3, miss_byte=0x7A)
    got, want = sh.query_batch(pb, po), full.query_batch(pb, po)
    assert got[3] == want[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3]))
    for j in range(0, 400, 17):
        kw = bytes(pb[int(po[j]):int(po[j + 1])])
        assert sh.query(kw) == full.query(kw)
    with pytest.raises(RuntimeError, match="Empty keywords"):
        sh.query(b"")
    # OR over a key's keywords, ranking and highlight spans across the shards = the single-index answers
    kws = [bytes(pb[int(po[j]):int(po[j + 1])]) for j in range(12)]
    one = capi.GpuStringIndex()
    one.add_bulk(ids, blob, ds)
    one.build()
    assert sh.query_or(kws) == one.query_or(kws) == full.filter_or(kws)
    assert sh.query_or(kws, ranked=True, lo=2, limit=25) == one.query_ranked(kws, lo=2, limit=25)
    assert sh.query_spans(kws) == one.query_spans(kws) == full.highlight_spans(kws, ids)
    one.close()
    # a rebuild after more documents replaces every shard
    sh.add(99999, b"abcabcabc")
    sh.build()
    assert sh.query(b"abcabc")[-1] == (99999, 2)
    sh.close()


def test_small_column_stays_on_one_gpu():
    from coffeedb_amd import capi
    blob, ds = W.ascii_corpus(500, 64, seed=4)
    ids = np.arange(500, dtype=np.int64)
    sh = capi.GpuShards([0, 0, 0])
    sh.add_bulk(ids, blob, ds)
    sh.build()                                                    # default policy: shard only beyond max_shard_bytes
    assert sh.count == 1 and sh.transport == "none"
    o = _oracle(blob, ds, ids)
    pats = W.sample_patterns(blob, ds, 100, 2, 6, seed=1)
    got, want = sh.query_batch(*pats), o.query_batch(*pats)
    assert got[3] == want[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3]))
    sh.set_option("max_shard_bytes", 8000)                        # 32000 bytes -> 4 shards wanted, 3 devices given
    sh.build()
    assert sh.count == 3
    got = sh.query_batch(*pats)
    assert got[3] == want[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3]))
    sh.close()


def test_rccl_communicator_of_one_rank_merges_its_own_shard():
    # the RCCL transport of cdb_comm_* (what bench.py --gpus N uses, one process per GPU): with a single rank the
    # all-gather and the grouped broadcasts talk to themselves, and the placement must reproduce the input
    import torch
    from coffeedb_amd import capi
    blob, ds = W.ascii_corpus(3000, 100, seed=9, lo=0x61, hi=0x66)
    ids = np.arange(3000, dtype=np.int64) + 7
    g = capi.GpuStringIndex()
    g.add_bulk(ids, blob, ds)
    g.build()
    pb, po = W.sample_patterns(blob, ds, 500, 1, 6, seed=2)
    d_blob = torch.from_numpy(np.concatenate([pb, np.zeros(16, dtype=np.uint8)])).cuda()
    d_offs = torch.from_numpy(po.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    r = g.query_batch_device(d_blob.data_ptr(), d_offs.data_ptr(), 500, len(pb))
    comm = capi.ShardComm(capi.ShardComm.unique_id(), 0, 1, 0)
    for _ in range(2):
        m = comm.merge(r)
        assert (int(m.npat), int(m.nrows)) == (500, int(r.nrows))

        def dev(ptr, n):
            class A:
                __cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(A(), device="cuda").cpu().numpy()
        assert np.array_equal(dev(m.d_row_ptr, 501), dev(r.d_row_ptr, 501))
        assert np.array_equal(dev(m.d_ids, int(r.nrows)), dev(r.d_ids, int(r.nrows)))
        assert np.array_equal(dev(m.d_counts, int(r.nrows)), dev(r.d_counts, int(r.nrows)))
    comm.close()
    g.close()
