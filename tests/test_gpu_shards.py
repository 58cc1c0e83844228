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
    # the same batch merged on the devices (all-gatherv + placement kernel) instead of on the host
    sh.set_option("device_merge", 1)
    got = sh.query_batch(pb, po)
    assert got[3] == want[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3]))
    sh.set_option("device_merge", 0)
    # occurrence offsets for the whole batch (BASELINE config 2) across shards = the single index's
    go, wo = sh.query_batch_offsets(pb, po), one.query_batch_offsets(pb, po)
    assert all(np.array_equal(a, b) for a, b in zip(go, wo))
    one.close()
    # a rebuild after more documents replaces every shard
    sh.add(99999, b"abcabcabc")
    sh.build()
    assert sh.query(b"abcabc")[-1] == (99999, 2)
    # a failing rebuild leaves the serving generation untouched (transactional build)
    sh.set_option("debug_fail_build", 1)
    sh.add(100000, b"zzzz")
    with pytest.raises(RuntimeError, match="build failure requested"):
        sh.build()
    assert sh.count == nshards and sh.query(b"abcabc")[-1] == (99999, 2) and sh.query(b"zzzz") == []
    sh.set_option("debug_fail_build", 0)
    sh.build()
    assert sh.query(b"zzzz") == [(100000, 1)]
    sh.close()


def test_sharded_and_persistence_raw_ingest(tmp_path):
    # the rest of the string_index surface over shards: AND across keys (interface.cpp:114-134) with two sharded columns
    # that are cut at different documents, save / load, bulk raw-directory ingest (database.cpp:170-275)
    import struct
    from coffeedb_amd import capi
    nd = 3000
    ids = np.arange(nd, dtype=np.int64) * 3 - 100
    blob_a, ds_a = W.ragged_corpus(nd, 120, seed=5, lo=0x61, hi=0x64, empty_every=0)
    blob_b, ds_b = W.ragged_corpus(nd, 40, seed=6, lo=0x61, hi=0x63, empty_every=7)
    sa, sb = capi.GpuShards([0, 0, 0]), capi.GpuShards([0, 0])
    one_a, one_b = capi.GpuStringIndex(), capi.GpuStringIndex()
    for sh, one, blob, ds in ((sa, one_a, blob_a, ds_a), (sb, one_b, blob_b, ds_b)):
        sh.set_option("use_all_devices", 1)
        sh.add_bulk(ids, blob, ds)
        sh.build()
        one.add_bulk(ids, blob, ds)
        one.build()
    assert [sa.first_doc(i) for i in range(4)] != [sb.first_doc(i) for i in range(3)] + [nd]
    kws_a, kws_b = [b"ab", b"cd", b"dda"], [b"a", b"bc"]
    rows_c = [(int(i), 0) for i in ids[::2]]                      # a numeric key's rows: (id, 0), ascending id
    for ranked, kw in ((False, {}), (True, dict(lo=2, hi=40, limit=50))):
        got = capi.query_and([(sa, kws_a), (sb, kws_b), (None, rows_c)], ranked=ranked, **kw)
        want = capi.query_and([(one_a, kws_a), (one_b, kws_b), (None, rows_c)], ranked=ranked, **kw)
        assert got == want and len(want) > 0
        # a plain index beside a sharded key (the binding used to hand the cdb_index* over as a cdb_shards*)
        mixed = capi.query_and([(one_a, kws_a), (sb, kws_b), (None, rows_c)], ranked=ranked, **kw)
        assert mixed == want
        assert capi.query_and([(sa, kws_a), (one_b, kws_b), (None, rows_c)], ranked=ranked, **kw) == want
    with pytest.raises(RuntimeError, match="Empty keywords"):
        capi.query_and([(sa, kws_a), (one_b, [b""])])              # the error comes from the plain index, reported by type
    with pytest.raises(TypeError):
        capi.query_and([(sa, kws_a), (object(), kws_b)])
    # save / load: same shards, same answers; the column comes back for a later add + rebuild
    path = str(tmp_path / "col.cdbs")
    sa.save(path)
    sl = capi.GpuShards([0, 0, 0])
    sl.load(path)
    assert sl.count == 3 and [sl.first_doc(i) for i in range(4)] == [sa.first_doc(i) for i in range(4)]
    pb, po = W.sample_patterns(blob_a, ds_a, 200, 1, 6, seed=2)
    g1, g2 = sa.query_batch(pb, po), sl.query_batch(pb, po)
    assert g1[3] == g2[3] and all(np.array_equal(a, b) for a, b in zip(g1[:3], g2[:3]))
    sl.add(777777, b"abcdabcd")
    sl.set_option("use_all_devices", 1)
    sl.build()
    assert sl.query(b"abcdabcd")[-1] == (777777, 1) and sl.query(b"ab")[:-1] == sa.query(b"ab")[:len(sl.query(b"ab")) - 1]
    with pytest.raises(RuntimeError, match="Cannot open file"):
        sl.load(str(tmp_path / "missing"))
    assert sl.query(b"abcdabcd")[-1] == (777777, 1)              # a failed load leaves the serving shards alone
    sl.close()
    # raw directory in the reference's record layout (database.cpp:334-378): int64 id, int32 nfields, per field
    # int32 keylen, key, int8 type, value (string: int32 len + bytes)
    raw = tmp_path / "raw"
    raw.mkdir()
    docs = {}
    for k in range(200):
        val = bytes(blob_a[int(ds_a[k]):int(ds_a[k + 1])])
        rec = struct.pack("<qi", 5000 + k, 1) + struct.pack("<i", 3) + b"val" + struct.pack("<b", 3) + struct.pack("<i", len(val)) + val
        (raw / f"{k:06d}").write_bytes(rec)
        docs[5000 + k] = val
    sr, one = capi.GpuShards([0, 0]), capi.GpuStringIndex()
    sr.set_option("use_all_devices", 1)
    assert sr.add_raw_dir(str(raw), b"val") == (200, 200) and one.add_raw_dir(str(raw), b"val") == (200, 200)
    sr.build()
    one.build()
    for kw in (b"ab", b"dcb", b"a"):
        assert sr.query(kw) == one.query(kw)
    # build straight from the caller's separate strings (the shim's string_index::build over several GPUs)
    sv = capi.GpuShards([0, 0, 0])
    sv.set_option("use_all_devices", 1)
    sv.build_views(ids, [bytes(blob_a[int(ds_a[d]):int(ds_a[d + 1])]) for d in range(nd)])
    assert sv.count == 3 and [sv.first_doc(i) for i in range(4)] == [sa.first_doc(i) for i in range(4)]
    g3 = sv.query_batch(pb, po)
    assert g1[3] == g3[3] and all(np.array_equal(a, b) for a, b in zip(g1[:3], g3[:3]))
    sv.add(31337, b"abcdabcdabcd")                               # (the column comes back from the shards)
    sv.build()
    assert sv.query(b"abcdabcdabcd") == [(31337, 1)]
    for x in (sr, one, sa, sb, one_a, one_b, sv):
        x.close()


def test_sharded_queries_run_while_a_rebuild_takes_over():
    # database.cpp:276-280: the old index keeps answering (shared lock) while build() prepares the new one, which then
    # takes over under the exclusive lock.  Four query threads hammer the handle while it is rebuilt three times.
    import threading
    from coffeedb_amd import capi
    blob, ds = W.ascii_corpus(3000, 80, seed=13, lo=0x61, hi=0x65)
    ids = np.arange(3000, dtype=np.int64)
    sh = capi.GpuShards([0, 0, 0])
    sh.set_option("use_all_devices", 1)
    sh.add_bulk(ids, blob, ds)
    sh.build()
    pb, po = W.sample_patterns(blob, ds, 64, 2, 5, seed=3)
    kws = [bytes(pb[int(po[j]):int(po[j + 1])]) for j in range(16)]
    want = {kw: sh.query(kw) for kw in kws}
    want_batch = sh.query_batch(pb, po)
    stop, errors = threading.Event(), []

    def worker(t):
        try:
            n = 0
            while not stop.is_set():
                kw = kws[(n + t) % len(kws)]
                got = sh.query(kw)
                assert got[:len(want[kw])] == want[kw]            # (a rebuilt column only ever adds documents behind)
                if n % 5 == t % 5:
                    gb = sh.query_batch(pb, po)
                    assert gb[3] >= want_batch[3]
                    assert sh.query_or(kws[:4])[:1] == sh.query_or(kws[:4])[:1]
                n += 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    import time
    took = []
    for r in range(3):
        sh.add(10_000 + r, b"zzzz" + kws[0])
        sh.set_option("fast_search", r % 2)                       # options set around / during rebuilds reach the NEW generation
        t0 = time.perf_counter()
        sh.build()
        took.append(time.perf_counter() - t0)
    stop.set()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    assert max(took) < 30, took                                   # the writer is not starved by four overlapping readers
    assert sh.query(b"zzzz") == [(10_000 + r, 1) for r in range(3)]
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
    # counts-only merge (host / rank-local consumers): merged row_ptr + this rank's first merged row of every pattern
    sl = comm.merge_counts(r)
    assert (int(sl.npat), int(sl.nrows_total), int(sl.nrows_local)) == (500, int(r.nrows), int(r.nrows))
    rp = dev(sl.d_row_ptr, 501)
    assert np.array_equal(rp, dev(r.d_row_ptr, 501)) and np.array_equal(dev(sl.d_row_base, 500), rp[:-1])
    assert comm.world == 1 and comm.transport == "rccl"
    comm.close()
    g.close()


def _visible_gpus():
    import torch
    return torch.cuda.device_count()


def test_shards_on_distinct_devices_over_rccl():
    # cdb_shards over DISTINCT GPUs: ncclCommInitAll, one communicator per shard; needs >= 2 visible devices
    from coffeedb_amd import capi
    ng = _visible_gpus()
    if ng < 2:
        pytest.skip("needs at least 2 visible GPUs")
    devs = list(range(min(ng, 4)))
    blob, ds = W.ragged_corpus(6000, 120, seed=8, lo=0x61, hi=0x66, empty_every=9)
    ids = np.arange(6000, dtype=np.int64) * 2 + 1
    full = _oracle(blob, ds, ids)
    sh = capi.GpuShards(devs)
    sh.set_option("use_all_devices", 1)
    sh.add_bulk(ids, blob, ds)
    sh.build()
    assert sh.count == len(devs) and sh.transport == "rccl"
    pb, po = W.sample_patterns(blob, ds, 1000, 1, 6, seed=5, miss_byte=0x7A)
    want = full.query_batch(pb, po)
    for dm in (0, 1):                                            # host merge, then RCCL all-gatherv + placement on the devices
        sh.set_option("device_merge", dm)
        got = sh.query_batch(pb, po)
        assert got[3] == want[3] and all(np.array_equal(a, b) for a, b in zip(got[:3], want[:3]))
    for j in range(0, 1000, 41):
        kw = bytes(pb[int(po[j]):int(po[j + 1])])
        assert sh.query(kw) == full.query(kw)
    kws = [bytes(pb[int(po[j]):int(po[j + 1])]) for j in range(10)]
    assert sh.query_or(kws) == full.filter_or(kws)
    sh.close()


def test_two_process_rccl_merge():
    # cdb_comm_* as bench.py uses it under torchrun: one process per GPU, ncclCommInitRank, merge + merge_counts
    import os
    import subprocess
    import sys
    if _visible_gpus() < 2:
        pytest.skip("needs at least 2 visible GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", os.path.join(root, "tests", "mp_comm_worker.py")], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    assert "MP_COMM_OK" in p.stdout
