"""Verify after publish (option self_check = 3, the default): the order proof behind every build.

The reference's suffix array is sorted by construction (std::sort leaves, index.cpp:86-95).  This library's radix passes rest on
an observed LDS lane order (radix_sort.h:12-15), and the sample behind a build proves nothing about one stray pair — so after
cdb_build* / cdb_load return, a helper thread compares EVERY adjacent pair of the published array against the text, beside the
queries; damage makes the handle rebuild itself under its lock.  Checked here with the oracle as the referee."""
import threading
import time

import numpy as np
import pytest

from coffeedb_amd import workloads as W

pytestmark = pytest.mark.gpu


def _corpus(nd=60000, wide=False, seed=3):
    lens = W.random_bytes(nd, seed, 1, 200).astype(np.uint64)
    if wide:
        lens[nd // 2] = 150_000          # one long document: 8-byte entries (stored packed)
    ds = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blob = W.random_bytes(int(ds[-1]), seed + 1, 0x61, 0x6A)
    return blob, ds


def _oracle(blob, ds, ids):
    from oracle import OracleIndex
    o = OracleIndex()
    o.add_bulk(ids, blob, ds)
    o.build(4)
    o.canonicalize(4)
    return o


@pytest.mark.parametrize("wide", [False, True])
def test_default_build_is_proved_after_it_returns(wide):
    from coffeedb_amd import capi
    blob, ds = _corpus(wide=wide)
    ids = np.arange(len(ds) - 1, dtype=np.int64)
    g = capi.GpuStringIndex()
    g.add_bulk(ids, blob, ds)
    g.build()
    assert g.proof_wait(20_000) == 2                       # proved
    assert g.stat("order_proved") == 1 and g.stat("proof_pairs") == g.size - 1
    assert g.stat("proof_bad_pairs") == 0 and g.stat("proof_invalid_entries") == 0 and g.stat("self_check_fallbacks") == 0
    assert g.stat("proof_skipped_pairs") == 0             # (pure ASCII: every pair is judged)
    assert g.stat("self_check_coverage") < 1               # (the build itself only sampled)
    # the other levels: 1 = sample only (nothing is proved), 2 = every pair before the build returns
    g.set_option("self_check", 1)
    g.build()
    assert g.proof_wait(0) == 0 and g.stat("order_proved") == 0
    g.set_option("self_check", 2)
    g.build()
    assert g.proof_wait(0) == 0 and g.stat("order_proved") == 1 and g.stat("self_check_coverage") == 1
    g.close()


@pytest.mark.parametrize("wide", [False, True])
def test_one_damaged_pair_is_found_and_repaired_while_four_threads_query(wide):
    """ONE swapped adjacent pair behind the build's own check (test hook debug_damage_after_build): the sample cannot see it (it ran
    before), the proof must; the handle rebuilds itself and serves the right array afterwards.  Four threads send lone keywords
    the whole time: none of their calls may fail, and every answer from the moment the proof reports "repaired" is the oracle's."""
    from coffeedb_amd import capi
    blob, ds = _corpus(wide=wide, seed=11)
    nd = len(ds) - 1
    ids = np.arange(nd, dtype=np.int64) * 2 + 1
    o = _oracle(blob, ds, ids)
    pb, po = W.sample_patterns(blob, ds, 64, 3, 7, seed=5)
    kws = [bytes(pb[int(po[j]):int(po[j + 1])]) for j in range(64)]
    want = {kw: o.query(kw) for kw in kws}
    g = capi.GpuStringIndex()
    g.add_bulk(ids, blob, ds)
    g.set_option("debug_damage_after_build", int(ds[-1]) // 2)
    stop = threading.Event()
    repaired_at = [None]
    errors, served, late_wrong = [], [0, 0, 0, 0], []

    def client(t):
        j = t
        while not stop.is_set():
            kw = kws[j % len(kws)]
            try:
                asked = time.monotonic()
                got = g.query(kw)
            except Exception as e:  # noqa: BLE001 - the assertion below reports it
                errors.append(repr(e))
                return
            if repaired_at[0] is not None and asked > repaired_at[0] and got != want[kw]:
                late_wrong.append((kw, got, want[kw]))
            served[t] += 1
            j += 4

    g.build()
    threads = [threading.Thread(target=client, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    state = g.proof_wait(60_000)
    repaired_at[0] = time.monotonic()
    time.sleep(0.3)                                        # (the clients keep asking the repaired index for a while)
    stop.set()
    for th in threads:
        th.join()
    assert state == 3, state                               # damage found and repaired
    assert not errors, errors[:3]
    assert not late_wrong, late_wrong[:3]
    assert min(served) > 0
    assert g.stat("proof_bad_pairs") >= 1 and g.stat("self_check_fallbacks") == 1 and g.stat("order_proved") == 1
    assert g.stat("proof_repair_ms") > 0
    assert np.array_equal(g.sa(), o.sa())                  # the replacement is the oracle's array
    for kw in kws:
        assert g.query(kw) == want[kw]
    # the next build of the same handle is undamaged (the hook fires once) and is proved like any other
    g.build()
    assert g.proof_wait(20_000) == 2 and g.stat("self_check_fallbacks") == 1
    g.close()


def test_the_proof_also_judges_the_pairs_whose_order_depends_on_bucket_sizes():
    """Under reference_compat (default) text with bytes >= 0x80 is laid out in the reference's signed child order inside radix nodes
    (index.h:66-73, buckets of more than chuck_size suffixes) and in unsigned order below them (index.cpp:86-95): for a pair whose
    first differing bytes lie on different sides of 0x80 the right order depends on the size of the bucket the two suffixes share.
    A local comparison cannot judge those (8 % of the pairs of synthetic UTF-8); the lane that meets one finds the bucket size by
    galloping over the array (ref_bucket_is_node, the code of cdb_debug_verify_reference).  Nothing is left unjudged, and damage
    among exactly those pairs is found and repaired."""
    from coffeedb_amd import capi
    blob, ds = W.utf8_corpus(4000, 300, seed=17)
    ids = np.arange(len(ds) - 1, dtype=np.int64)
    o = _oracle(blob, ds, ids)
    g = capi.GpuStringIndex()
    g.add_bulk(ids, blob, ds)
    g.build()
    assert g.proof_wait(30_000) == 2 and g.stat("proof_bad_pairs") == 0 and g.stat("proof_skipped_pairs") == 0
    mixed = g.stat("proof_mixed_pairs")
    r = g.verify_reference()
    assert r["violations"] == 0 and 0 < mixed == r["mixed_pairs"], (mixed, r)
    # damage: swap an adjacent pair that is "mixed" (stage 1 alone would wave it through)
    sa = g.sa()
    mask, bits = g.mask, g.bits
    dsl = ds.astype(np.int64)

    def first_byte_after_common_prefix(i):
        a = [int(sa[i - 1]), int(sa[i])]
        p = [dsl[e & mask] + (e >> bits) for e in a]
        e_ = [dsl[(e & mask) + 1] for e in a]
        l = 0
        while p[0] + l < e_[0] and p[1] + l < e_[1] and blob[p[0] + l] == blob[p[1] + l]:
            l += 1
        if p[0] + l >= e_[0] or p[1] + l >= e_[1]:
            return None
        return int(blob[p[0] + l]), int(blob[p[1] + l])
    k = next(i for i in range(g.size // 2, g.size - 1)
             if (lambda xy: xy is not None and (xy[0] >= 0x80) != (xy[1] >= 0x80))(first_byte_after_common_prefix(i)))
    g.set_option("debug_damage_after_build", k - 1)        # (the hook swaps entries k - 1 and k)
    g.build()
    assert g.proof_wait(60_000) == 3 and g.stat("self_check_fallbacks") == 1
    assert np.array_equal(g.sa(), o.sa())
    g.set_option("reference_compat", 0)
    g.build()
    assert g.proof_wait(30_000) == 2 and g.stat("proof_mixed_pairs") == 0
    g.close()


def test_a_loaded_file_with_entries_out_of_order_is_repaired(tmp_path):
    """cdb_load checks every entry (it names a real suffix) but not the ORDER of the entries: the proof runs behind a load too."""
    import struct
    from coffeedb_amd import capi
    blob, ds = _corpus(seed=21)
    ids = np.arange(len(ds) - 1, dtype=np.int64)
    o = _oracle(blob, ds, ids)
    g = capi.GpuStringIndex()
    g.add_bulk(ids, blob, ds)
    g.build()
    assert g.proof_wait(20_000) == 2
    path = str(tmp_path / "ix.cdb")
    g.save(path)
    raw = bytearray(open(path, "rb").read())
    w = g.sa_width
    sa_off = len(raw) - g.size * w
    k = g.size // 3
    a = bytes(raw[sa_off + k * w: sa_off + (k + 1) * w])
    raw[sa_off + k * w: sa_off + (k + 1) * w] = raw[sa_off + (k + 1) * w: sa_off + (k + 2) * w]
    raw[sa_off + (k + 1) * w: sa_off + (k + 2) * w] = a
    open(path, "wb").write(bytes(raw))
    g2 = capi.GpuStringIndex()
    g2.load(path)
    assert g2.proof_wait(60_000) == 3 and g2.stat("self_check_fallbacks") == 1
    assert np.array_equal(g2.sa(), o.sa())
    g.close()
    g2.close()


def test_a_build_cancels_the_proof_of_the_array_it_replaces():
    """database.cpp:276-280 builds again whenever it is told to: a build must not wait for a proof of the array it is about to
    replace (it waits for at most one slice of the sweep), and the last array is the one that ends up proved."""
    import torch
    from coffeedb_amd import capi
    nd, dl = 1 << 18, 1024
    text = W.random_bytes_torch(nd * dl, 99, device="cuda")
    dsz = W.uniform_docs(nd, dl)
    ids = np.arange(nd, dtype=np.int64)
    torch.cuda.synchronize()
    g = capi.GpuStringIndex()
    times = []
    for _ in range(4):
        t = time.perf_counter()
        g.build_device(text.data_ptr(), dsz, ids)
        times.append(time.perf_counter() - t)
    assert g.stat("proof_runs") == 4
    assert g.proof_wait(30_000) == 2 and g.stat("order_proved") == 1
    v = g.verify()
    assert v["inversions"] == 0 and v["tie_violations"] == 0 and v["invalid_entries"] == 0
    g.close()                                               # (destroying a handle joins its proof)
    g = capi.GpuStringIndex()
    g.build_device(text.data_ptr(), dsz, ids)
    g.close()                                               # ... also one that is still running
    del text


def test_a_process_may_end_while_a_proof_is_still_running():
    """A handle nobody destroyed (an interpreter shutting down): the helper thread is cancelled and joined at process exit, before
    the block caches and the HIP runtime go away — the process ends with exit code 0, not with a crash in a static destructor."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "import torch\n"
            "from coffeedb_amd import capi, workloads as W\n"
            "nd, dl = 1 << 19, 1024\n"
            "text = W.random_bytes_torch(nd * dl, 5, device='cuda'); torch.cuda.synchronize()\n"
            "g = capi.GpuStringIndex()\n"
            "g.build_device(text.data_ptr(), W.uniform_docs(nd, dl), np.arange(nd, dtype=np.int64))\n"
            "assert g.proof_wait(0) in (1, 2)\n"
            "capi.GpuStringIndex.__del__ = lambda self: None   # (nobody closes the handle)\n"
            "print('LEAVING', flush=True)\n" % root)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "LEAVING" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-1500:])


def test_reserve_maps_a_spare_generation_for_the_rebuild_beside_a_serving_index():
    """database.cpp:276-280 builds the next generation while the old one serves: the second build asks for the arrays an index keeps
    once more while they are held.  cdb_reserve therefore maps twins of those blocks as SPARE blocks of the cache (round 6).  Not a
    timing test: the pool counts the requests of 16 MiB and more that had to go to hipMalloc (stat pool_big_mallocs) — after a
    reservation neither the first build nor the second one beside it may cause any; without the reservation both do."""
    import torch
    from coffeedb_amd import capi
    lib = capi.load_library()
    nd, dl = 3 << 18, 1024                                    # 768 MiB: the reservation's 4 % + 64 MiB of head room stay inside the
    blob = W.random_bytes_torch(nd * dl, 31, device="cuda").cpu().numpy()   # cache's 25 % slack for a fitting block
    torch.cuda.empty_cache()
    ds = W.uniform_docs(nd, dl)
    ids = np.arange(nd, dtype=np.int64)

    def two_generations(**opts):
        g1 = capi.GpuStringIndex()
        for k, v in opts.items():
            g1.set_option(k, v)
        m0 = g1.stat("pool_big_mallocs"); g1.build_view(ids, blob, ds); m1 = g1.stat("pool_big_mallocs")
        assert g1.proof_wait(60_000) == 2
        spare = g1.stat("premap_bytes")
        m1b = g1.stat("pool_big_mallocs")                      # (premap_generation maps on the helper thread: not the caller's path)
        g2 = capi.GpuStringIndex(); g2.build_view(ids, blob, ds); m2 = g2.stat("pool_big_mallocs")
        v = g2.verify()
        assert v["inversions"] == 0 and v["tie_violations"] == 0 and v["invalid_entries"] == 0 and v["entry_sum"] == v["expected_entry_sum"]
        g1.close(); g2.close()
        return m1 - m0, m2 - m1b, spare

    lib.cdb_release_cached_memory()
    first, second, _ = two_generations()
    assert first > 0 and second > 0                            # cold cache: the first build maps its working set, the second its arrays
    lib.cdb_release_cached_memory()
    assert lib.cdb_reserve(0, len(blob), nd, bytes(blob[:4096]), 4096) == 0
    lib.cdb_reserve_wait()
    first, second, _ = two_generations()
    assert first == 0 and second == 0, (first, second)        # both generations come out of the reservation
    lib.cdb_release_cached_memory()
    # the same BEHIND a build instead of before it (option premap_generation, off by default: hipMalloc of tens of GB holds up launches)
    first, second, spare = two_generations(premap_generation=1)
    assert first > 0 and second == 0 and spare > 4 * len(blob), (first, second, spare)
    lib.cdb_release_cached_memory()
