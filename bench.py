#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X string index (BASELINE.json metric:
"SA build GiB/s + batched substring matches/sec").

One step = one pass of the hot path over one batch of synthetic input, per GPU:
    cdb_build_resident : suffix-array construction over the rank's corpus shard (text + document table resident in HBM)
    cdb_query_batch_device : the whole pattern batch against that suffix array (patterns resident in HBM)
    (N > 1) RCCL all-gather merge of the per-shard match lists into one CSR result.
Default workload = BASELINE.json configs[1] ("c1"): 2^20 docs x 1024 B printable ASCII = 1 GiB of text
per GPU, 100 000 patterns of length 4..16 (weak scaling: the corpus grows with N).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline"     — the dominant kernel (radix-sort onesweep pass) timed live with HIP events
  "cpu_baseline" — the CPU restatement of the reference (oracle/, kind "port") timed on this host on a
                   bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (docs per GPU, doc length, patterns, min len, max len)
    "c0": (10_000, 256, 1_000, 4, 16),
    "c1": (1 << 20, 1024, 100_000, 4, 16),
    "mid": (1 << 16, 1024, 100_000, 4, 16),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


class _DevArr:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def merge_on_device(torch, dist, shard, r, npat, world, device, coll_device):
    """Wraps the library's device-resident CSR of this shard and merges it across ranks over RCCL."""
    nrows = int(r.nrows)
    row_ptr = torch.as_tensor(_DevArr(r.d_row_ptr, npat + 1, "<i8"), device=device)
    if nrows:
        ids = torch.as_tensor(_DevArr(r.d_ids, nrows, "<i8"), device=device)
        cnt = torch.as_tensor(_DevArr(r.d_counts, nrows, "<i8"), device=device)
    else:
        ids = cnt = torch.empty(0, dtype=torch.int64, device=device)
    return shard.merge_shard_results(torch, dist, row_ptr.to(coll_device), ids.to(coll_device), cnt.to(coll_device), world)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/traffic_latest.json, written by tools/summarize_profile.py; FETCH_SIZE x2-corrected as
    MI355X_MICROARCH.md prescribes).  bench.py cannot collect PMC counters itself; null if absent."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        k = json.load(open(path))["kernels"][kernel]
        return round(k["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(W, host_text, ndocs_sample, doclen, pb, po, budget_docs):
    """Times the CPU restatement (oracle/cpu_ref.cpp) on a bounded prefix of the same corpus."""
    from oracle import OracleIndex
    cores = os.cpu_count() or 1
    nd = min(ndocs_sample, budget_docs)
    while True:  # grow the sample until the build takes a few seconds (bounded: <= 2^18 docs)
        ds = W.uniform_docs(nd, doclen)
        blob = host_text[: nd * doclen]
        o = OracleIndex()
        o.add_bulk(np.arange(nd, dtype=np.int64), blob, ds)
        t = time.time()
        o.build(0)  # hardware_concurrency threads, as index.cpp:225
        tb = time.time() - t
        if tb >= 4.0 or nd * 4 > min(ndocs_sample, 1 << 18):
            break
        nd *= 4
    # patterns drawn from the sample itself so that the hit structure matches the full-size run
    spb, spo = W.sample_patterns(blob, ds, min(len(po) - 1, 100_000), 4, 16, seed=99)
    t = time.time()
    _, _, _, hits1 = o.query_batch(spb, spo, nthreads=1, want_rows=False)
    tq1 = time.time() - t
    t = time.time()
    o.query_batch(spb, spo, nthreads=cores, want_rows=False)
    tqa = time.time() - t
    npat = len(spo) - 1
    return {
        "value": round(nd * doclen / 2**30 / tb, 6),
        "unit": "GiB/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {nd} docs ({nd * doclen / 2**20:.0f} MiB) of the same corpus, SA build with {cores} threads; "
                  f"{npat} patterns len 4-16 sampled from that prefix",
        "build_s": round(tb, 3),
        "query_patterns_per_s_1thread": round(npat / tq1, 1),
        "query_patterns_per_s_allcores": round(npat / tqa, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c1", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample-docs", type=int, default=1 << 15, help="docs in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo + --share-gpu only exists to "
                         "exercise the N > 1 code path on a one-GPU box)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (testing only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from coffeedb_amd import capi, shard, workloads as W

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X GPU (the HIP path has no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_device = device if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    ndocs, doclen, npat, mmin, mmax = WORKLOADS[args.workload]
    n = ndocs * doclen
    # ---- synthetic shard of this rank (stream = rank => different text per shard), generated in HBM
    text = W.random_bytes_torch(n, 12345, 0x20, 0x7E, stream=rank, device=device)
    doc_start = W.uniform_docs(ndocs, doclen)
    ids = np.arange(ndocs, dtype=np.int64) + rank * ndocs
    host_text = text.cpu().numpy() if rank == 0 else None

    # ---- one pattern batch for every shard (rank 0 samples it from its own text, then broadcast)
    if rank == 0:
        pb, po = W.sample_patterns(host_text, doc_start, npat, mmin, mmax, seed=99)
        meta = torch.tensor([len(pb)], dtype=torch.int64, device=device)
    else:
        meta = torch.zeros(1, dtype=torch.int64, device=device)
    def bcast(t):
        if world == 1:
            return
        if coll_device == t.device:
            dist.broadcast(t, 0)
        else:
            c = t.to(coll_device)
            dist.broadcast(c, 0)
            t.copy_(c)

    bcast(meta)
    nbytes = int(meta.item())
    d_blob = torch.zeros(nbytes + 16, dtype=torch.uint8, device=device)
    d_offs = torch.zeros(npat + 1, dtype=torch.int64, device=device)
    if rank == 0:
        d_blob[:nbytes] = torch.from_numpy(pb).to(device)
        d_offs.copy_(torch.from_numpy(po.astype(np.int64)).to(device))
    bcast(d_blob)
    bcast(d_offs)

    g = capi.GpuStringIndex(device=local_rank)
    g.set_option("profile", 1)

    # the document table is resident like the text (cdb_build_resident): nothing but scalars crosses PCIe in a step
    d_doc_start = torch.from_numpy(doc_start.astype(np.int64)).to(device)
    d_ids = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64)).to(device)

    def step():
        g.build_resident(text.data_ptr(), d_doc_start.data_ptr(), d_ids.data_ptr(), len(ids))
        tb = g.stat("build_ms")
        r = g.query_batch_device(d_blob.data_ptr(), d_offs.data_ptr(), npat, nbytes)
        tq = g.stat("query_ms")
        if world > 1:
            merge_on_device(torch, dist, shard, r, npat, world, device, coll_device)
        return tb, tq, int(r.nhits), int(r.nrows)

    torch.cuda.synchronize()  # inputs complete before the library's own stream touches them
    for _ in range(args.warmup):
        step()
    g.profile_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    build_ms = query_ms = 0.0
    hits = rows = 0
    step_build_ms = []
    for _ in range(args.steps):
        tb, tq, hits, rows = step()
        build_ms += tb
        query_ms += tq
        step_build_ms.append(round(tb, 3))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, build_ms, query_ms], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, build_ms, query_ms = (float(x) for x in t.tolist())

    prof = g.profile()
    dom_name = max((k for k in prof if k.startswith("rs_onesweep_k")), key=lambda k: prof[k]["ms"])
    dom = prof[dom_name]
    dom_avg_ms = dom["ms"] / dom["launches"]
    dom_gbs = dom["bytes"] / dom["launches"] / (dom_avg_ms * 1e-3) / 1e9

    if rank == 0:
        steps = args.steps
        gib_total = world * n * steps / 2**30
        out = {
            "metric": "sa_build_GiB_per_s",
            "value": round(gib_total / elapsed, 4),
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",  # sort keys; text is u8, suffix entries u32 (u64 for corpora the reference stores as u64)
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {ndocs} docs x {doclen} B printable ASCII per GPU ({n / 2**30:.3f} GiB), "
                            f"{npat} patterns len {mmin}-{mmax}; step = SA build + batched query"
                            + (" + RCCL all-gather merge" if world > 1 else ""),
                "docs_per_gpu": ndocs, "doc_len": doclen, "patterns": npat,
            },
            "sa_build_only_GiB_per_s": round(world * n * steps / 2**30 / (build_ms * 1e-3), 4),
            "query_patterns_per_s": round(world * npat * steps / (query_ms * 1e-3), 1),
            "query_hits_per_batch": hits,
            "query_rows_per_batch": rows,
            "build_ms_per_step": step_build_ms,
            "build_stats": {k: g.stat(k) for k in ("rounds", "ext_rounds", "dbl_rounds", "unresolved_after_initial",
                                                   "sort_passes", "sort_passes_skipped", "key_symbols", "symbol_bits",
                                                   "isa_built")},
            "roofline": {
                "bound": "hbm",
                "kernel": dom_name,
                "achieved": round(dom_gbs, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(dom_gbs / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic(dom_name),
                "avg_launch_ms": round(dom_avg_ms, 4),
                "launches": dom["launches"],
                "algorithmic_bytes_per_launch": dom["bytes"] // dom["launches"],
            },
            "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if world == 1 and not args.no_cpu_baseline and args.cpu_sample_docs > 0:
            out["cpu_baseline"] = cpu_baseline(W, host_text, ndocs, doclen, pb, po, args.cpu_sample_docs)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
