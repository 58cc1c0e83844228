"""bench.py --gpus N must run N ranks by itself: without a launcher around it the script re-runs itself
under torch.distributed.run, and it refuses to print a line for fewer GPUs than it was asked for."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_plan_decisions():
    argv = ["--gpus", "4", "--steps", "3"]
    assert bench.launch_plan(1, {}, []) is None                                  # N = 1: run in place
    assert bench.launch_plan(4, {"WORLD_SIZE": "4"}, argv) is None              # already a rank of a launcher
    plan = bench.launch_plan(4, {}, argv, device_count=8)
    assert plan[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert plan[plan.index("--nproc-per-node") + 1] == "4" and plan[plan.index("--master-addr") + 1] == "127.0.0.1"
    assert plan[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and plan[-len(argv):] == argv
    with pytest.raises(SystemExit) as e:                                         # fewer GPUs than asked for: no line at all
        bench.launch_plan(8, {}, argv, device_count=1)
    assert "only 1 GPU(s) visible" in str(e.value)
    assert bench.launch_plan(2, {}, argv, share_gpu=True, device_count=1) is not None   # one-GPU test mode
    with pytest.raises(SystemExit):
        bench.launch_plan(0, {}, argv)


def test_gpus_without_devices_exits_nonzero_and_prints_no_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""   # (no GPU here anyway; on a GPU box this hides them)
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "n_gpus" not in p.stdout and "GPU(s) visible" in p.stderr


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr and "n_gpus" not in p.stdout


@pytest.mark.gpu
def test_gpus_2_self_launches_two_ranks_on_one_gpu():
    """No torchrun around it: the line must say n_gpus 2 (gloo rendezvous, both ranks on cuda:0 — the one-GPU stand-in
    for the driver's `python bench.py --gpus N`)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                        "--workload", "mid", "--steps", "2", "--warmup", "1", "--configs", "none", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and len(line["rows_per_rank"]) == 2 and line["value"] > 0


def _canned_detail():
    """A record shaped like rank 0's full output, with every free-text field far longer than bench.py ever writes."""
    long = "x" * 5000
    block = {"workload": long, "build_ms": [261.4, 262.6], "sa_build_GiB_per_s": 30.6, "build_frac_of_hbm_peak_over_wall_time": 0.49,
             "roofline": {"frac": 0.61, "kernel": long}, "query_patterns_per_s": 5.4e7, "peak_hbm_bytes": 205616407552, "dtype": "u64",
             "verify": {"invalid_entries": 0, "inversions": 0, "tie_violations": 0, "entry_sum_ok": True},
             "build_stats": {f"k{i}": float(i) for i in range(60)}, "kernels_ms": {f"kernel_{i}": 1.0 for i in range(40)},
             "cpu_baseline": {"value": 0.0078, "query_patterns_per_s_allcores": 1.7e6, "sample": long}, "hbm_note": long}
    return {
        "metric": "sa_build_GiB_per_s", "value": 28.1234, "unit": "GiB/s", "n_gpus": 1, "steps": 5, "warmup": 2, "ms_per_step": 284.5,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": long, "docs_per_gpu": 1 << 23, "bytes_per_gpu": 1 << 33, "patterns": 1_000_000, "junk": {"a": long}},
        "commit": "abcdef0", "merge": long, "rccl_ranks": 8, "rows_per_rank": [10**9] * 8,
        "mg_selfcheck": {"ok": True, "world": 8, "transport": long, "cdb_comm_world": 8, "devices_distinct": True, "bus_ids": ["05:00"] * 8,
                         "merged_rows": 8 * 10**9, "sum_of_local_rows": 8 * 10**9, "patterns": 1000, "local_rows": 5},
        "sa_build_only_GiB_per_s": 30.6, "sa_build_GiB_per_s_incl_h2d": 17.1, "query_patterns_per_s": 5.4e7, "query_hits_per_batch": 105508206,
        "query_rows_per_batch": 104177337, "build_ms_per_step": [262.123] * 50, "build_stats": {f"k{i}": float(i) for i in range(60)},
        "roofline": {"bound": "hbm", "kernel": "rs_seg_k32_v32_w16_t16384", "achieved": 4889.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.6111,
                     "traffic": 44846505174, "avg_launch_ms": 35.1, "launches": 20, "algorithmic_bytes_per_launch": 42949672960,
                     "traffic_source": {"profile": "r06a_c2", "commit": "abcdef0", "source": long}, "traffic_note": long},
        "query_roofline": {"patterns_per_s": 5.4e7, "probes_per_s": 3.5e9, "achieved": 595.0, "unit": "GB/s", "levels": 33, "note": long},
        "kernels_ms": {f"kernel_{i}": 1.0 for i in range(40)}, "single_query_us": {"note": long}, "pcie_inclusive": {"note": long},
        "cold_start": {"note": long}, "configs": {n: dict(block) for n in ("c1", "c0", "utf8_4g", "c4shard", "c3", "c4")},
        "cpu_baseline": {"value": 0.0041, "unit": "GiB/s", "cores": 32, "kind": "port", "sample": long, "build_s": 30.5, "host_threads_available": 256,
                         "query_patterns_per_s_1thread": 1.2e5, "query_patterns_per_s_allcores": 1.6e6, "c0_build_MiB_per_s_by_threads": {"8": 1.0}},
        "c1_sa_bit_exact": True, "c1_rows_bit_exact": True, "c1_bit_exact_check": {"note": long},
    }


def test_final_line_is_bounded_and_strict_json():
    """a 21.6 KB line came back from the driver as `parsed: null`.  The final line stays under 8 KB whatever the
    detail record holds, parses under a strict parser, and carries the contract's keys."""
    def no_constants(x):
        raise ValueError(f"non-finite constant {x} in the bench line")

    s = bench.short_line(_canned_detail())
    line = json.dumps(s, allow_nan=False)
    assert len(line) < 8192 and len(line) <= bench.SHORT_LINE_LIMIT + 1200, len(line)
    back = json.loads(line, parse_constant=no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "configs", "c1_sa_bit_exact", "c1_rows_bit_exact", "mg_selfcheck", "rccl_ranks"):
        assert k in back, k
    assert set(back["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(back["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert back["config"]["workload"] and "junk" not in back["config"]
    for name, d in back["configs"].items():   # the digest: numbers only
        assert all(not isinstance(v, (dict, list)) for v in d.values()), name
        assert {"build_ms", "sa_build_GiB_per_s"} <= set(d)
    assert len(json.dumps(back["configs"])) < 2600
    # a record with nothing optional in it still yields the contract's keys
    bare = bench.short_line({"metric": "sa_build_GiB_per_s", "value": 1.0, "unit": "GiB/s", "n_gpus": 2, "steps": 1, "warmup": 0,
                             "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
                             "data": "synthetic", "config": {"workload": "w"}, "roofline": None, "cpu_baseline": None})
    assert bare["cpu_baseline"] is None and bare["roofline"] is None and bare["configs"] == {}


def test_emit_prints_the_short_line_last(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(_canned_detail())
    lines = [l for l in capsys.readouterr().out.splitlines() if l]
    assert len(lines) == 2 and lines[0].startswith('{"bench_detail"') and len(lines[1]) < 8192
    assert json.loads(lines[1])["metric"] == "sa_build_GiB_per_s"
    assert json.load(open(tmp_path / "bench_detail.json"))["cold_start"]


@pytest.mark.gpu
def test_gpus_8_dress_rehearsal_on_one_gpu():
    """The first 8-GPU run of bench.py is the driver's.  Here the same command runs 8 ranks that share
    cuda:0 (gloo rendezvous) on BASELINE config 4 in miniature — 8 x 256 MiB of valid UTF-8, 10^6 patterns, the counts merge and
    the $correlation ranking over the shards: rank arithmetic, row bases, the self check before the timed region, the per-config
    block and the bounded final line are the ones the driver's `--gpus 8` run will execute."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--share-gpu",
                        "--workload", "c4mini", "--steps", "2", "--warmup", "1", "--configs", "c4mini", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines[-1]) < 8192
    line = json.loads(lines[-1])
    detail = json.loads(lines[-2])["bench_detail"]
    assert line["n_gpus"] == 8 and line["steps"] == 2 and line["value"] > 0 and line["cpu_baseline"] is None
    sc = line["mg_selfcheck"]
    assert sc["ok"] and sc["world"] == 8 and len(sc["bus_ids"]) == 8 and sc["merged_rows"] == sc["sum_of_local_rows"]
    assert len(line["rows_per_rank"]) == 8 and sum(line["rows_per_rank"]) == line["merged_rows"] > 0
    blk = detail["configs"]["c4mini"]
    assert "error" not in blk, blk
    assert len(blk["rows_per_rank"]) == 8 and sum(blk["rows_per_rank"]) == blk["merged_rows"] > 0
    assert blk["ranked"]["rows"] == 1000 and blk["ranked"]["global"]["rows"] == 1000
    assert blk["all_ranks"]["n_gpus"] == 8 and line["configs"]["c4mini"]["aggregate_GiB_per_s"] > 0
