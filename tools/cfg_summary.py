"""Prints the per-configuration figures of a bench.py JSON line (stdin or file): build ms, rates, kernel table."""
import json, sys
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
line = [l for l in src.read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print("main:", d["config"]["workload"][:60], "ms/step", d["ms_per_step"], "value", d["value"], "roof", d["roofline"]["kernel"], d["roofline"]["frac"])
for name, c in (d.get("configs") or {}).items():
    if "error" in c:
        print(name, "ERROR", c["error"]); continue
    print(f"{name}: build_ms {c['build_ms']} first {c['first_build_ms_incl_allocation']} GiB/s {c['sa_build_GiB_per_s']} kernels_ms {c['build_kernels_ms']} "
          f"algB/suffix {c['build_algorithmic_bytes_per_suffix']} GB/s(kern) {c['build_algorithmic_GBps_over_kernel_time']}")
    bs = c["build_stats"]
    print("   stats", {k: bs.get(k) for k in ("segmented", "bucket_groups", "sort_passes", "key_symbols", "bucket_low_digits", "unresolved_after_initial", "rounds")})
    print("   roof", c["roofline"]["kernel"], c["roofline"]["frac"], "avg_ms", c["roofline"]["avg_launch_ms"], "launches", c["roofline"]["launches"])
    print("   kernels", c["kernels_ms"])
    print("   query_ms", c["query_ms"], "verify", c["verify"])
