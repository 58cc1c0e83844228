W=${1:-utf8_4g}
python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --configs none --no-pcie > gpurun_out/b2_$W.json 2> gpurun_out/b2_$W.err; tail -c 600 gpurun_out/b2_$W.err
python - <<PY
import json
d=json.loads(open("gpurun_out/b2_$W.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["build_ms_per_step"])
print({k: round(v/3,2) for k,v in d["kernels_ms"].items()})
print(d["build_stats"]); print(d["roofline"])
PY
