python -m pytest tests/test_gpu_parity.py -x -q -k "msd_first or last_radix_pass" 2>&1 | tail -15
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --configs none --no-pcie > gpurun_out/b1.json 2> gpurun_out/b1.err; tail -c 800 gpurun_out/b1.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/b1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["build_ms_per_step"]); print(d["kernels_ms"]); print(d["build_stats"]); print(d["roofline"])
PY
