set -x
python bench.py --force-merge --steps 2 --warmup 1 --configs none --no-cpu-baseline --no-pcie > gpurun_out/mg1.json 2> gpurun_out/mg1.err; tail -c 1500 gpurun_out/mg1.json; tail -3 gpurun_out/mg1.err
python bench.py --gpus 2 --backend gloo --share-gpu --workload mid --steps 2 --warmup 1 --configs c0 --no-cpu-baseline > gpurun_out/mg2.json 2> gpurun_out/mg2.err; tail -c 2500 gpurun_out/mg2.json; tail -5 gpurun_out/mg2.err
python bench.py --workload c3 --scaling strong --steps 1 --warmup 1 --configs none --no-cpu-baseline --no-pcie > gpurun_out/mg3.json 2> gpurun_out/mg3.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/mg3.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','scaling','n_gpus')}, d['config']['workload'], d['roofline']['kernel'], d['roofline']['frac'], d['build_ms_per_step'])
PY
tail -3 gpurun_out/mg3.err
