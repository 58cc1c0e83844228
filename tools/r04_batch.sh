python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -12 | cut -c1-700
python bench.py --cpu-full-budget 0 --no-cold-start --no-pcie --configs c4shard > gpurun_out/r04_bench5.json 2> gpurun_out/r04_bench5.err; tail -c 200 gpurun_out/r04_bench5.json; tail -3 gpurun_out/r04_bench5.err
