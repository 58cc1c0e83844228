python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sort.py -x -q 2>&1 | tail -4 | cut -c1-500
python bench.py --cpu-full-budget 0 --no-cold-start --no-pcie --configs c2,utf8_4g > gpurun_out/r04_bench7.json 2> gpurun_out/r04_bench7.err; tail -2 gpurun_out/r04_bench7.err
