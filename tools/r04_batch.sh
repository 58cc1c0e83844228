set -x
python -m pytest tests/test_gpu_fullsize.py -k "rebuild_8gib" -x -q -s 2>&1 | tail -15 | cut -c1-900
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sort.py -x -q 2>&1 | tail -5
python bench.py --cpu-full-budget 0 --configs c0,utf8_4g > gpurun_out/r04_bench3.json 2> gpurun_out/r04_bench3.err; tail -c 300 gpurun_out/r04_bench3.json; tail -3 gpurun_out/r04_bench3.err
python tools/single_breakdown.py 2>&1 | tail -6
