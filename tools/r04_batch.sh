CDB_FUZZ_N=300 CDB_FUZZ_SEG_N=500 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | grep -E "passed|failed|Error|assert|seed" | head -12
