python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -k "not 8gib and not shard" 2>&1 | grep -E "passed|failed|Error|assert" | head -8
