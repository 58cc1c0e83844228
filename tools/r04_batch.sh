python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -5
python -m pytest tests/test_gpu_fullsize.py -x -q -k "c1_full" 2>&1 | grep -E "passed|failed|Error|assert" | head -5
python tools/keywidth_ab.py c1 0 3 2>&1 | grep workload | cut -c1-700
