python -m pytest tests/test_gpu_parity.py -x -q -k "bucket or segmented or packed or reference_order or 256" 2>&1 | grep -E "passed|failed|Error|assert" | head -8
CDB_TOP=12 timeout 600 python tools/keywidth_ab.py c4shard 0 2 2>&1 | grep -E "workload|Error|error" | cut -c1-1500
