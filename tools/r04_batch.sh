python -m pytest tests/test_gpu_parity.py -x -q -k "packed or records_generators" 2>&1 | grep -E "passed|failed|Error|assert" | head -8
for a in 0 1; do
CDB_SWEEP_ABL=$a CDB_TOP=8 CDB_OPTS=self_check=0 timeout 120 python tools/keywidth_ab.py utf8_4g 0 1 2>&1 | grep -E "workload|Error|error" | grep -o '"kernels_ms.*' | grep -o '"rs_sweep[^,]*'
done
CDB_TOP=6 timeout 600 python tools/keywidth_ab.py c4shard 0 2 2>&1 | grep -E "workload|Error|error" | grep -o '"build_ms.*' | cut -c1-700
