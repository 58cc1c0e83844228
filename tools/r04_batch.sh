python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -8
for w in utf8_4g c2; do
CDB_TOP=8 timeout 600 python tools/keywidth_ab.py $w 0 2 2>&1 | grep -E "workload|Error|error" | cut -c1-1200
done
