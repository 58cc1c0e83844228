python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -5
for w in utf8_4g c2; do
CDB_OPTS=gen_prebased=0 python tools/alloc_probe.py $w 3 2>&1 | grep build | tail -1 | cut -c1-110
python tools/keywidth_ab.py $w 0 2 2>&1 | grep workload | cut -c1-800
done
