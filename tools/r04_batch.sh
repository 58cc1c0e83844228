python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -8
for w in c1 utf8_4g c2 c4shard; do
CDB_TOP=30 timeout 600 python tools/keywidth_ab.py $w 0 3 2>&1 | grep workload | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['workload'], d['build_ms'], {k:v for k,v in d['kernels_ms'].items() if 'update' in k or 'compact' in k}, d['verify'])"
done
