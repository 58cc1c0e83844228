python -m pytest tests/test_gpu_parity.py -x -q -k "msd or variant or xcd" 2>&1 | grep -E "passed|failed|Error|assert" | head -8
CDB_TOP=8 timeout 300 python tools/keywidth_ab.py c1 0 3 2>&1 | grep workload | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['workload'], d['build_ms'], d['kernels_ms'], d['verify'])"
