python -m pytest tests/test_gpu_parity.py -x -q -k "self_check or packed" 2>&1 | tail -12 | cut -c1-600
for w in c1 utf8_4g; do CDB_OPTS=self_check=2 python tools/alloc_probe.py $w 3 2>&1 | grep build | tail -1 | cut -c1-260; done
