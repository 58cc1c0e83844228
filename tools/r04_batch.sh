for L in "" /root/repo/tools/experiments/abl/lib_u1.so /root/repo/tools/experiments/abl/lib_u4.so ""; do
CDB_LIB_PATH=$L CDB_TOP=6 timeout 300 python tools/keywidth_ab.py c1 0 3 2>&1 | grep workload | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['build_ms'], d['kernels_ms'].get('rs_sweep_msd'), d['verify'])"
done
