for cfg in "1024 32" "256 8" "256 16" "256 4"; do set -- $cfg
CDB_HIST_NT=$1 CDB_HIST_SPAN=$2 CDB_TOP=8 timeout 300 python tools/keywidth_ab.py c4shard 0 2 2>&1 | grep workload | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4shard $cfg', d['build_ms'], d['kernels_ms'].get('rs_seg_hist'), d['verify'])"
done
for cfg in "1024 32" "256 8" "256 16"; do set -- $cfg
CDB_HIST_NT=$1 CDB_HIST_SPAN=$2 CDB_TOP=8 timeout 300 python tools/keywidth_ab.py c2 0 2 2>&1 | grep workload | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 $cfg', d['build_ms'], d['kernels_ms'].get('rs_seg_hist'), d['verify'])"
done
