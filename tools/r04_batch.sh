CDB_TOP=8 timeout 120 python tools/keywidth_ab.py utf8_4g 0 2 2>&1 | grep -E "workload|Error|error" | grep -o '"build_ms.*' | cut -c1-500
