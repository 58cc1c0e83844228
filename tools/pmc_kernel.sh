#!/bin/bash
# SQ counters of ONE kernel family of a build (tools/big_one.py <workload> 1): usage pmc_kernel.sh <workload> <kernel substring> [lib.so]
W=${1:-c1}; K=${2:-rs_sweep_msd}; L=${3:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmck; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  CDB_LIB_PATH=$L timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python tools/big_one.py $W 1 > $O/p$i.log 2>&1
done
python3 - "$K" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob('gpurun_out/pmck/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, c) in sorted(acc.items()): print(f'{sys.argv[1]} {k} per-launch {v / c:.5g} launches {c}')
PY
rm -rf $O
