cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/tools/experiments
O=$GRAFT_REPO_ROOT/gpurun_out/pmc1
mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
i=0
for set in "MemUnitBusy MemUnitStalled WriteUnitStalled" "LDSBankConflict SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA_WR_UNCACHED_32B_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- ./pass_bench 30 1 > $O/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc1'
for f in sorted(glob.glob(O+'/p*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: [0,0])
    for r in csv.DictReader(open(f)):
        if 'rs_onesweep' not in r['Kernel_Name']: continue
        a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    for k,(v,c) in acc.items(): print(os.path.basename(os.path.dirname(os.path.dirname(f))), k, 'per-launch', v/c, 'launches', c)
PY
find $O -name "*.csv" -size +2M -delete
