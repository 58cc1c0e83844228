# PMC counters of the C1 build's kernels (usage: pmc_build.sh <kernel substring>), one rocprofv3 pass per counter set
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
K=${1:-textgen}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_build
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python tools/pass_ab.py 31 > $O/p$i.log 2>&1
done
python3 - "$K" <<'PY'
import csv, glob, os, collections, sys
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_build'
for f in sorted(glob.glob(O+'/p*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: [0,0])
    for r in csv.DictReader(open(f)):
        if sys.argv[1] not in r['Kernel_Name']: continue
        a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    for k,(v,c) in acc.items(): print(k, 'per-launch %.4g' % (v/c), 'launches', c)
PY
find $O -name "*.csv" -size +1M -delete
