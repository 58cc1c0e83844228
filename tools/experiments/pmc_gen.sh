# SQ counters of the generated pass of the C1 initial sort (gen_bench: rs_onesweep_kernel<..., TextGen, ...>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/tools/experiments
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_gen
mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- ./gen_bench_0 30 1 1 > $O/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_gen'
for f in sorted(glob.glob(O+'/p*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: [0,0])
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        fam='gen' if 'TextGen' in n else ('keep' if 'SegFinalKeepArgs' in n else ('seg' if 'SegArgs' in n else None))
        if not fam: continue
        a=acc[(fam,r['Counter_Name'])]; a[0]+=float(r['Counter_Value']); a[1]+=1
    for (fam,k),(v,c) in sorted(acc.items()): print(fam, k, 'per-launch %.4g' % (v/c), 'launches', c)
PY
rm -rf $O/p*/
