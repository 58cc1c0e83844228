#!/bin/bash
# SQ / TCP counters of the build's kernels on the real builds (bench.py --workload W, one step), one rocprofv3 pass per counter
# set (gfx950 has few SQ counter slots per pass).  usage: pmc_build_sq.sh <tag> [workloads...]; prints per kernel family and
# counter the average per launch; the table lands in gpurun_out/sq_<tag>.txt (copy into profiles/).
set -u
TAG=${1:-r05}
shift
WORKLOADS=${*:-"c1 utf8_4g"}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sq_$TAG
rm -rf $O; mkdir -p $O
for W in $WORKLOADS; do
  mkdir -p $O/$W; i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
             "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
             "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
             "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$W/p$i -o p -- \
      python bench.py --workload $W --configs none --no-cpu-baseline --no-pcie --steps 1 --warmup 0 > $O/$W/p$i.log 2>&1
  done
done
python3 - $O "$WORKLOADS" > gpurun_out/sq_$TAG.txt <<'PY'
import csv, glob, collections, sys, re
O, wl = sys.argv[1], sys.argv[2].split()
def fam(n):
    head = n.split('(')[0]
    m = re.match(r'(?:void )?(?:cdb::)?(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)', head.strip())
    k = m.group(1) if m else head[:40]
    if k == 'rs_onesweep_kernel':   # the pass's role is in its template arguments
        for tag, name in (('SegFinalKeepMsdArgs', 'rs_seg_final_keepmsd'), ('SegFinalKeepArgs', 'rs_final_keep'), ('SegFinalArgs', 'rs_seg_final'),
                          ('TextGenPair', 'rs_gen_pair'), ('TextGenRec', 'rs_gen_records'), ('TextGen', 'rs_gen'), ('SegArgs', 'rs_seg')):
            if tag in n:
                return name
    return k
for W in wl:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(f'{O}/{W}/p*/**/*counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            k = fam(r['Kernel_Name'])
            if not (k.startswith('rs_') or k.startswith('sa_')): continue
            # distinguish template instantiations by LDS / workgroup size where the family name is shared
            key = (k, r.get('Workgroup_Size', ''), r.get('LDS_Block_Size', ''), r['Counter_Name'])
            a = acc[key]; a[0] += float(r['Counter_Value']); a[1] += 1
    print(f'== {W}: counter sums per launch (kernel, workgroup, LDS bytes)')
    for (k, wg, lds, c), (v, cnt) in sorted(acc.items()):
        print(f'{W} {k} wg{wg} lds{lds} {c} per-launch {v / cnt:.5g} launches {cnt}')
PY
find $O -name "*.csv" -size +2M -delete
find $O -name "*agent_info.csv" -delete
head -5 gpurun_out/sq_$TAG.txt; wc -l gpurun_out/sq_$TAG.txt; tail -3 $O/c1/p1.log
