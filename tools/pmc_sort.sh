#!/bin/bash
# SQ-side PMC counters of the onesweep kernel (tools/sort_bench.py, one variant), for the diagnostic loop.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_sort; rm -rf $OUT; mkdir -p $OUT
V=${1:-21}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/a -o s -- python tools/sort_bench.py 1073741824 42 $V > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $OUT/b -o s -- python tools/sort_bench.py 1073741824 42 $V > $OUT/b.log 2>&1
python - <<'PY'
import csv, glob, collections
for sub in ("a", "b"):
    f = glob.glob(f"gpurun_out/pmc_sort/{sub}/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv", sub); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if "rs_onesweep" in r["Kernel_Name"] and "1024" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
    for k, (v, c) in sorted(acc.items()): print(f"{k:28s} per launch {v/c:16.0f}  (launches {c})")
PY
tail -3 $OUT/a.log
find $OUT -name "*.csv" -size +4M -delete
