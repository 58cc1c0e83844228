#!/bin/bash
# The query half of the metric against a MEASURED ceiling:
#   1. tools/experiments/gather_ceiling: random 64-byte-sector reads, dependent and independent, over 0.125 / 4 / 40 / 100 GB and
#      1-32 waves per CU -> G sectors/s by HIP events
#   2. the same program under rocprofv3 --pmc (memory-side read requests of the L2, L2 hits / misses, L1 -> L2 read requests) at
#      40 GB: what ONE random sector read costs in those counters (the calibration the guide asks for)
#   3. the same counters over bench.py's query kernels (q_*) at c1 / c2 / c4shard: memory-side requests per batch
# usage (through gpurun): tools/query_pmc.sh <tag> [workloads...]; results in gpurun_out/qpmc_<tag>/ and a table in
# gpurun_out/qpmc_<tag>.txt (copied to profiles/ by hand)
set -u
TAG=${1:-r06}
shift
WORKLOADS=${*:-"c1 c2"}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/qpmc_$TAG
rm -rf $O; mkdir -p $O
G=tools/experiments/gather_ceiling
[ -x $G ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/experiments/gather_ceiling.hip -o $G
timeout 600 $G 0.125 4 40 100 > $O/gather_ceiling.jsonl 2> $O/gather_ceiling.err
SET="TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"
GATHER_WPC=16 timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/gather_pmc -o p -- $G 40 > $O/gather_pmc.jsonl 2> $O/gather_pmc.err
for W in $WORKLOADS; do
  timeout 900 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/$W -o p -- \
    python bench.py --workload $W --configs none --no-cpu-baseline --no-pcie --steps 1 --warmup 0 > $O/$W.bench.out 2> $O/$W.err
done
python3 - $O "$WORKLOADS" > gpurun_out/qpmc_$TAG.txt <<'PY'
import csv, glob, collections, json, re, sys
O, wl = sys.argv[1], sys.argv[2].split()
def fam(n):
    m = re.match(r'(?:void )?(?:cdb::)?(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)', n.split('(')[0].split('<')[0].strip())
    return m.group(1) if m else n[:40]
def table(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[fam(r['Kernel_Name'])][r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
    return acc
print('== gather_ceiling at 40 GB under the counters (per launch, 16 waves per CU: 256 x 16 x 64 lanes x 512 loads = 134 217 728 sector reads per launch)')
for k, cs in sorted(table(f'{O}/gather_pmc').items()):
    if k.startswith('gather'):
        print(k, {c: round(v[0] / v[1]) for c, v in sorted(cs.items())}, 'launches', max(v[1] for v in cs.values()))
for W in wl:
    print(f'== {W}: query kernels, totals over the run (one step: one batch)')
    for k, cs in sorted(table(f'{O}/{W}').items()):
        if k.startswith('q_') or 'query' in k:
            print(W, k, {c: round(v[0]) for c, v in sorted(cs.items())}, 'launches', max(v[1] for v in cs.values()))
    try:
        line = [l for l in open(f'{O}/{W}.bench.out').read().splitlines() if l.startswith('{')][-1]
        j = json.loads(line)
        print(W, 'bench (under the profiler):', {k: j.get(k) for k in ('query_patterns_per_s', 'query_hits_per_batch', 'query_rows_per_batch')})
    except Exception as e:
        print(W, 'bench line unreadable', e)
PY
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +16M -delete
find $O -name "*agent_info.csv" -delete
cat gpurun_out/qpmc_$TAG.txt | cut -c1-400
du -sh $O
