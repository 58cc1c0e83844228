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
python3 - $O "$WORKLOADS" $TAG > gpurun_out/qpmc_$TAG.txt <<'PY'
import csv, glob, collections, json, re, sys
O, wl, tag = sys.argv[1], sys.argv[2].split(), sys.argv[3]
def fam(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'(?:void )?(?:cdb::)?([A-Za-z0-9_]+)', n.strip())
    return m.group(1) if m else n[:40]
def launches(d):
    """{family: [ {counter: value} per dispatch, in dispatch order ]}"""
    per = collections.defaultdict(dict)
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            per[(int(r['Dispatch_Id']), fam(r['Kernel_Name']))][r['Counter_Name']] = float(r['Counter_Value'])
    out = collections.defaultdict(list)
    for (did, k), cs in sorted(per.items()):
        out[k].append(cs)
    return out
summary = {"profile": f"qpmc_{tag}", "source": "tools/query_pmc.sh: rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum over "
           "bench.py --workload W --steps 1 --warmup 0; gather_ceiling.hip by HIP events", "ceiling": {}, "calibration": {}, "batches": {}}
best = collections.defaultdict(float)
for l in open(f'{O}/gather_ceiling.jsonl'):
    if l.startswith('{'):
        r = json.loads(l)
        if 'G_sectors_per_s' in r:
            best[str(r['working_set_GB'])] = max(best[str(r['working_set_GB'])], r['G_sectors_per_s'])
summary["ceiling"] = {"G_sectors_per_s_by_working_set_GB": dict(best),
                      "note": "best of dependent / independent random 64-byte-sector reads over 1-32 waves per CU (saturates at 4 waves per CU)"}
print('== gather_ceiling: best G sectors/s by working set (GB):', dict(best))
print('== gather_ceiling at 40 GB under the counters (per launch, 16 waves per CU: 256 x 16 x 64 lanes x 512 loads = 134 217 728 sector reads per launch)')
for k, ls in sorted(launches(f'{O}/gather_pmc').items()):
    if k.startswith('gather'):
        avg = {c: round(sum(x.get(c, 0) for x in ls) / len(ls)) for c in ls[0]}
        print(k, avg, 'launches', len(ls))
        summary["calibration"][k] = dict(avg, sector_reads_per_launch=134217728)
for W in wl:
    print(f'== {W}: query kernels, per launch')
    try:
        line = [l for l in open(f'{O}/{W}.bench.out').read().splitlines() if l.startswith('{')][-1]
        npat = json.loads(line)["config"]["patterns"]
    except Exception as e:
        npat = None
    fams = {}
    for k, ls in sorted(launches(f'{O}/{W}').items()):
        if not k.startswith('q_'):
            continue
        # the batch launches of the timed step are the LARGEST ones of a family (the lone-keyword probes and the short tail are small)
        top = max(x.get('TCP_TCC_READ_REQ_sum', 0) for x in ls)
        big = [x for x in ls if x.get('TCP_TCC_READ_REQ_sum', 0) >= 0.5 * top]
        avg = {c: round(sum(x.get(c, 0) for x in big) / len(big)) for c in big[0]}
        print(W, k, avg, 'launches', len(ls), 'of which batch-sized', len(big))
        fams[k] = dict(avg, launches_averaged=len(big))
    summary["batches"][W] = {"patterns": npat, "kernels": fams}
json.dump(summary, open(f'{O}/query_counters.json', 'w'), indent=1)
PY
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +16M -delete
find $O -name "*agent_info.csv" -delete
cat gpurun_out/qpmc_$TAG.txt | cut -c1-400
du -sh $O
