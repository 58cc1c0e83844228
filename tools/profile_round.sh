#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   1. the judged bench line (defaults, unless SKIP_BENCH=1)
#   2. for every workload of WORKLOADS (default: c1 utf8_4g c2 c4shard — the C1 step and the three named targets that run
#      on the bucket-wise >= 2^32 path): --kernel-trace --stats of `bench.py --workload W` (1 warm-up + 2 steps) and two
#      separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950; 1 step each)
# usage: profile_round.sh <tag> <commit>.  Outputs land in gpurun_out/prof_<tag>/<workload>/ ; tools/collect_profiles.sh
# copies the summaries into profiles/.
set -u
TAG=${1:-r03}
COMMIT=${2:-unknown}
WORKLOADS=${WORKLOADS:-"c1 utf8_4g c2 c4shard"}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  ( time python bench.py ) > $OUT/bench_stdout.txt 2> $OUT/bench.err   # two lines: {"bench_detail": ...}, then the bounded contract line
  cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
  tail -n 1 $OUT/bench_stdout.txt | wc -c
  tail -n 1 $OUT/bench_stdout.txt | cut -c1-600
  tail -n 4 $OUT/bench.err
fi
for W in $WORKLOADS; do
  D=$OUT/$W
  mkdir -p $D
  CMD="python bench.py --workload $W --configs none --no-cpu-baseline --no-pcie --no-proof-leg"
  case $W in
    c1) N=1073741824;; utf8_4g) N=4294967296;; c2|c3shard) N=8589934592;; c4shard) N=17179869184;; *) N=0;;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o bench -- $CMD --steps 2 --warmup 1 > $D/bench_traced.json 2> $D/trace.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/pmc_fetch -o bench -- $CMD --steps 1 --warmup 0 > /dev/null 2> $D/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/pmc_write -o bench -- $CMD --steps 1 --warmup 0 > /dev/null 2> $D/pmc_write.err
  python tools/summarize_profile.py $D $D/traffic.json ${TAG}_$W $COMMIT $N > $D/summary.txt 2>&1
  echo "=== $W"; head -24 $D/summary.txt
done
# keep only small artefacts for the merge back
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
