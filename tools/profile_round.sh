#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   1. the judged bench line (defaults)
#   2. --kernel-trace --stats of the C1 step, and of the bucket-wise (>= 2^32) build on the C3 shard shape (8 GiB)
#   3. two separate --pmc passes each (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950)
# usage: profile_round.sh <tag> <commit>.  Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
set -u
TAG=${1:-r02}
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
C1="python bench.py --configs none --no-cpu-baseline --no-pcie"
BIG="python bench.py --workload c3shard --configs none --no-cpu-baseline --no-pcie"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $C1 --steps 3 --warmup 1 > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- $C1 --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- $C1 --steps 1 --warmup 0 > /dev/null 2> $OUT/pmc_write.err
python tools/summarize_profile.py $OUT $OUT/traffic.json $TAG $COMMIT 1073741824 > $OUT/summary.txt 2>&1
mkdir -p $OUT/big
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/big/trace -o bench -- $BIG --steps 2 --warmup 1 > $OUT/big/bench_traced.json 2> $OUT/big/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/big/pmc_fetch -o bench -- $BIG --steps 1 --warmup 0 > /dev/null 2> $OUT/big/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/big/pmc_write -o bench -- $BIG --steps 1 --warmup 0 > /dev/null 2> $OUT/big/pmc_write.err
python tools/summarize_profile.py $OUT/big $OUT/big/traffic.json ${TAG}_big $COMMIT 8589934592 > $OUT/big/summary.txt 2>&1
cat $OUT/summary.txt | head -40
cat $OUT/big/summary.txt | head -40
# keep only small artefacts for the merge back
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*.csv" | head -20
