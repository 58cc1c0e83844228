#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   1. --kernel-trace --stats of the default bench command
#   2. two separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950)
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20
python tools/summarize_profile.py $OUT $OUT/traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small artefacts for the merge back
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
