#!/bin/bash
# copies the summaries of gpurun_out/prof_<tag>/ into profiles/ (tracked): <tag>_<workload>_{summary.txt,kernel_stats.csv,
# traffic.json,bench_traced.json}; the C1 traffic becomes profiles/traffic_latest.json (what bench.py quotes)
set -u
TAG=${1:-r03}
SRC=gpurun_out/prof_$TAG
[ -f $SRC/bench_stdout.txt ] && cp $SRC/bench_stdout.txt profiles/${TAG}_bench_stdout.txt
[ -f $SRC/bench_stdout.txt ] && tail -n 1 $SRC/bench_stdout.txt > profiles/${TAG}_bench_line.json
[ -f $SRC/bench_detail.json ] && cp $SRC/bench_detail.json profiles/${TAG}_bench_detail.json
for D in $SRC/*/; do
  W=$(basename $D)
  [ -f $D/summary.txt ] && cp $D/summary.txt profiles/${TAG}_${W}_summary.txt
  [ -f $D/traffic.json ] && cp $D/traffic.json profiles/${TAG}_${W}_traffic.json
  [ -f $D/bench_traced.json ] && cp $D/bench_traced.json profiles/${TAG}_${W}_bench_traced.json
  S=$(find $D/trace -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && cp $S profiles/${TAG}_${W}_kernel_stats.csv
done
[ -f $SRC/c1/traffic.json ] && cp $SRC/c1/traffic.json profiles/traffic_latest.json
for W in utf8_4g c2 c4shard c3shard; do
  [ -f $SRC/$W/traffic.json ] && cp $SRC/$W/traffic.json profiles/traffic_latest_$W.json
done
ls profiles | grep "^${TAG}_"
