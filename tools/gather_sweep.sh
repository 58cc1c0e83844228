#!/bin/bash
# record-gather launch geometry sweep on a bench workload (kernel times from the library's profiler)
W=${1:-utf8_4g}
for G in 1024 2048 4096 8192; do
  echo "== CDB_GATHER_WGS=$G"; CDB_GATHER_WGS=$G timeout 300 python tools/big_one.py $W 1 2>&1 | grep -E "build_ms|sa_bucket_records"
done
