#!/bin/bash
# A/B of two builds of the library on ONE box: tools/ab_lib.sh <workload> <reps> [lib_a.so] — prints big_one.py's kernel table for the
# product library and for lib_a (default tools/experiments/abl/lib_r05base.so)
W=${1:-c1}; R=${2:-3}; A=${3:-tools/experiments/abl/lib_r05base.so}
echo "=== base ($A)"; CDB_LIB_PATH=$A python tools/big_one.py $W $R 2>&1 | grep -v amdgpu.ids | head -16
echo "=== new"; python tools/big_one.py $W $R 2>&1 | grep -v amdgpu.ids | head -16
