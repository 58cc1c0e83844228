#!/bin/bash
# timing experiments on the segmented bucket-wise build (ablations give WRONG results; only kernel times matter).
# Every ablation is a COMPILE-time choice of a separate library (never a switch of the product build):
#   make -C coffeedb_amd/csrc clean all HIPFLAGS+="-DRS_GATHER_ABL=1"   (or -DRS_SWEEP_ABL=n / -DRS_SEG_ABL=n / -DRS_GEN_ABL=n)
#   cp coffeedb_amd/csrc/libcoffeedb_gpu.so tools/experiments/abl/lib_<name>.so ; rebuild the product library
#   CDB_LIB_PATH=tools/experiments/abl/lib_<name>.so CDB_OPTIONS=self_check=0 python tools/big_one.py <workload> 1
W=${1:-utf8_4g}
run() { echo "== $*"; env CDB_OPTIONS=self_check=0 "$@" timeout 300 python tools/big_one.py $W 1 2>&1 | grep -v amdgpu.ids | head -12; }
run CDB_X=0
for L in tools/experiments/abl/lib_*.so; do [ -f "$L" ] && run CDB_LIB_PATH=$L; done
