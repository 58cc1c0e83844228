#!/bin/bash
# timing experiments on the segmented bucket-wise build (ablations give WRONG results; only kernel times matter)
W=${1:-utf8_4g}
run() { echo "== $*"; env CDB_OPTIONS=self_check=0 "$@" timeout 300 python tools/big_one.py $W 1 2>&1 | grep -v amdgpu.ids | head -12; }
run CDB_X=0
# (the final pass's ablations are compile-time now: make -C coffeedb_amd/csrc clean all HIPFLAGS+=-DRS_SEG_ABL=1|2|4, then run CDB_X=0)
run CDB_GATHER_ABL=1
run CDB_GATHER_ABL=2
run CDB_GATHER_ABL=4
run CDB_GATHER_ABL=8
