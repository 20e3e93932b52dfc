#!/bin/bash
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06k; mkdir -p $O
for rep in 1 2; do for L in "$@"; do
  [ "$L" = base ] && export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd.so || export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$L.so
  echo "== lib $L" >> $O/k2_ab.txt
  NUTS_AMD_SELFTEST=0 timeout 300 python tools/quick_k2.py 4096 1024 400 200 2>&1 | grep "M1 =" >> $O/k2_ab.txt
done; done
cat $O/k2_ab.txt
