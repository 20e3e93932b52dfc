#!/bin/bash
# round 6: K2 without the tree's scratch stores (timing only, results wrong): rate and phase profile
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06f; mkdir -p $O
for L in "" _xs; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so
  echo "== lib $L" >> $O/k2_nostores.txt
  NUTS_AMD_SELFTEST=0 timeout 300 python tools/quick_k2.py 4096 1024 400 200 2>&1 | grep "M1 =\|sample s" >> $O/k2_nostores.txt
done
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_xsp.so
NUTS_AMD_SELFTEST=0 timeout 300 python tools/prof_phases.py 4096 1024 400 200 2>&1 | sed -n '/sampling/,$p' >> $O/k2_nostores.txt
cat $O/k2_nostores.txt
