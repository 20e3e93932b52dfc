#!/bin/bash
# round 6: K2, loads of the U-turn tests in larger groups (NM_CHECK_GROUP = iterations between scheduling barriers; 2 = record)
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06g; mkdir -p $O
for rep in 1 2; do for L in "" _g1 _g4 _g8; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so
  echo "== lib $L" >> $O/k2_groups.txt
  NUTS_AMD_SELFTEST=0 timeout 300 python tools/quick_k2.py 4096 1024 400 200 2>&1 | grep "M1 =" >> $O/k2_groups.txt
done; done
cat $O/k2_groups.txt
