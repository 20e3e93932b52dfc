#!/bin/bash
# round 6: where K2's 12 k cycles per leapfrog go — timing-only builds of the (16,1) iid kernel (results wrong by construction):
# xs: no scratch stores (NM_X_NO_SCRATCH_STORES); xl: the level >= 2 / top-level U-turn tests read registers instead of scratch slots (NM_X_NO_TEST_LOADS);
# xsl: both; xm: no merge arithmetic (NM_X_NO_MERGE_MATH).  Rates are per leapfrog actually taken (the trees differ).
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06d; mkdir -p $O
for L in "" _xs _xl _xm; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so
  echo "== lib $L" >> $O/k2_where.txt
  NUTS_AMD_SELFTEST=0 timeout 300 python tools/quick_k2.py 4096 1024 400 200 2>&1 | grep "M1 =\|sample s" >> $O/k2_where.txt
done
cat $O/k2_where.txt
