#!/bin/bash
# round 6: K2 with the top-level test's third pair summed by the last level-`depth` test (FUSE_TOP): parity of the (16,1) instantiations, then the rate
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06w; mkdir -p $O; export NUTS_AMD_SELFTEST=0
T=${1:-ft}
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$T.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_every_instantiation.py -x -q -m gpu -k "iid or k2 or K2 or north_star or diag" 2>&1 | tail -3 > $O/parity_$T.txt; cat $O/parity_$T.txt
for rep in 1 2 3; do for L in "" _$T; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so
  echo "== lib $L" >> $O/k2_$T.txt
  timeout 300 python tools/quick_k2.py 4096 1024 400 200 2>&1 | grep "M1 =" >> $O/k2_$T.txt
done; done
cat $O/k2_$T.txt
