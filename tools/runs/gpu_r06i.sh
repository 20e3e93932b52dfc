#!/bin/bash
# round 6: K2 with gradient-free points + the doubling's first leaf in registers (NOG / FD): parity, then rate against the record
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06i; mkdir -p $O
T=${1:-nog}
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$T.so
NUTS_AMD_SELFTEST=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_every_instantiation.py -x -q -m gpu -k "iid or k2 or K2 or north_star or diag" 2>&1 | tail -4 > $O/parity_$T.txt; cat $O/parity_$T.txt
for rep in 1 2; do for L in "" _$T; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so
  echo "== lib $L" >> $O/k2_$T.txt
  NUTS_AMD_SELFTEST=0 timeout 300 python tools/quick_k2.py 4096 1024 400 200 2>&1 | grep "M1 =" >> $O/k2_$T.txt
done; done
cat $O/k2_$T.txt
