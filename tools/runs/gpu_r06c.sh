#!/bin/bash
# round 6: regenerated known answers (exact inputs) on the library of record; K2 A/B: batched merges on the (16,1) tiling (b16) against the record; GPU suite
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06c; mkdir -p $O
( time NUTS_AMD_SELFTEST=0 python -c "import nuts_rs_amd.selftest as s; print(s.run_all(), 'instantiations ok')" ) > $O/run_all.txt 2>&1
tail -4 $O/run_all.txt
for rep in 1 2; do for L in "" _b16; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so
  echo "== lib $L" >> $O/k2_ab.txt
  NUTS_AMD_SELFTEST=0 timeout 300 python tools/quick_k2.py 4096 1024 400 200 2>&1 | grep "M1 =\|sample s" >> $O/k2_ab.txt
done; done
cat $O/k2_ab.txt
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_b16.so
NUTS_AMD_SELFTEST=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "iid or k2 or K2" 2>&1 | tail -3 > $O/b16_parity.txt; cat $O/b16_parity.txt
unset NUTS_AMD_LIB
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
