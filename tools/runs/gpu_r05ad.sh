#!/bin/bash
# round 5: K2's kernel ((16,1) only) with merge_math alone inlined (mm) against out of line (m0), same box, three rounds
export TMPDIR=/tmp; O=gpurun_out/r05ad; mkdir -p $O
for L in m0 mm m0 mm m0 mm; do
  NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$L.so timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" | sed "s/^/$L /" >> $O/k2_ab.txt
done
cat $O/k2_ab.txt
NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_mm.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "k2 or dim1024 or iid" 2>&1 | tail -2
