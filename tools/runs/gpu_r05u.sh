#!/bin/bash
# round 5: every instantiation of the one-chain kernels against the oracle on the library of record (8e9c172); K2 with the leapfrog's fused
# multiply-adds as inline asm (the library) against __builtin_fma (f2)
export TMPDIR=/tmp; O=gpurun_out/r05u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_every_instantiation.py -q -n 4 2>&1 | tail -40 > $O/every_instantiation_head_8e9c172.txt
tail -15 $O/every_instantiation_head_8e9c172.txt
for L in "" _f2 "" _f2; do
  NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" | sed "s/^/lib$L /" >> $O/k2_ab.txt
done
cat $O/k2_ab.txt
