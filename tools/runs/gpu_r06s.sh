#!/bin/bash
# round 6: K2 on this round's code — the tilings with two wavefronts per SIMD against the record
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06s; mkdir -p $O; export NUTS_AMD_SELFTEST=0
for rep in 1 2; do
  for t in "0 0" "8 2" "4 4" "16 2"; do
    echo "== tiling (doubles per lane, wavefronts per chain) = ($t), 0 0 = automatic = (16, 1)" >> $O/k2_tilings.txt
    timeout 300 python tools/quick_k2.py 4096 1024 400 200 $t 2>&1 | grep "M1 =" >> $O/k2_tilings.txt
  done
done
cat $O/k2_tilings.txt
