#!/bin/bash
# round 5: the iid dim 129 LowRankNutsSettings mismatch ((4,1) tiling, fused leapfrog + packed sums): forms of the leapfrog's two fused multiply-adds
export TMPDIR=/tmp; O=gpurun_out/r05r; mkdir -p $O
for L in "$@"; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$L.so
  echo "== lib $L" >> $O/fuzz_only.txt
  timeout 600 python tools/fuzz_parity.py --cases 120 --seed 551 --only 54,109 2>&1 | tail -3 >> $O/fuzz_only.txt
done
cat $O/fuzz_only.txt
