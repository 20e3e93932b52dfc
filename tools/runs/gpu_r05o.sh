#!/bin/bash
# round 5: bisecting the 4-doubles ExactNormal / Microcanonical parity failure on single-tiling builds of kern_kin_diag_normal
export TMPDIR=/tmp; O=gpurun_out/r05o; mkdir -p $O
for L in "$@"; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$L.so
  echo "== $L" >> $O/out.txt
  timeout 600 python -m pytest tests/test_gpu_trajectory_kinds.py -q -k "dim130_dpl4 and wave" 2>&1 | tail -4 >> $O/out.txt
done
cat $O/out.txt
