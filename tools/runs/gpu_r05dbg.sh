#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r05dbg; mkdir -p $O
for L in "$@"; do
  NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$L.so timeout 300 python tools/probes/repro_rng_log.py 2 2>&1 | grep -v amdgpu.ids >> $O/out2.txt
done
cat $O/out2.txt
