#!/bin/bash
# round 6: shader-clock cycles per phase of the K2 kernel (block 0, whole chip loaded): -DNM_PROF=1 build of the (16,1) iid kernel
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06e; mkdir -p $O
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_prof.so
NUTS_AMD_SELFTEST=0 timeout 300 python tools/prof_phases.py 4096 1024 400 200 > $O/k2_phases.txt 2>&1
cat $O/k2_phases.txt
