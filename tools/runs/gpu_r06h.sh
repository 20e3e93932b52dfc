#!/bin/bash
# round 6: phase profile of K2 with the U-turn tests' loads taken out (timing only): where does the wait go?
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06h; mkdir -p $O
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_xlp.so
NUTS_AMD_SELFTEST=0 timeout 300 python tools/prof_phases.py 4096 1024 400 200 2>&1 | sed -n '/sampling/,$p' > $O/k2_xl_phases.txt
cat $O/k2_xl_phases.txt
