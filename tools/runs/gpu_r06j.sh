#!/bin/bash
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06j; mkdir -p $O
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_${1:-nogp}.so
NUTS_AMD_SELFTEST=0 timeout 300 python tools/prof_phases.py 4096 1024 400 200 2>&1 | sed -n '/sampling/,$p' > $O/k2_${1:-nogp}_phases.txt
cat $O/k2_${1:-nogp}_phases.txt
