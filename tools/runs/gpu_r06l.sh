#!/bin/bash
# round 6: is K2 waiting for instructions?  (a) per-wavefront speed against the number of resident wavefronts; (b) the instruction-cache counters this box offers
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06l; mkdir -p $O
cd /tmp; rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|ifetch\|SQ_WAIT\|SQ_WAVE_\|INST_FETCH\|SQC_" | head -60 > $O/counters.txt; cd $GRAFT_REPO_ROOT
cat $O/counters.txt | cut -c1-160
for L in base nog; do
  [ "$L" = base ] && export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd.so || export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$L.so
  for gb in 1024 512 256 128; do
    echo "== lib $L chains $gb (one wavefront each) grid $gb" >> $O/k2_occupancy.txt
    NUTS_AMD_SELFTEST=0 timeout 300 python tools/quick_k2.py $gb 1024 400 200 0 0 $gb 2>&1 | grep "M1 =\|sample s" >> $O/k2_occupancy.txt
  done
done
cat $O/k2_occupancy.txt
