#!/bin/bash
# round 6: K2 — the U-turn tests request their slots' rows NM_TEST_CHUNK iterations ahead (explicit arrays + scheduling barriers); xl2 = timing only, no loads at all
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06u; mkdir -p $O; export NUTS_AMD_SELFTEST=0
for rep in 1 2; do for L in "" _so; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so
  echo "== lib $L" >> $O/k2_chunks.txt
  timeout 300 python tools/quick_k2.py 4096 1024 400 200 2>&1 | grep "M1 =" >> $O/k2_chunks.txt
done; done
cat $O/k2_chunks.txt
