#!/bin/bash
# round 6: more fuzz on the binary of record (the library in the tree)
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06x; mkdir -p $O
sha256sum nuts_rs_amd/libnuts_amd.so | cut -c1-16 > $O/fuzz.txt
for seed in 6501 6502 6503; do timeout 1500 python tools/fuzz_parity.py --cases 360 --seed $seed 2>&1 | tail -1 >> $O/fuzz.txt; done
for seed in 6511 6512; do timeout 900 python tools/fuzz_parity.py --cases 120 --seed $seed --scale 2>&1 | tail -1 >> $O/fuzz.txt; done
cat $O/fuzz.txt
