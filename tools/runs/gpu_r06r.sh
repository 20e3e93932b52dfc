#!/bin/bash
# round 6: fuzz campaign on the library in the tree (gradient-free points on the (16,1) tiling): random densities / dims / tilings / settings against the oracle
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06r; mkdir -p $O
sha256sum nuts_rs_amd/libnuts_amd.so | cut -c1-16 > $O/fuzz.txt
for seed in ${SEEDS:-6101 6102 6103}; do
  timeout 1500 python tools/fuzz_parity.py --cases 360 --seed $seed 2>&1 | tail -4 >> $O/fuzz.txt
done
timeout 900 python tools/fuzz_parity.py --cases 120 --seed ${SCALE_SEED:-6104} --scale 2>&1 | tail -4 >> $O/fuzz.txt
cat $O/fuzz.txt
