#!/bin/bash
# Tuning build of the matrix-core kernels: tools/build_tile_variant.sh <chains per block: 16|8> <blocks per CU> -> nuts_rs_amd/libnuts_amd_tc<chains>o<occ>.so
# (run a tool with NUTS_AMD_LIB=<that file>; the other units are taken from the regular build)
set -e
cd "$(dirname "$0")/../nuts_rs_amd/csrc"
TCN=$1; OCC=$2; TAG=tc${TCN}o${OCC}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-pass-failed -DNM_TILE_CHAINS=$TCN -DNM_TILE_OCC=$OCC"
mkdir -p build/$TAG
for u in kern_tile_mvn_diag kern_tile_mvn_prec nuts_engine; do /opt/rocm/bin/hipcc $FLAGS -c $u.hip -o build/$TAG/$u.o & done
wait
OBJS=$(ls build/*.o | grep -v "kern_tile_mvn_diag.o\|kern_tile_mvn_prec.o\|nuts_engine.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/$TAG/*.o -o ../libnuts_amd_$TAG.so
echo built ../libnuts_amd_$TAG.so
