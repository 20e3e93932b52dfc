#!/bin/bash
# Tuning build of some translation units with extra flags: tools/build_unit_variant.sh <tag> "<flags>" unit [unit ...]
# -> nuts_rs_amd/libnuts_amd_<tag>.so (the other units come from the regular build; the unit is compiled as ONE translation unit, NM_TU_PART 0,
# and replaces both of its regular objects); run a tool with NUTS_AMD_LIB=<that file>
set -e
cd "$(dirname "$0")/../nuts_rs_amd/csrc"
TAG=$1; EXTRA=$2; shift 2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-pass-failed -mllvm -amdgpu-function-calls=false $EXTRA"
mkdir -p build/$TAG
EXCL=""
for u in "$@"; do /opt/rocm/bin/hipcc $FLAGS -c $u.hip -o build/$TAG/$u.o & EXCL="$EXCL\|build/$u.o\|build/${u}_small.o\|build/${u}_large.o\|build/${u}_inl.o"; done
wait
OBJS=$(ls build/*.o | grep -v "XXXX$EXCL")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/$TAG/*.o -o ../libnuts_amd_$TAG.so
echo built ../libnuts_amd_$TAG.so
