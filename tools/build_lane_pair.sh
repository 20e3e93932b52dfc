#!/bin/bash
# Rebuild the library after a nuts_lane.hpp edit and, beside it, the -DNM_LANE_PROF=1 variant (libnuts_amd_lprof.so) for tools/prof_lane.py
cd "$(dirname "$0")/.."
(python -m nuts_rs_amd.build > /tmp/b1.log 2>&1 &)
sleep 1
cd nuts_rs_amd/csrc && mkdir -p build/lprof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-pass-failed -DNM_LANE_PROF=1 ${LPROF_EXTRA} -c kern_lane.hip -o build/lprof/kern_lane.o 2>&1 | tail -3
while ! grep -q libnuts_amd.so /tmp/b1.log; do sleep 5; if grep -qi "error" /tmp/b1.log; then cat /tmp/b1.log | tail -20; exit 1; fi; done
OBJS=$(ls build/*.o | grep -v "build/kern_lane.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/lprof/kern_lane.o -o ../libnuts_amd_lprof.so
ls -la ../*.so
