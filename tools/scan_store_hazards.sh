#!/bin/bash
# usage: tools/scan_store_hazards.sh [jobs]  -- device assembly of every kernel unit, scanned by tools/check_store_hazard.py
cd "$(dirname "$0")/../nuts_rs_amd/csrc" || exit 1
OUT=${TMPDIR:-/tmp}/nm_hazard_scan; mkdir -p $OUT
J=${1:-4}
ls kern_*.hip math_seam.hip nuts_engine.hip | xargs -P $J -I{} sh -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-pass-failed --cuda-device-only -S {} -o $OUT/{}.s 2>/dev/null; python ../../tools/check_store_hazard.py $OUT/{}.s | tail -1"
