#!/bin/bash
# K5 with the regular build and the tuning builds of the matrix-core kernels (tools/build_tile_variant.sh)
for lib in "" ${1:-tc8o1 tc8o2}; do
  if [ -n "$lib" ]; then export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$lib.so; else unset NUTS_AMD_LIB; fi
  echo "== ${lib:-regular}"
  timeout 300 python tools/bench_configs.py k5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in d if k in ('leapfrogs_per_s','kernel_ms','warmup_s','leapfrogs_per_draw','tile_launches','mean','var')} or d)"
  timeout 300 python tools/bench_k5.py --mode shared 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('leapfrogs_per_s','f64_dense_TFLOPs','matrix_core_launches','whitened_draws')})"
done
