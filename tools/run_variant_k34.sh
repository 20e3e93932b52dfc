#!/bin/bash
# K3 / K4 with the regular build and a tuning build (python tools/build_variant.py <tag> "<flags>" unit.hip[@variant]): tools/run_variant_k34.sh <tag>
for lib in "" $1; do
  if [ -n "$lib" ]; then export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$lib.so; else unset NUTS_AMD_LIB; fi
  echo "== ${lib:-regular}"
  for cfg in "k3" "k4" "k4 --chains 65536"; do
    timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', {k: d[k] for k in ('leapfrogs_per_s','kernel_ms','warmup_kernel_ms','group_launches')})"
  done
done
