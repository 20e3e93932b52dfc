#!/bin/bash
# round 5: K2 / K3 / K4 units with EVERY special function inlined (no s_swappc_b64 at all: NM_DETMATH_INLINE = 1 for merge_math, sin-cos, expm1 too) against the library
export TMPDIR=/tmp; O=gpurun_out/r05ac; mkdir -p $O
for L in "" _il "" _il; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so
  echo "== lib$L" >> $O/speed.txt
  timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" >> $O/speed.txt
  timeout 300 python tools/bench_configs.py k4 --draws 200 2>/dev/null | cut -c300-520 >> $O/speed.txt
  timeout 300 python tools/bench_configs.py k3 --draws 100 2>/dev/null | cut -c300-500 >> $O/speed.txt
  timeout 300 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --chains 1 2>/dev/null | cut -c150-330 >> $O/speed.txt
done
cat $O/speed.txt
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_il.so
timeout 1500 python -m pytest tests/test_gpu_every_instantiation.py -q -n 4 -k "iid-nuts or funnel-nuts or schools-nuts" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
