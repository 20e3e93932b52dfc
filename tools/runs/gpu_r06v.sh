#!/bin/bash
# round 6: K3 A/B of single-unit builds of kern_funnel.hip@small: one chain's us per leapfrog and the K3 launch
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06v; mkdir -p $O; export NUTS_AMD_SELFTEST=0
for rep in 1 2; do for L in "$@"; do
  [ "$L" = base ] && export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd.so || export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$L.so
  echo "== lib $L" >> $O/k3_ab.txt
  timeout 300 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --chains 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('one chain us/leapfrog', d['us_per_leapfrog_of_one_chain'])" >> $O/k3_ab.txt
  timeout 300 python tools/bench_configs.py k3 --draws 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('K3 leapfrogs/s', d['leapfrogs_per_s'], 'kernel ms', d['kernel_ms'])" >> $O/k3_ab.txt
done; done
cat $O/k3_ab.txt
