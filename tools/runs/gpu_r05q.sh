#!/bin/bash
# round 5: the two fuzz mismatches of r05p (iid dim 129, LowRankNutsSettings, (4,1) tiling) on bisecting builds of kern_lr_iid_normal;
# the LDS end-point cache build (ep) on K3
export TMPDIR=/tmp; O=gpurun_out/r05q; mkdir -p $O
for L in "" _la _lb _lc; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so
  echo "== lib $L" >> $O/fuzz_only.txt
  timeout 600 python tools/fuzz_parity.py --cases 120 --seed 551 --only 54,109 2>&1 | tail -3 >> $O/fuzz_only.txt
done
cat $O/fuzz_only.txt
for L in "" _ep; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$L.so
  echo "== lib $L" >> $O/k3.txt
  timeout 300 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --chains 1 2>/dev/null | cut -c1-330 >> $O/k3.txt
  timeout 300 python tools/bench_configs.py k3 --draws 100 2>/dev/null | cut -c1-500 >> $O/k3.txt
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "funnel" 2>&1 | tail -2 >> $O/k3.txt
done
cat $O/k3.txt
