#!/bin/bash
# round 5: variant v1 (orientation-free U-turn sums, merge_math routine, full-tile density path) against the round-4 binary: parity + speed
export TMPDIR=/tmp; O=gpurun_out/r05d; mkdir -p $O
for L in libnuts_amd.so libnuts_amd_v1.so; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/$L
  echo "== $L" >> $O/speed.txt
  timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" >> $O/speed.txt
  timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" >> $O/speed.txt
  timeout 300 python tools/leaf_latency.py --logp iid --dim 1024 --maxdepth 6 --step 0.05 --chains 1,4096 >> $O/speed.txt 2>&1
  timeout 300 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --chains 1,8192 >> $O/speed.txt 2>&1
  timeout 300 python tools/bench_configs.py k3 --draws 100 >> $O/speed.txt 2>&1
done
cat $O/speed.txt | cut -c1-400
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_v1.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -x -q > $O/pytest_v1.log 2>&1; tail -3 $O/pytest_v1.log
timeout 600 python tools/fuzz_parity.py --cases 60 --seed 501 > $O/fuzz_v1.txt 2>&1; tail -2 $O/fuzz_v1.txt
