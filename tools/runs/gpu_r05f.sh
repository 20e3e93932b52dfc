#!/bin/bash
# round 5: variant v7 (v3 + no s_load per leaf, the main tree's end points (z, v) in registers) against v3
export TMPDIR=/tmp; O=gpurun_out/r05f; mkdir -p $O
for L in libnuts_amd_v3.so libnuts_amd_v7.so; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/$L
  echo "== $L" >> $O/speed.txt
  timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" >> $O/speed.txt
  timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" >> $O/speed.txt
  timeout 300 python tools/leaf_latency.py --logp iid --dim 1024 --maxdepth 6 --step 0.05 --chains 1 2>/dev/null | cut -c1-330 >> $O/speed.txt
  timeout 300 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --chains 1 2>/dev/null | cut -c1-330 >> $O/speed.txt
  timeout 300 python tools/quick_k2.py 4096 1000 100 100 2>&1 | grep "M1 =" >> $O/speed.txt
  timeout 300 python tools/quick_k2.py 8192 512 100 100 2>&1 | grep "M1 =" >> $O/speed.txt
  timeout 300 python tools/quick_k2.py 8192 256 100 100 2>&1 | grep "M1 =" >> $O/speed.txt
  timeout 300 python tools/bench_configs.py k3 --draws 100 2>/dev/null | cut -c1-420 >> $O/speed.txt
done
cat $O/speed.txt
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_v7.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -x -q > $O/pytest_v7.log 2>&1; tail -3 $O/pytest_v7.log
timeout 600 python tools/fuzz_parity.py --cases 60 --seed 503 > $O/fuzz_v7.txt 2>&1; tail -1 $O/fuzz_v7.txt
timeout 900 python tools/pmc_mix.py $O/mix_k2_v7.json nuts_draw_kernel 1 -- python tools/mix_driver.py k2 100 > $O/mix_k2_v7.log 2>&1; tail -3 $O/mix_k2_v7.log
