#!/bin/bash
# round 5: packed reductions (wave_sum_packed, swap tails) in the wave kernels: parity + speed against the library
export TMPDIR=/tmp; O=gpurun_out/r05k; mkdir -p $O
for L in libnuts_amd.so libnuts_amd_pk.so; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/$L
  echo "== $L" >> $O/speed.txt
  timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" >> $O/speed.txt
  timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" >> $O/speed.txt
  timeout 300 python tools/leaf_latency.py --logp iid --dim 1024 --maxdepth 6 --step 0.05 --chains 1 2>/dev/null | cut -c1-330 >> $O/speed.txt
  timeout 300 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --chains 1 2>/dev/null | cut -c1-330 >> $O/speed.txt
  timeout 300 python tools/bench_configs.py k3 --draws 100 2>/dev/null | cut -c1-420 >> $O/speed.txt
  timeout 300 python tools/quick_k2.py 8192 256 100 100 2>&1 | grep "M1 =" >> $O/speed.txt
done
cat $O/speed.txt
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_pk.so
timeout 1200 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py -x -q > $O/pytest_pk.log 2>&1; tail -3 $O/pytest_pk.log
timeout 600 python tools/fuzz_parity.py --cases 60 --seed 531 > $O/fuzz_pk.txt 2>&1; tail -1 $O/fuzz_pk.txt
