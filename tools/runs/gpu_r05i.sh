#!/bin/bash
# round 5: the library rebuilt through the scanning build (lane units on the rounds-1-4 special functions, group kernels refilling all chains
# together): device-side equivalence of the branch-free functions, the whole GPU suite, the other configs, instruction mixes
export TMPDIR=/tmp; O=gpurun_out/r05i; mkdir -p $O
python -c "import hashlib;print('libnuts_amd.so sha256', hashlib.sha256(open('nuts_rs_amd/libnuts_amd.so','rb').read()).hexdigest())" > $O/binary.txt
timeout 300 tools/probes/sl_math_device_check > $O/sl_math_device_check.txt 2>&1; tail -3 $O/sl_math_device_check.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1; tail -5 $O/pytest_full.log
timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" > $O/speed.txt
timeout 300 python tools/bench_configs.py k4 --draws 200 2>/dev/null | cut -c1-700 >> $O/speed.txt
timeout 300 python tools/bench_configs.py k4 --draws 200 --chains 65536 2>/dev/null | cut -c1-700 >> $O/speed.txt
timeout 300 python tools/bench_configs.py k3 --draws 100 2>/dev/null | cut -c1-500 >> $O/speed.txt
timeout 300 python tools/bench_k5.py --mode shared 2>/dev/null | cut -c1-700 >> $O/speed.txt
timeout 300 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --chains 1 2>/dev/null | cut -c1-330 >> $O/speed.txt
cat $O/speed.txt
timeout 900 python tools/pmc_mix.py $O/mix_k4g.json nuts_group_draw_kernel 8 -- python tools/mix_driver.py k4g 200 > $O/mix_k4g.log 2>&1
timeout 900 python tools/pmc_mix.py $O/mix_k3deep.json nuts_draw_kernel 1 -- python tools/mix_driver.py k3deep 20 > $O/mix_k3deep.log 2>&1
timeout 900 python tools/fuzz_parity.py --cases 100 --seed 521 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
# the estimator launch's block count (scratch working set): libnuts_amd_est.so reads NM_LR_EST_BLOCKS
for G in 1024 512 256 128; do NM_LR_EST_BLOCKS=$G NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_est.so timeout 300 python tools/bench_lowrank_adapt.py 2>/dev/null | cut -c1-600 >> $O/est_blocks.txt; done; cat $O/est_blocks.txt
