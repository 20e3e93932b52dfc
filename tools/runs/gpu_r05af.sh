#!/bin/bash
# round 5: the estimator kernel with 16 (instead of 8) elements requested per thread before the first is used in the QR and tridiagonalisation loops: parity and time
export TMPDIR=/tmp; O=gpurun_out/r05af; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_lowrank.py tests/test_gpu_mclmc.py -q > $O/pytest_lowrank.log 2>&1; tail -3 $O/pytest_lowrank.log
timeout 900 python -m pytest tests/test_gpu_every_instantiation.py -q -n 4 -k "lr_adapt or lr_mclmc" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_density_module.py -q 2>&1 | tail -2
for i in 1 2; do timeout 600 python tools/bench_lowrank_adapt.py 2>/dev/null | tail -1 | cut -c1-900 >> $O/lowrank_adapt_dim128.txt; done
timeout 900 python tools/bench_lowrank_adapt.py --dim 384 --chains 256 --tune 150 2>/dev/null | tail -1 | cut -c1-900 > $O/lowrank_adapt_dim384.txt
cat $O/lowrank_adapt_dim128.txt $O/lowrank_adapt_dim384.txt
timeout 900 python tools/fuzz_parity.py --cases 120 --seed 751 2>&1 | tail -1
