#!/bin/bash
# round 5: the library built without device calls (-amdgpu-function-calls=false), device estimator to dim 512:
# (2,1) tiling: every instantiation, the whole GPU suite, smoke, fuzz (the seed that found the fourth incident and two new ones), figures
export TMPDIR=/tmp; O=gpurun_out/r05ab; mkdir -p $O
python -c "import hashlib;print('libnuts_amd.so sha256', hashlib.sha256(open('nuts_rs_amd/libnuts_amd.so','rb').read()).hexdigest())" > $O/binary.txt
timeout 1500 python -m pytest tests/test_gpu_every_instantiation.py -q -n 4 > $O/every_instantiation.log 2>&1; tail -5 $O/every_instantiation.log
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; tail -5 $O/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" > $O/speed.txt
timeout 300 python tools/bench_configs.py k4 --draws 200 2>/dev/null | cut -c1-700 >> $O/speed.txt
timeout 300 python tools/bench_configs.py k4 --draws 200 --chains 65536 2>/dev/null | cut -c1-700 >> $O/speed.txt
timeout 300 python tools/bench_configs.py k3 --draws 100 2>/dev/null | cut -c1-500 >> $O/speed.txt
timeout 300 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --chains 1 2>/dev/null | cut -c1-330 >> $O/speed.txt
cat $O/speed.txt
timeout 900 python tools/fuzz_parity.py --cases 120 --seed 583 > $O/fuzz_583.txt 2>&1; tail -1 $O/fuzz_583.txt
timeout 900 python tools/fuzz_parity.py --cases 120 --seed 581 > $O/fuzz_581.txt 2>&1; tail -1 $O/fuzz_581.txt
timeout 900 python tools/fuzz_parity.py --cases 25 --scale --seed 582 > $O/fuzz_scale.txt 2>&1; tail -1 $O/fuzz_scale.txt
timeout 900 python tools/bench_lowrank_adapt.py --dim 384 --chains 256 --tune 150 > $O/lowrank_adapt_dim384.txt 2>&1; tail -3 $O/lowrank_adapt_dim384.txt | cut -c1-600
timeout 600 python tools/bench_lowrank_adapt.py > $O/lowrank_adapt_dim128.txt 2>&1; tail -2 $O/lowrank_adapt_dim128.txt | cut -c1-600
