#!/bin/bash
# round 5, THE binary of record (estimator unit: overlapped QL, 16 loads in flight in QR / tridiagonalisation / QL application, 4-deep products): GPU suite with the driver's command, smoke both ways, bench line, fuzz, rocprofv3 of the bench command, speeds, low-rank warm-ups
export TMPDIR=/tmp; O=gpurun_out/r05fin6; mkdir -p $O
python -c "import hashlib;print('libnuts_amd.so sha256', hashlib.sha256(open('nuts_rs_amd/libnuts_amd.so','rb').read()).hexdigest())" > $O/binary.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_run1.log 2>&1; tail -1 $O/pytest_run1.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python __graft_entry__.py smoke > $O/smoke_after_build_in_one_process.log 2>&1; tail -1 $O/smoke_after_build_in_one_process.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
timeout 900 python tools/fuzz_parity.py --cases 120 --seed 781 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
timeout 900 python tools/fuzz_parity.py --cases 25 --scale --seed 782 > $O/fuzz_scale.txt 2>&1; tail -1 $O/fuzz_scale.txt
bash tools/pmc_run.sh r05fin6_k2 --other-configs none > $O/pmc_k2.log 2>&1; tail -3 $O/pmc_k2.log
timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" > $O/speed.txt
timeout 300 python tools/bench_configs.py k4 --draws 200 2>/dev/null | cut -c1-700 >> $O/speed.txt
timeout 300 python tools/bench_configs.py k3 --draws 100 2>/dev/null | cut -c1-500 >> $O/speed.txt
timeout 300 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --chains 1 2>/dev/null | cut -c1-330 >> $O/speed.txt
cat $O/speed.txt
timeout 900 python tools/fuzz_parity.py --cases 120 --seed 783 > $O/fuzz2.txt 2>&1; tail -1 $O/fuzz2.txt
timeout 900 python tools/fuzz_parity.py --cases 120 --seed 784 > $O/fuzz3.txt 2>&1; tail -1 $O/fuzz3.txt
for i in 1 2; do timeout 600 python tools/bench_lowrank_adapt.py 2>/dev/null | tail -1 | cut -c1-900 >> $O/lowrank_adapt_dim128.txt; done
timeout 900 python tools/bench_lowrank_adapt.py --dim 384 --chains 256 --tune 150 2>/dev/null | tail -1 | cut -c1-900 > $O/lowrank_adapt_dim384.txt
cat $O/lowrank_adapt_dim128.txt $O/lowrank_adapt_dim384.txt | grep -o "dim [0-9]* \|\"warmup_wall_s\": [0-9.]*\|\"estimator_rounds_s\": {\"total\": [0-9.]*"
