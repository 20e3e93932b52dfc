#!/bin/bash
# round 4, FINAL binary (after the early-group-draws change in nuts_engine.hip): GPU suite twice, smoke, bench line, rocprofv3 of the bench command (K2),
# of K4 (65536: lane + early group draws) and K5, fuzz 200 + 25 (scale)
export TMPDIR=/tmp; O=gpurun_out/r04zz; mkdir -p $O
python -c "import hashlib;print('libnuts_amd.so sha256', hashlib.sha256(open('nuts_rs_amd/libnuts_amd.so','rb').read()).hexdigest())" > $O/binary.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_run1.log 2>&1; tail -1 $O/pytest_run1.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python __graft_entry__.py smoke > $O/smoke_after_build_in_one_process.log 2>&1; tail -1 $O/smoke_after_build_in_one_process.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 200 $O/bench.json; echo
timeout 900 python tools/fuzz_parity.py --cases 200 --seed 31 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
timeout 900 python tools/fuzz_parity.py --cases 25 --scale --seed 32 > $O/fuzz_scale.txt 2>&1; tail -1 $O/fuzz_scale.txt
bash tools/pmc_run.sh r04zz_k2 --other-configs none > $O/pmc_k2.log 2>&1; tail -1 $O/pmc_k2.log
KSUB=nuts_lane_draw_kernel bash tools/pmc_cfg.sh r04zz_k4_lane k4 --chains 65536 > $O/pmc_k4.log 2>&1
bash tools/pmc_k5.sh r04zz_k5 --mode shared --tune 100 --draws 100 > $O/pmc_k5.log 2>&1; tail -1 $O/pmc_k5.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_run2.log 2>&1; tail -1 $O/pytest_run2.log
