#!/bin/bash
# round 5, the binary of record: GPU suite with the driver's command, smoke both ways, the bench line, fuzz, rocprofv3 of the bench command
export TMPDIR=/tmp; O=gpurun_out/r05last; mkdir -p $O
python -c "import hashlib;print('libnuts_amd.so sha256', hashlib.sha256(open('nuts_rs_amd/libnuts_amd.so','rb').read()).hexdigest())" > $O/binary.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_run1.log 2>&1; tail -1 $O/pytest_run1.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python __graft_entry__.py smoke > $O/smoke_after_build_in_one_process.log 2>&1; tail -1 $O/smoke_after_build_in_one_process.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
timeout 900 python tools/fuzz_parity.py --cases 120 --seed 591 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
timeout 900 python tools/fuzz_parity.py --cases 25 --scale --seed 592 > $O/fuzz_scale.txt 2>&1; tail -1 $O/fuzz_scale.txt
bash tools/pmc_run.sh r05last_k2 --other-configs none > $O/pmc_k2.log 2>&1; tail -3 $O/pmc_k2.log
