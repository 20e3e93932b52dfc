#!/bin/bash
# round 4, final records (r04z): whole GPU suite twice on the final binary, smoke both ways (import order), the bench line, K5 rocprof, fuzz 200 + 25 (scale)
export TMPDIR=/tmp; O=gpurun_out/r04z; mkdir -p $O
python -c "import hashlib;print('libnuts_amd.so sha256', hashlib.sha256(open('nuts_rs_amd/libnuts_amd.so','rb').read()).hexdigest())" > $O/binary.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_run1.log 2>&1; tail -2 $O/pytest_run1.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 200 $O/bench.json; echo
timeout 900 python tools/fuzz_parity.py --cases 200 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
timeout 900 python tools/fuzz_parity.py --cases 25 --scale > $O/fuzz_scale.txt 2>&1; tail -2 $O/fuzz_scale.txt
bash tools/pmc_k5.sh r04z_k5 --mode shared --tune 100 --draws 100 > $O/pmc_k5.log 2>&1; tail -3 $O/pmc_k5.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_run2.log 2>&1; tail -2 $O/pytest_run2.log
