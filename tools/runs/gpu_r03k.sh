#!/bin/bash
# round 3: suite on the ABI v13 build (pooled reduction kernel, RCCL tests, reverted MFMA loop), K5 both forms, K2 bench line
export TMPDIR=/tmp; O=gpurun_out/r03k; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; tail -14 $O/pytest.log | cut -c1-200
timeout 300 python tools/bench_k5.py --mode shared > $O/k5_shared.json 2>> $O/err; cut -c150-330 $O/k5_shared.json
timeout 300 python tools/bench_configs.py k5 > $O/k5_diag.json 2>> $O/err; cut -c330-520 $O/k5_diag.json
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo
