#!/bin/bash
# round 4, run q: fuzz with the adapting low-rank cases (estimator kernel against its twin inside whole runs), whole GPU suite
export TMPDIR=/tmp; O=gpurun_out/r04q; mkdir -p $O
timeout 1200 python tools/fuzz_parity.py --cases 120 --seed 7 > $O/fuzz.txt 2>&1; grep -c "lowrank-adapt" $O/fuzz.txt; grep "lowrank-adapt" $O/fuzz.txt | head -40; tail -1 $O/fuzz.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_run1.log 2>&1; tail -4 $O/pytest_run1.log
