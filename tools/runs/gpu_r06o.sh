#!/bin/bash
# round 6: the first full build with gradient-free points: whole GPU suite, the bench line, K3 phase profile
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06o; mkdir -p $O
( time NUTS_AMD_SELFTEST=0 python -c "import nuts_rs_amd.selftest as s; print(s.run(), 'small runs ok'); print(s.run_all(), 'instantiations ok')" ) > $O/run_all.txt 2>&1; tail -4 $O/run_all.txt
timeout 900 python bench.py > $O/bench.txt 2> $O/bench.err; tail -c 1500 $O/bench.txt
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_k3p.so
PROF_LOGP=funnel NUTS_AMD_SELFTEST=0 timeout 300 python tools/prof_phases.py 8192 101 400 100 2>&1 | sed -n '/sampling/,$p' > $O/k3_phases.txt; cat $O/k3_phases.txt
unset NUTS_AMD_LIB
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
