#!/bin/bash
# round 6: the records of the library in the tree — the driver's commands (pytest -m gpu, smoke, bench.py), rocprofv3 kernel trace + counters of K2
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06fin4; mkdir -p $O
sha256sum nuts_rs_amd/libnuts_amd.so > $O/binary.txt; ls -la nuts_rs_amd/libnuts_amd.so >> $O/binary.txt
( time python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python __graft_entry__.py smoke > $O/build_then_smoke.txt 2>&1; tail -2 $O/build_then_smoke.txt
( time python bench.py ) > $O/bench_line.json 2> $O/bench.err; tail -c 400 $O/bench_line.json; tail -3 $O/bench.err
bash tools/pmc_run.sh r06fin4_pmc --other-configs none > $O/pmc_run.log 2>&1
cp gpurun_out/r06fin4_pmc/kernel_trace.txt $O/k2_kernel_trace.txt 2>/dev/null; cp gpurun_out/r06fin4_pmc/pmc_k2.json $O/k2_pmc.json 2>/dev/null
ls $O
SEEDS="6401" SCALE_SEED=6404 bash tools/runs/gpu_r06r.sh > /dev/null 2>&1; cp gpurun_out/r06r/fuzz.txt $O/fuzz.txt; grep -v "^\[" $O/fuzz.txt
