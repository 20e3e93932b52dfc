#!/bin/bash
# first GPU pass of round 2: parity suite, the new bench line (draws recorded, live counters), probe + calibration, profiles
export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
tools/hbm_calibrate.sh r02a_cal > $O/cal.log 2>&1; tail -5 $O/cal.log
cp gpurun_out/r02a_cal/hbm_calibration.json profiles/r02a_hbm_calibration.json 2>/dev/null
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | head -c 3000; tail -3 $O/bench.err
python bench.py --no-record --pmc off --no-cpu-baseline > $O/bench_norecord.json 2> $O/bench_norecord.err
tools/pmc_run.sh r02a_pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
(which cargo; which rustc; nproc; lscpu | head -20) > $O/host.txt 2>&1
