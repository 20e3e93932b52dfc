#!/bin/bash
# usage: tools/hbm_calibrate.sh <tag>   -- un-profiled probe rates, then the two --pmc passes, then the factors
set -u
TAG=$1
R=$PWD; export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python tools/hbm_probe.py run > $OUT/probe_rates.json 2> $OUT/probe_rates.err
rocprofv3 --pmc FETCH_SIZE -d $OUT -o fetch -- python tools/hbm_probe.py run --iters 3 > $OUT/fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT -o write -- python tools/hbm_probe.py run --iters 3 > $OUT/write_stdout.log 2>&1
F=$(find $OUT -name fetch_results.db | head -1); D=$(dirname "$F")
python tools/hbm_probe.py calibrate "$D" $OUT/hbm_calibration.json --rates $OUT/probe_rates.json > $OUT/calibrate.log 2>&1
cat $OUT/probe_rates.json; cat $OUT/calibrate.log | tail -30
find $OUT -name "*_results.db" -delete   # the summaries are what travels back (gpurun merges at most 64 MiB)
