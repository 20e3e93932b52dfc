#!/bin/bash
# usage: tools/pmc_run.sh <tag> [bench args...]  -- three rocprofv3 passes: kernel-trace, FETCH_SIZE, WRITE_SIZE (+TCC hit/miss)
set -u
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --no-cpu-baseline "$@" > $OUT/trace_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT -o fetch -- python bench.py --no-cpu-baseline "$@" > $OUT/fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT -o write -- python bench.py --no-cpu-baseline "$@" > $OUT/write_stdout.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT -o tcc -- python bench.py --no-cpu-baseline "$@" > $OUT/tcc_stdout.log 2>&1
ls -la $OUT
