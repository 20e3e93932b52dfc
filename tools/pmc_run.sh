#!/bin/bash
# usage: tools/pmc_run.sh <tag> [bench args...]  -- three rocprofv3 passes: kernel-trace, FETCH_SIZE, WRITE_SIZE (+TCC hit/miss)
set -u
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --no-cpu-baseline --pmc off --repeats 1 "$@" > $OUT/trace_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT -o fetch -- python bench.py --no-cpu-baseline --pmc off --repeats 1 "$@" > $OUT/fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT -o write -- python bench.py --no-cpu-baseline --pmc off --repeats 1 "$@" > $OUT/write_stdout.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT -o tcc -- python bench.py --no-cpu-baseline --pmc off --repeats 1 "$@" > $OUT/tcc_stdout.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT -o sq -- python bench.py --no-cpu-baseline --pmc off --repeats 1 "$@" > $OUT/sq_stdout.log 2>&1
D=$(dirname $(find $OUT -name trace_results.db | head -1))
python tools/rocprof_summary.py trace $D/trace_results.db $OUT/kernel_trace.txt > /dev/null
python tools/rocprof_summary.py pmc $D $OUT/pmc_k2.json nuts_draw_kernel $OUT > /dev/null
ls -la $OUT
find $OUT -name "*_results.db" -delete   # the summaries are what travels back (gpurun merges at most 64 MiB)
