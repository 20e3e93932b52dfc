#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> <kernel substring> <command ...>  -- rocprofv3 passes over any command: kernel trace, issue counters,
# HBM bytes, L2 hit rate of the LAST dispatch of the kernel (tools/rocprof_summary.py k5)
set -u
TAG=$1; KSUB=$2; shift 2
R=$PWD; export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
"$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- "$@" > $OUT/trace_stdout.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT -o sq -- "$@" > $OUT/sq_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT -o fetch -- "$@" > $OUT/fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT -o write -- "$@" > $OUT/write_stdout.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT -o tcc -- "$@" > $OUT/tcc_stdout.log 2>&1
D=$(dirname $(find $OUT -name trace_results.db | head -1))
python tools/rocprof_summary.py trace $D/trace_results.db $OUT/kernel_trace.txt > /dev/null
python tools/rocprof_summary.py k5 $D $OUT/pmc.json "$KSUB" $OUT | tail -25
find $OUT -name "*_results.db" -delete
