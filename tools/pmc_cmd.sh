#!/bin/bash
# usage: KSUB=<kernel substring> CMD="python tools/<bench>.py args" tools/pmc_cmd.sh <tag>
#   rocprofv3 passes over any bench command: kernel trace, issue counters, HBM bytes, L2
set -u
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
$CMD > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $CMD > $OUT/trace_stdout.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT -o sq -- $CMD > $OUT/sq_stdout.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT -o lds -- $CMD > $OUT/lds_stdout.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA -d $OUT -o mfma -- $CMD > $OUT/mfma_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT -o fetch -- $CMD > $OUT/fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT -o write -- $CMD > $OUT/write_stdout.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT -o tcc -- $CMD > $OUT/tcc_stdout.log 2>&1
D=$(dirname $(find $OUT -name trace_results.db | head -1))
python tools/rocprof_summary.py trace $D/trace_results.db $OUT/kernel_trace.txt > /dev/null
python tools/rocprof_summary.py k5 $D $OUT/pmc.json "${KSUB:-nuts_draw_kernel}" $OUT > /dev/null
cat $OUT/bench.json; cat $OUT/pmc.json | head -60
find $OUT -name "*_results.db" -delete   # the summaries are what travels back (gpurun merges at most 64 MiB)
