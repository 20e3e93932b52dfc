#!/bin/bash
# usage: tools/pmc_k5.sh <tag> [bench_k5 args...]  -- rocprofv3 passes over tools/bench_k5.py: kernel trace, MFMA, issue, HBM
set -u
TAG=$1; shift
R=$PWD; export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python tools/bench_k5.py "$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python tools/bench_k5.py "$@" > $OUT/trace_stdout.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $OUT -o mfma -- python tools/bench_k5.py "$@" > $OUT/mfma_stdout.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT -o sq -- python tools/bench_k5.py "$@" > $OUT/sq_stdout.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT -o lds -- python tools/bench_k5.py "$@" > $OUT/lds_stdout.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT -o fetch -- python tools/bench_k5.py "$@" > $OUT/fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT -o write -- python tools/bench_k5.py "$@" > $OUT/write_stdout.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT -o tcc -- python tools/bench_k5.py "$@" > $OUT/tcc_stdout.log 2>&1
D=$(dirname $(find $OUT -name trace_results.db | head -1))
python tools/rocprof_summary.py trace $D/trace_results.db $OUT/kernel_trace.txt > /dev/null
python tools/rocprof_summary.py k5 $D $OUT/pmc_k5.json "${KSUB:-nuts_lockstep_kernel}" $OUT
cat $OUT/bench.json
find $OUT -name "*_results.db" -delete   # the summaries are what travels back (gpurun merges at most 64 MiB)
