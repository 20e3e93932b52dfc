#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r04b; mkdir -p $O
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_lowrank.py -q -k "matrix_core_kernel_bit_exact" 2>&1 | tail -4; done > $O/alone.log 2>&1
timeout 600 python -m pytest tests/test_gpu_lowrank.py -q -x 2>&1 | tail -8 > $O/file.log
cat $O/alone.log $O/file.log
