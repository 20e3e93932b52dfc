#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r04d; mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_lowrank.py -q -x -k "matrix_core_kernel_bit_exact and tile" 2>&1 | tail -5
timeout 180 python -m pytest tests/test_gpu_lowrank.py -q -k "matrix_core_kernel_bit_exact and lockstep" 2>&1 | tail -30
