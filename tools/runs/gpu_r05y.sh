#!/bin/bash
# round 5: the whole GPU suite without -x on the r05x library (which tests fail besides the user-module one?)
export TMPDIR=/tmp; O=gpurun_out/r05y; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_every_instantiation.py > $O/pytest_full_no_x.log 2>&1; tail -8 $O/pytest_full_no_x.log
