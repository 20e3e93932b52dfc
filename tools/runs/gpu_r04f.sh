#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r04f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lowrank.py tests/test_gpu_tile_diag.py -q 2>&1 | tail -15
timeout 300 python tools/bench_k5.py --mode shared --tune 100 --draws 100 > $O/k5_lockstep.json 2>$O/k5.err; cat $O/k5_lockstep.json; tail -3 $O/k5.err
