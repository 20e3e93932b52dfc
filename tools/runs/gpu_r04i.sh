#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r04i; mkdir -p $O
NM_LANE_SWEEP_CASES=40 timeout 600 python -m pytest tests/test_gpu_lane_chains.py -q -x -k "sweep" 2>&1 | tail -6
timeout 300 python -m pytest tests/test_gpu_distributed.py -q -k "exchange_and_finish or partials" 2>&1 | tail -3
for lc in 2 3; do timeout 200 python tools/bench_configs.py k4 --chains 65536 --lane-chains $lc --draws 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lane_chains', d['lane_chains'], 'lf/s %.3g'%d['leapfrogs_per_s'], 'kernel_ms', round(d['kernel_ms'],1), 'warmup_kernel_ms', round(d['warmup_kernel_ms'],1), 'lane launches', d['lane_launches'])"; done
