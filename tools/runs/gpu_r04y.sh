#!/bin/bash
# round 4, run y: the trajectory kinds in the one-chain-per-lane kernels: parity, the 65536-chain K4 job with the kinds
export TMPDIR=/tmp; O=gpurun_out/r04y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_trajectory_kinds.py tests/test_gpu_lane_chains.py -q 2>&1 | tail -8
timeout 600 python tools/bench_kinds.py --k4 --chains 65536 2>&1 | grep case > $O/kinds65536.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r04y/kinds65536.jsonl"):
    d = json.loads(l); print(d["case"], "%.3g" % d["leapfrogs_per_s"], d["group_launches"], d["lane_launches"])
PY
