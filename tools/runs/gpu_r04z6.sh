#!/bin/bash
# the warm-up's first draws on the 8-lane kernels inside a lane engine (NM_LANE_EARLY_GROUP_DRAWS): parity across the switch, K4 65536 rates
export TMPDIR=/tmp; O=gpurun_out/r04z6; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_lane_chains.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
for i in 1 2; do timeout 300 python tools/bench_configs.py k4 --chains 65536 2>/dev/null | grep "^{" >> $O/k4_65536.jsonl; done
timeout 300 python tools/bench_configs.py k4 --chains 65536 --lane-chains 2 2>/dev/null | grep "^{" >> $O/k4_65536.jsonl
timeout 300 python tools/bench_configs.py k4 --chains 32768 2>/dev/null | grep "^{" >> $O/k4_65536.jsonl
timeout 300 python tools/bench_configs.py k4 --chains 32768 --lane-chains 2 2>/dev/null | grep "^{" >> $O/k4_65536.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r04z6/k4_65536.jsonl"):
    d = json.loads(l); print(d["chains"], "lane_chains", d["lane_chains"], "lf/s %.4g" % d["leapfrogs_per_s"], "kernel_ms %.2f" % d["kernel_ms"], "warm_ms %.1f" % d["warmup_kernel_ms"], "lane", d["lane_launches"], "group", d["group_launches"])
PY
