#!/bin/bash
# quick accept / reject run for a lane-kernel change: parity (120-case sweep + the other lane tests), K4 and deep-tree throughput
export TMPDIR=/tmp; O=gpurun_out/${RTAG:-lane_quick}; mkdir -p $O; rm -f $O/*.jsonl
timeout 900 python -m pytest tests/test_gpu_lane_chains.py -x -q -m gpu > $O/pytest_lane.log 2>&1; tail -2 $O/pytest_lane.log | cut -c1-200
for i in 1 2; do timeout 300 python tools/bench_configs.py k4 --chains 65536 --lane-chains 2 >> $O/k4.jsonl 2>> $O/err; done
timeout 200 python tools/leaf_latency.py --logp schools --dim 10 --maxdepth 8 --draws 20 --chains 65536 --lane-chains 2 >> $O/leaf.jsonl 2>> $O/err
timeout 200 python tools/leaf_latency.py --logp iid --dim 10 --maxdepth 8 --draws 20 --chains 65536 --lane-chains 2 >> $O/leaf.jsonl 2>> $O/err
timeout 200 python tools/leaf_latency.py --logp iid --dim 4 --maxdepth 6 --draws 20 --chains 65536 --lane-chains 2 >> $O/leaf.jsonl 2>> $O/err
python - <<PY
import json
for l in open("$O/k4.jsonl"):
    d=json.loads(l); print('K4 ms',round(d['kernel_ms'],2),'warm',round(d['warmup_kernel_ms'],1),'lf/s %.3g'%d['leapfrogs_per_s'])
for l in open("$O/leaf.jsonl"):
    d=json.loads(l); print(d['logp'],d['dim'],'%.3g'%d['leapfrogs_per_s'])
PY
