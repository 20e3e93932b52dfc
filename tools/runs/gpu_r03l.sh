#!/bin/bash
# round 3: one chain per lane (nuts_lane.hpp) — parity, then K4 / K1 throughput against the 8-lanes-per-chain kernels
export TMPDIR=/tmp; O=gpurun_out/${RTAG:-r03l}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lane_chains.py -x -q -m gpu > $O/pytest_lane.log 2>&1; tail -12 $O/pytest_lane.log | cut -c1-400
for lc in 1 2; do
  timeout 200 python tools/leaf_latency.py --logp schools --dim 10 --maxdepth 8 --draws 20 --chains 8192,65536 --lane-chains $lc >> $O/leaf.jsonl 2>> $O/err
  timeout 200 python tools/leaf_latency.py --logp iid --dim 10 --maxdepth 8 --draws 20 --chains 65536 --lane-chains $lc >> $O/leaf.jsonl 2>> $O/err
  timeout 300 python tools/bench_configs.py k4 --chains 65536 --lane-chains $lc >> $O/k4.jsonl 2>> $O/err
  timeout 300 python tools/bench_configs.py k4 --chains 8192 --lane-chains $lc >> $O/k4.jsonl 2>> $O/err
done
python - <<'PY'
import json
for l in open('gpurun_out/'+__import__("os").environ.get("RTAG","r03l")+'/leaf.jsonl'):
    d=json.loads(l); print(d['logp'],d['dim'],d['chains'],'lane_chains',d.get('lane_chains'),round(d['us_per_leapfrog_of_one_chain'],3),'%.3g'%d['leapfrogs_per_s'],'lane',d.get('lane_launches'),'grp',d['group_launches'])
for l in open('gpurun_out/'+__import__("os").environ.get("RTAG","r03l")+'/k4.jsonl'):
    d=json.loads(l); print('K4 chains',d['chains'],'lane_chains',d.get('lane_chains'),'ms',round(d['kernel_ms'],2),'warm',round(d['warmup_kernel_ms'],1),'lf/s %.3g'%d['leapfrogs_per_s'],'util',round(d['lockstep_lane_utilisation'],3))
PY
tail -3 $O/err
