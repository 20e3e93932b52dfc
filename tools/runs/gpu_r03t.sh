#!/bin/bash
# round 3: lane / group crossover on K4 (chain counts between 8192 and 65536), then the whole GPU suite on the current build
export TMPDIR=/tmp; O=gpurun_out/r03t; mkdir -p $O
for n in 8192 16384 24576 32768 49152; do for lc in 1 2; do
  timeout 300 python tools/bench_configs.py k4 --chains $n --lane-chains $lc >> $O/k4_crossover.jsonl 2>> $O/err
done; done
for n in 16384 32768; do for lc in 1 2; do
  timeout 200 python tools/leaf_latency.py --logp iid --dim 4 --maxdepth 6 --draws 20 --chains $n --lane-chains $lc >> $O/leaf_crossover.jsonl 2>> $O/err
  timeout 200 python tools/leaf_latency.py --logp iid --dim 16 --maxdepth 6 --draws 20 --chains $n --lane-chains $lc >> $O/leaf_crossover.jsonl 2>> $O/err
done; done
python - <<PY
import json
for l in open("$O/k4_crossover.jsonl"):
    d=json.loads(l); print('K4 chains',d['chains'],'lane_chains',d.get('lane_chains'),'ms',round(d['kernel_ms'],2),'warm',round(d['warmup_kernel_ms'],1),'lf/s %.3g'%d['leapfrogs_per_s'])
for l in open("$O/leaf_crossover.jsonl"):
    d=json.loads(l); print(d['logp'],d['dim'],d['chains'],'lane_chains',d.get('lane_chains'),'%.3g'%d['leapfrogs_per_s'],'lane',d.get('lane_launches'),'grp',d['group_launches'])
PY
timeout 1700 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; tail -14 $O/pytest.log | cut -c1-200
