#!/bin/bash
# round 3 (late): U-turn tests two levels at a time in the batched wave kernels — parity, then single-chain latency and K3
export TMPDIR=/tmp; O=gpurun_out/${RTAG:-r04h}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_statistics.py -x -q -m gpu > $O/pytest_core.log 2>&1; tail -3 $O/pytest_core.log | cut -c1-200
for lp in funnel iid; do timeout 200 python tools/leaf_latency.py --logp $lp --dim 101 --maxdepth 8 --draws 20 --chains 1,1024,8192 >> $O/leaf.jsonl 2>> $O/err; done
timeout 200 python tools/leaf_latency.py --logp iid --dim 256 --maxdepth 8 --draws 20 --chains 1,1024 >> $O/leaf.jsonl 2>> $O/err
timeout 400 python tools/bench_configs.py k3 >> $O/k3.jsonl 2>> $O/err
python - <<PY
import json
for l in open("$O/leaf.jsonl"):
    d=json.loads(l); print(d['logp'],d['dim'],d['chains'],round(d['us_per_leapfrog_of_one_chain'],3),'%.3g'%d['leapfrogs_per_s'])
for l in open("$O/k3.jsonl"):
    d=json.loads(l); print('K3 ms',round(d['kernel_ms'],1),'lf/s %.3g'%d['leapfrogs_per_s'])
PY
