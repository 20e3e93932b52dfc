#!/bin/bash
# round 3: batched merges (resolve_chunk) — parity (suite + fuzz), then single-chain latency and K3 / K2 / K4 throughput
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 900 python tools/fuzz_parity.py --cases 150 --seed 31 > $O/fuzz.log 2>&1; tail -4 $O/fuzz.log
for lp in funnel iid; do
  timeout 300 python tools/leaf_latency.py --logp $lp --dim 101 --maxdepth 8 --draws 20 --chains 1,1024,8192 >> $O/leaf_latency.jsonl 2>> $O/leaf.err
done
timeout 120 python tools/leaf_latency.py --logp iid --dim 256 --maxdepth 8 --draws 20 --chains 1,1024 >> $O/leaf_latency.jsonl 2>> $O/leaf.err
timeout 120 python tools/leaf_latency.py --logp schools --dim 10 --maxdepth 8 --draws 20 --chains 1,1024 --lane-groups 1 >> $O/leaf_latency.jsonl 2>> $O/leaf.err
cat $O/leaf_latency.jsonl | cut -c1-330
timeout 600 python tools/bench_configs.py k3 > $O/k3.json 2> $O/k3.err; cut -c1-700 $O/k3.json
timeout 600 python tools/bench_configs.py k4 > $O/k4.json 2> $O/k4.err; cut -c1-400 $O/k4.json
