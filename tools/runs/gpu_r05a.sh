#!/bin/bash
# round 5, first call: baseline figures of record on this box + a PC-sampling attempt on K2 (where do the wavefront's cycles go, by instruction)
export TMPDIR=/tmp; O=gpurun_out/r05a; mkdir -p $O
timeout 300 python tools/quick_k2.py 4096 1024 100 200 > $O/k2_quick.txt 2>&1; tail -3 $O/k2_quick.txt
timeout 300 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --chains 1,8192 > $O/k3_leaf_latency.jsonl 2>&1; tail -2 $O/k3_leaf_latency.jsonl
timeout 300 python tools/leaf_latency.py --logp iid --dim 1024 --maxdepth 6 --step 0.05 --chains 1,1024,4096 > $O/k2_leaf_latency.jsonl 2>&1; tail -3 $O/k2_leaf_latency.jsonl
cd /tmp
for M in stochastic host_trap; do
  timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $( [ $M = stochastic ] && echo cycles || echo time ) --pc-sampling-interval $( [ $M = stochastic ] && echo 1048576 || echo 1000 ) --output-format csv -d /tmp/pcs_$M -- python $GRAFT_REPO_ROOT/tools/quick_k2.py 4096 1024 20 200 > $GRAFT_REPO_ROOT/$O/pcs_$M.log 2>&1
  echo "pcs $M rc $?"; tail -5 $GRAFT_REPO_ROOT/$O/pcs_$M.log
  python $GRAFT_REPO_ROOT/tools/pcsample_agg.py /tmp/pcs_$M $GRAFT_REPO_ROOT/$O/pcs_${M}_agg.json 2>&1 | tail -3
done
