#!/bin/bash
# round 3: occupancy x batched-merge variants of the DPL 2 kernels (single-chain latency and loaded throughput), and the LR + host-callback abort
export TMPDIR=/tmp; O=gpurun_out/r03g; mkdir -p $O
for tag in b4 b2 n2; do
  for lp in funnel iid; do
    NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$tag.so timeout 200 python tools/leaf_latency.py --logp $lp --dim 101 --maxdepth 8 --draws 20 --chains 1,1024,8192 >> $O/variants.jsonl 2>> $O/variants.err
  done
  NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$tag.so timeout 300 python tools/bench_configs.py k3 >> $O/k3_variants.jsonl 2>> $O/variants.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r03g/variants.jsonl'):
    d=json.loads(l); print(d['lib'],d['logp'],d['chains'],round(d['us_per_leapfrog_of_one_chain'],3),'%.3g'%d['leapfrogs_per_s'])
for l in open('gpurun_out/r03g/k3_variants.jsonl'):
    d=json.loads(l); print('K3',round(d['kernel_ms'],1),'warm',round(d['warmup_kernel_ms'],1),'%.3g'%d['leapfrogs_per_s'])
PY
for tag in nobatch_lrcb lrcb_lds; do
NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$tag.so timeout 300 python -m pytest tests/test_gpu_host_callback.py -q -m gpu -k "low_rank" > $O/hostcb_lr_$tag.log 2>&1; echo $tag; tail -3 $O/hostcb_lr_$tag.log | cut -c1-200
done
