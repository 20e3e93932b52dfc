#!/bin/bash
# round 3: batched merges with ONE inlined resolve site, DPL 2 kernels at 2 waves / SIMD — parity (suite + fuzz), latency, K3 / K4 / K1, bench line
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log | cut -c1-200
timeout 900 python tools/fuzz_parity.py --cases 200 --seed 32 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log | cut -c1-200
timeout 900 python tools/fuzz_parity.py --cases 40 --seed 33 --scale > $O/fuzz_scale.log 2>&1; tail -2 $O/fuzz_scale.log | cut -c1-200
for lp in funnel iid; do
  timeout 300 python tools/leaf_latency.py --logp $lp --dim 101 --maxdepth 8 --draws 20 --chains 1,1024,8192 >> $O/leaf_latency.jsonl 2>> $O/leaf.err
done
timeout 120 python tools/leaf_latency.py --logp iid --dim 256 --maxdepth 8 --draws 20 --chains 1,1024 >> $O/leaf_latency.jsonl 2>> $O/leaf.err
timeout 120 python tools/leaf_latency.py --logp iid --dim 1024 --maxdepth 8 --draws 10 --chains 1,1024 >> $O/leaf_latency.jsonl 2>> $O/leaf.err
timeout 120 python tools/leaf_latency.py --logp schools --dim 10 --maxdepth 8 --draws 20 --chains 1,1024 --lane-groups 1 >> $O/leaf_latency.jsonl 2>> $O/leaf.err
python - <<'PY'
import json
for l in open('gpurun_out/r03h/leaf_latency.jsonl'):
    d=json.loads(l); print(d['logp'],d['dim'],d['chains'],round(d['us_per_leapfrog_of_one_chain'],3),'%.3g'%d['leapfrogs_per_s'])
PY
timeout 900 python tools/bench_configs.py all > $O/other_configs.jsonl 2> $O/other.err
python - <<'PY'
import json
for l in open('gpurun_out/r03h/other_configs.jsonl'):
    d=json.loads(l); print(d['config'][:40],'chains',d['chains'],'ms',round(d['kernel_ms'],1),'warm',round(d['warmup_kernel_ms'],1),'lf/s %.3g'%d['leapfrogs_per_s'])
PY
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; head -c 500 $O/bench.json; echo; tail -2 $O/bench.err
for mode in shared; do
  NM_TILE_PROF=1 NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_tileprof.so timeout 300 python tools/bench_k5.py --mode $mode > $O/k5_tileprof_$mode.json 2>> $O/other.err; cut -c1-1500 $O/k5_tileprof_$mode.json
done
timeout 300 python tools/bench_k5.py --mode shared > $O/k5_shared.json 2>> $O/other.err; cut -c1-600 $O/k5_shared.json
