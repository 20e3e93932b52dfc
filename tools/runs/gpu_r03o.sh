#!/bin/bash
# round 3: lane kernel iteration — parity, phase profile (lprof variant), K4 / deep-tree throughput
export TMPDIR=/tmp; O=gpurun_out/${RTAG:-r03o}; mkdir -p $O
NM_LANE_SWEEP_CASES=${SWEEP:-40} timeout 1500 python -m pytest tests/test_gpu_lane_chains.py -x -q -m gpu > $O/pytest_lane.log 2>&1; tail -3 $O/pytest_lane.log | cut -c1-300
if [ -f nuts_rs_amd/libnuts_amd_lprof.so ]; then
NUTS_AMD_LIB=nuts_rs_amd/libnuts_amd_lprof.so timeout 300 python tools/prof_lane.py > $O/prof_lane_k4.json 2> $O/err
NUTS_AMD_LIB=nuts_rs_amd/libnuts_amd_lprof.so timeout 300 python tools/prof_lane.py --fixed-step 0.01 --draws 20 --maxdepth 8 --logp iid > $O/prof_lane_deep.json 2>> $O/err
fi
timeout 200 python tools/leaf_latency.py --logp schools --dim 10 --maxdepth 8 --draws 20 --chains 65536 --lane-chains 2 >> $O/leaf.jsonl 2>> $O/err
timeout 200 python tools/leaf_latency.py --logp iid --dim 10 --maxdepth 8 --draws 20 --chains 65536 --lane-chains 2 >> $O/leaf.jsonl 2>> $O/err
timeout 300 python tools/bench_configs.py k4 --chains 65536 --lane-chains 2 >> $O/k4.jsonl 2>> $O/err
timeout 300 python tools/bench_configs.py k4 --chains 131072 --lane-chains 2 >> $O/k4.jsonl 2>> $O/err
python - <<PY
import json
O="$O"
for f in ("k4","deep"):
    try: d=json.load(open(f"{O}/prof_lane_{f}.json"))
    except Exception as e: print(f, e); continue
    for tag,v in d.items():
        print(f,tag,round(v["ticks_per_draw"]),"ticks/draw  steps",v["mean_steps_per_draw_block0"],v["max_steps_per_draw_block0"])
        for k,p in v["phases"].items(): print("   %-40s %6.3f  %9.1f ticks/draw  %7.2f marks/draw  -> %.0f ticks/mark"%(k,p["share"],p["ticks_per_draw"],p["marks_per_draw"],p["ticks_per_draw"]/max(p["marks_per_draw"],1e-9)))
for l in open(f"{O}/leaf.jsonl"):
    d=json.loads(l); print(d['logp'],d['dim'],d['chains'],'lane_chains',d.get('lane_chains'),round(d['us_per_leapfrog_of_one_chain'],3),'%.3g'%d['leapfrogs_per_s'],'lane',d.get('lane_launches'),'grp',d['group_launches'])
for l in open(f"{O}/k4.jsonl"):
    d=json.loads(l); print('K4 chains',d['chains'],'lane_chains',d.get('lane_chains'),'ms',round(d['kernel_ms'],2),'warm',round(d['warmup_kernel_ms'],1),'lf/s %.3g'%d['leapfrogs_per_s'])
PY
tail -3 $O/err
