#!/bin/bash
# round 5, the binary of record: rocprofv3 kernel trace + counters (each --pmc group alone) of K3, K4 (8192 / 65536), K5 and the estimator kernel
export TMPDIR=/tmp
bash tools/pmc_cmd.sh r05fin_k3 nuts_draw_kernel python tools/bench_configs.py k3 --draws 100 > gpurun_out/r05fin_k3.log 2>&1
bash tools/pmc_cmd.sh r05fin_k4_8192 nuts_group_draw_kernel python tools/bench_configs.py k4 --draws 200 > gpurun_out/r05fin_k4_8192.log 2>&1
bash tools/pmc_cmd.sh r05fin_k4_65536 nuts_lane_draw_kernel python tools/bench_configs.py k4 --draws 200 --chains 65536 > gpurun_out/r05fin_k4_65536.log 2>&1
bash tools/pmc_cmd.sh r05fin_k5 nuts_lockstep_kernel python tools/bench_k5.py --mode shared > gpurun_out/r05fin_k5.log 2>&1
bash tools/pmc_cmd.sh r05fin_lowrank lr_estimate_kernel python tools/bench_lowrank_adapt.py > gpurun_out/r05fin_lowrank.log 2>&1
for t in k3 k4_8192 k4_65536 k5 lowrank; do echo "== $t"; tail -12 gpurun_out/r05fin_$t.log | cut -c1-200; done
