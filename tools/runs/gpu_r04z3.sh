#!/bin/bash
# round 4, final binary: rocprofv3 kernel trace + HBM / issue counters of the bench command itself (K2) and of K3 / K4 (65536 and 8192 chains)
export TMPDIR=/tmp
bash tools/pmc_run.sh r04z_k2 --other-configs none > gpurun_out/r04z_k2.log 2>&1; tail -3 gpurun_out/r04z_k2.log
KSUB=nuts_draw_kernel bash tools/pmc_cfg.sh r04z_k3 k3 > gpurun_out/r04z_k3.log 2>&1; tail -2 gpurun_out/r04z_k3.log | cut -c1-300
KSUB=nuts_lane_draw_kernel bash tools/pmc_cfg.sh r04z_k4_lane k4 --chains 65536 > gpurun_out/r04z_k4_lane.log 2>&1; tail -2 gpurun_out/r04z_k4_lane.log | cut -c1-300
KSUB=nuts_group_draw_kernel bash tools/pmc_cfg.sh r04z_k4_group k4 --chains 8192 > gpurun_out/r04z_k4_group.log 2>&1; tail -2 gpurun_out/r04z_k4_group.log | cut -c1-300
