#!/bin/bash
# round 3: rocprofv3 evidence on the final kernels — K2 (the bench line), K3 (batched merges), K4 at 65536 chains (lane kernel, and the
# 8-lane kernels beside it), K5 shared low-rank (matrix cores).  Each: --kernel-trace --stats, then the counter passes on their own.
export TMPDIR=/tmp
timeout 900 bash tools/pmc_run.sh r03v_k2 > gpurun_out/r03v_k2.log 2>&1
KSUB=nuts_draw_kernel timeout 900 bash tools/pmc_cfg.sh r03v_k3 k3 > gpurun_out/r03v_k3.log 2>&1
KSUB=nuts_lane_draw_kernel timeout 900 bash tools/pmc_cfg.sh r03v_k4_lane k4 --chains 65536 > gpurun_out/r03v_k4_lane.log 2>&1
KSUB=nuts_group_draw_kernel timeout 900 bash tools/pmc_cfg.sh r03v_k4_group k4 --chains 65536 --lane-chains 1 > gpurun_out/r03v_k4_group.log 2>&1
timeout 900 bash tools/pmc_k5.sh r03v_k5 --mode shared > gpurun_out/r03v_k5.log 2>&1
ls gpurun_out/r03v_*; du -sh gpurun_out
