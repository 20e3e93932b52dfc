#!/bin/bash
# round 5: dynamic instruction mix of the BASELINE kernels (one chain alone = the issue-bound latency; the full grid = what the bench times)
export TMPDIR=/tmp; O=gpurun_out/r05c; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1; grep -c . $O/avail.txt
for K in k2one k2; do timeout 900 python tools/pmc_mix.py $O/mix_$K.json nuts_draw_kernel 1 -- python tools/mix_driver.py $K 100 > $O/mix_$K.log 2>&1; tail -4 $O/mix_$K.log | head -3; done
for K in k3one; do timeout 900 python tools/pmc_mix.py $O/mix_$K.json nuts_draw_kernel 1 -- python tools/mix_driver.py $K 100 > $O/mix_$K.log 2>&1; done
for K in k4g; do timeout 900 python tools/pmc_mix.py $O/mix_$K.json nuts_group_draw_kernel 8 -- python tools/mix_driver.py $K 200 > $O/mix_$K.log 2>&1; done
ls -la $O
