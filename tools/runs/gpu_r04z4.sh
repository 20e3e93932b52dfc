#!/bin/bash
# round 4, final binary: more randomised parity (other seeds) — fuzz 3 x 150 cases, lane sweep with 300 cases
export TMPDIR=/tmp; O=gpurun_out/r04z; mkdir -p $O
for s in 11 12 13; do timeout 900 python tools/fuzz_parity.py --cases 150 --seed $s > $O/fuzz_seed$s.txt 2>&1; tail -1 $O/fuzz_seed$s.txt; done
timeout 900 python tools/fuzz_parity.py --cases 25 --seed 14 --scale > $O/fuzz_scale_seed14.txt 2>&1; tail -1 $O/fuzz_scale_seed14.txt
NM_LANE_SWEEP_CASES=300 timeout 1500 python -m pytest tests/test_gpu_lane_chains.py -q -k sweep > $O/lane_sweep_300.log 2>&1; tail -2 $O/lane_sweep_300.log
