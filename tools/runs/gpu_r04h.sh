#!/bin/bash
# round 4, run h: the whole GPU suite on the lockstep build; K5 rocprof passes (kernel trace + MFMA / issue counters)
export TMPDIR=/tmp; O=gpurun_out/r04h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -12 $O/pytest.log
bash tools/pmc_k5.sh r04h_k5 --mode shared --tune 100 --draws 100 2>&1 | tail -3
