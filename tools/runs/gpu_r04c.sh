#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r04c; mkdir -p $O
python tools/probes/tile_dpl2_probe.py > $O/probe.log 2>&1; tail -12 $O/probe.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
