#!/bin/bash
# round 4, run n: the estimator kernel — bit for bit against its twin, whole-run parity, the adaptation bench on host and device
export TMPDIR=/tmp; O=gpurun_out/r04n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lowrank.py -q -x -k "twin or device_estimator or estimator_place" > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
for place in host device; do
  timeout 600 python tools/bench_lowrank_adapt.py --place $place > $O/adapt_$place.json 2> $O/adapt_$place.err; tail -2 $O/adapt_$place.err; cat $O/adapt_$place.json
done
timeout 1200 python -m pytest tests/test_gpu_lowrank.py -q > $O/pytest_lowrank.log 2>&1; tail -5 $O/pytest_lowrank.log
