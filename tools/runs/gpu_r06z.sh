#!/bin/bash
# round 6: the bench line of the binary of record with the final bench.py; the GPU unit tests
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06z; mkdir -p $O
sha256sum nuts_rs_amd/libnuts_amd.so bench.py > $O/binary.txt
( time python bench.py ) > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench_line.json; tail -3 $O/bench.err
timeout 900 python -m pytest tests/test_gpu_units.py tests/test_selftest_instantiations.py -q -m gpu 2>&1 | tail -2 > $O/units.txt; cat $O/units.txt
