#!/bin/bash
# round 3: the matrix-core kernels with the batched merges (NM_BATCH_IN_TILES=1): parity, K5 shared / K5 diag throughput, phase profile
export TMPDIR=/tmp; O=gpurun_out/r03j; mkdir -p $O
NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_tb.so timeout 300 python -m pytest tests/test_gpu_lowrank.py tests/test_gpu_tile_diag.py -q -m gpu 2>&1 | tail -n 6 | grep -E "passed|failed|FAILED" | tee $O/tile_tb.log
NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_tb.so timeout 300 python tools/bench_k5.py --mode shared > $O/k5_shared_tb.json 2>> $O/err; cut -c150-420 $O/k5_shared_tb.json
NM_TILE_PROF=1 NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_tbprof.so timeout 300 python tools/bench_k5.py --mode shared > $O/k5_shared_tbprof.json 2>> $O/err; python -c "
import json; d=json.load(open('$O/k5_shared_tbprof.json')); print(d['tile_prof'])"
NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_tb.so timeout 300 python tools/bench_configs.py k5 > $O/k5_diag_tb.json 2>> $O/err; cut -c150-520 $O/k5_diag_tb.json
timeout 300 python tools/bench_configs.py k5 > $O/k5_diag.json 2>> $O/err; cut -c150-520 $O/k5_diag.json
