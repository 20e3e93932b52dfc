#!/bin/bash
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06q; mkdir -p $O
bash tools/runs/gpu_r06p.sh > /dev/null 2>&1; cat gpurun_out/r06p/sabotage.txt
export NUTS_AMD_SELFTEST=0
NM_LOCK_PROF=1 NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_k5p.so timeout 600 python tools/bench_k5.py --mode shared > $O/k5_lock_prof.json 2>$O/k5.err; python -c "
import json; d=json.loads(open('$O/k5_lock_prof.json').read().strip().split('\n')[-1]); print(d['leapfrogs_per_s'], d['f64_dense_TFLOPs']); print(json.dumps(d['lock_prof'], indent=1))"
