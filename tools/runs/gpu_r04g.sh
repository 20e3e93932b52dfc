#!/bin/bash
timeout 100 python tools/probes/lockstep_sweep.py 2>&1 | grep "^dim" | cut -c1-150
for t in lockprof; do
  echo "== $t"; NM_LOCK_PROF=1 NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$t.so timeout 200 python tools/bench_k5.py --mode shared --tune 50 --draws 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['lock_prof']
print('kernel_ms',round(d['kernel_ms'],1),'lf/s %.3g'%d['leapfrogs_per_s'],'us/round',round(p['us_per_round'],1),{k:round(v*p['us_per_round'],1) for k,v in p['phases'].items()})"
done
timeout 200 python tools/bench_k5.py --mode shared --tune 100 --draws 100 2>/dev/null | cut -c1-400
