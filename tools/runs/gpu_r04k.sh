#!/bin/bash
for t in lockprof lp_noload lp_nolds lp_neither; do
NM_LOCK_PROF=1 NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$t.so timeout 200 python tools/bench_k5.py --mode shared --tune 50 --draws 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['lock_prof']
print('$t','us/round',round(p['us_per_round'],1),{k[:12]:round(v*p['us_per_round'],1) for k,v in p['phases'].items()})"; done
