#!/bin/bash
# round 4, run j: the whole GPU suite, smoke, the bench line (K2 + bound model + other configs), fuzz
export TMPDIR=/tmp; O=gpurun_out/r04j; mkdir -p $O
python -c "import hashlib;print('libnuts_amd.so sha256', hashlib.sha256(open('nuts_rs_amd/libnuts_amd.so','rb').read()).hexdigest())" > $O/binary.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_run1.log 2>&1; tail -4 $O/pytest_run1.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04j/bench.json').read().strip().splitlines()[-1])
bm=d['roofline']['bound_model']
print('K2', d['value'], 'frac_moved', d['roofline']['frac_moved'], 'bound frac', bm.get('frac'), 'hbm', bm.get('hbm',{}).get('frac'), 'moved/necessary', bm.get('hbm',{}).get('moved_over_necessary'))
for o in d.get('other_configs', []):
    r=o.get('roofline') or {}
    print(o.get('key'), '%.4g'%o.get('value',0), 'lf/s %.4g'%o.get('leapfrogs_per_s',0), 'ms/step %.4g'%o.get('ms_per_step',0), o.get('error'), 'roof', r.get('bound'), r.get('frac'), r.get('mfma_busy'), 'warmup s', (o.get('warmup') or {}).get('seconds'))
PY
tail -3 $O/bench.err
timeout 600 python tools/fuzz_parity.py --cases 80 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
