#!/bin/bash
# round 4, run a: full GPU suite on the new layout (maxdepth + extra_doublings), the new tests, the bench line with other_configs
export TMPDIR=/tmp; O=gpurun_out/r04a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04a/bench.json').read().strip().splitlines()[-1])
print('K2', d['value'], d['roofline']['frac'])
for o in d.get('other_configs', []):
    print(o.get('key'), o.get('value'), o.get('leapfrogs_per_s'), o.get('ms_per_step'), o.get('error'), json.dumps(o.get('roofline'))[:400], o.get('warmup'))
PY
tail -5 $O/bench.err
