#!/bin/bash
# round 3: whole GPU suite on the build with the re-measured lane thresholds; lane / group at 65536 chains for dim 8 and 16
export TMPDIR=/tmp; O=gpurun_out/r03u; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; tail -14 $O/pytest.log | cut -c1-200
for d in 8 16; do for lc in 1 2; do
  timeout 200 python tools/leaf_latency.py --logp iid --dim $d --maxdepth 6 --draws 20 --chains 65536 --lane-chains $lc >> $O/leaf_crossover.jsonl 2>> $O/err
done; done
python - <<PY
import json
for l in open("$O/leaf_crossover.jsonl"):
    d=json.loads(l); print(d['logp'],d['dim'],d['chains'],'lane_chains',d.get('lane_chains'),'%.3g'%d['leapfrogs_per_s'],'lane',d.get('lane_launches'),'grp',d['group_launches'])
PY
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; head -c 600 $O/bench.json; echo
