#!/bin/bash
# round 4, the binary of record (+ the one-wavefront builds of the 8-lane kernels for every sampler): kinds on K4 shard, GPU suite twice, smoke, bench line, fuzz
export TMPDIR=/tmp; O=gpurun_out/r04fin3; mkdir -p $O
python -c "import hashlib;print('libnuts_amd.so sha256', hashlib.sha256(open('nuts_rs_amd/libnuts_amd.so','rb').read()).hexdigest())" > $O/binary.txt
timeout 600 python tools/bench_kinds.py --k4 --chains 8192 2>&1 | grep case > $O/kinds8192.jsonl; python -c "
import json
for l in open(\"gpurun_out/r04fin3/kinds8192.jsonl\"):
    d = json.loads(l); print(d[\"case\"], \"%.4g\" % d[\"leapfrogs_per_s\"], d.get(\"group_launches\"))
"
for c in 8192; do timeout 300 python tools/bench_configs.py k4 --chains $c 2>/dev/null | grep "^{" >> $O/k4_group.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/r04fin3/k4_group.jsonl"):
    d = json.loads(l); print(d["chains"], "lf/s %.4g" % d["leapfrogs_per_s"], "kernel_ms %.2f" % d["kernel_ms"], "warm_ms %.1f" % d["warmup_kernel_ms"], "lane", d["lane_launches"], "group", d["group_launches"])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_run1.log 2>&1; tail -1 $O/pytest_run1.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 200 $O/bench.json; echo
timeout 900 python tools/fuzz_parity.py --cases 200 --seed 71 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
timeout 900 python tools/fuzz_parity.py --cases 25 --scale --seed 72 > $O/fuzz_scale.txt 2>&1; tail -1 $O/fuzz_scale.txt
KSUB=nuts_group_draw_kernel bash tools/pmc_cfg.sh r04fin3_k4_group k4 --chains 8192 > $O/pmc_k4_group.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_run2.log 2>&1; tail -1 $O/pytest_run2.log
