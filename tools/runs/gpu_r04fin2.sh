#!/bin/bash
# round 4, FINAL binary (+ the one-wavefront-per-SIMD builds of the 8-lane kernels for grids that fit): K4 rates, GPU suite twice, smoke, bench line, fuzz 200 + 25 (scale)
export TMPDIR=/tmp; O=gpurun_out/r04fin2; mkdir -p $O
python -c "import hashlib;print('libnuts_amd.so sha256', hashlib.sha256(open('nuts_rs_amd/libnuts_amd.so','rb').read()).hexdigest())" > $O/binary.txt
for c in 2048 4096 8192 8192 16384; do timeout 300 python tools/bench_configs.py k4 --chains $c 2>/dev/null | grep "^{" >> $O/k4_group.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/r04fin2/k4_group.jsonl"):
    d = json.loads(l); print(d["chains"], "lf/s %.4g" % d["leapfrogs_per_s"], "kernel_ms %.2f" % d["kernel_ms"], "warm_ms %.1f" % d["warmup_kernel_ms"], "lane", d["lane_launches"], "group", d["group_launches"])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_run1.log 2>&1; tail -1 $O/pytest_run1.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 200 $O/bench.json; echo
timeout 900 python tools/fuzz_parity.py --cases 200 --seed 51 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
timeout 900 python tools/fuzz_parity.py --cases 25 --scale --seed 52 > $O/fuzz_scale.txt 2>&1; tail -1 $O/fuzz_scale.txt
KSUB=nuts_group_draw_kernel bash tools/pmc_cfg.sh r04fin2_k4_group k4 --chains 8192 > $O/pmc_k4_group.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_run2.log 2>&1; tail -1 $O/pytest_run2.log
