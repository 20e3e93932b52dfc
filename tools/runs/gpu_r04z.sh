#!/bin/bash
# round 4, run z (re-entry after the container was re-created): whole GPU suite on the rebuilt binary, smoke, bench line; K4 warm-up per-chunk timing
export TMPDIR=/tmp; O=gpurun_out/r04z; mkdir -p $O
sha256sum nuts_rs_amd/libnuts_amd.so > $O/binary.txt
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu_full_suite.log; tail -3 $O/pytest_gpu_full_suite.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.log; tail -c 600 $O/bench_line.json
timeout 600 python tools/k4_warmup_chunks.py > $O/k4_warmup_chunks.jsonl 2>&1; tail -4 $O/k4_warmup_chunks.jsonl
