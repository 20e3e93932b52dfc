#!/bin/bash
# final GPU pass of round 2: suite, smoke, the bench line, rocprofv3 summaries of the same command, the other configurations
export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; head -c 600 $O/bench.json; echo; tail -2 $O/bench.err
timeout 1200 tools/pmc_run.sh r02g_pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
timeout 900 python tools/bench_configs.py all > $O/other_configs.jsonl 2> $O/other.err; tail -c 400 $O/other_configs.jsonl
timeout 300 python tools/bench_k5.py --mode shared > $O/k5_shared.json 2>> $O/other.err
timeout 300 python tools/bench_to_host.py > $O/to_host.json 2>> $O/other.err; cat $O/to_host.json
