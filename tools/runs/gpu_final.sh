#!/bin/bash
# end of round: the whole GPU suite three times on the final binary (logs kept), smoke, the bench line, a fuzz campaign
export TMPDIR=/tmp; O=gpurun_out/${RTAG:-r03x}; mkdir -p $O
python -c "import hashlib;print('libnuts_amd.so sha256', hashlib.sha256(open('nuts_rs_amd/libnuts_amd.so','rb').read()).hexdigest())" > $O/binary.txt
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_run$i.log 2>&1; tail -1 $O/pytest_run$i.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
timeout 900 python tools/fuzz_parity.py --cases ${FUZZ:-120} > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
