#!/bin/bash
# round 5: the whole GPU suite on the rebuilt library (every kernel family compiled with the branch-free special functions, the
# orientation-free U-turn sums, merge_math), then the other configs' figures
export TMPDIR=/tmp; O=gpurun_out/r05g; mkdir -p $O
python -c "import hashlib;print('libnuts_amd.so sha256', hashlib.sha256(open('nuts_rs_amd/libnuts_amd.so','rb').read()).hexdigest())" > $O/binary.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1; tail -5 $O/pytest_full.log
timeout 300 python tools/quick_k2.py 4096 1024 100 200 2>&1 | grep "M1 =" > $O/speed.txt
timeout 300 python tools/bench_configs.py k4 --draws 200 2>/dev/null | cut -c1-700 >> $O/speed.txt
timeout 300 python tools/bench_configs.py k4 --draws 200 --chains 65536 2>/dev/null | cut -c1-700 >> $O/speed.txt
timeout 300 python tools/bench_configs.py k3 --draws 100 2>/dev/null | cut -c1-500 >> $O/speed.txt
timeout 300 python tools/bench_configs.py k5 --draws 100 2>/dev/null | cut -c1-500 >> $O/speed.txt
timeout 300 python tools/bench_k5.py > $O/k5.txt 2>&1; tail -2 $O/k5.txt | cut -c1-600
cat $O/speed.txt
timeout 900 python tools/fuzz_parity.py --cases 100 --seed 511 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
timeout 900 python tools/fuzz_parity.py --cases 25 --scale --seed 512 > $O/fuzz_scale.txt 2>&1; tail -1 $O/fuzz_scale.txt
