#!/bin/bash
# round 5: the lane kernels with the branch-free special functions (library) against the same units with the rounds-1-4 forms (lk_old):
# the failing microcanonical lane case, K4 on 65536 chains, then the rest of the suite from the failing test on
export TMPDIR=/tmp; O=gpurun_out/r05h; mkdir -p $O
for L in libnuts_amd.so libnuts_amd_lk_old.so; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/$L
  echo "== $L" >> $O/out.txt
  timeout 600 python -m pytest tests/test_gpu_trajectory_kinds.py -q -k "lane" 2>&1 | tail -4 >> $O/out.txt
  timeout 300 python tools/bench_configs.py k4 --draws 200 --chains 65536 2>/dev/null | cut -c1-900 >> $O/out.txt
done
cat $O/out.txt
unset NUTS_AMD_LIB
timeout 2400 python -m pytest tests -m gpu -q --deselect "tests/test_gpu_trajectory_kinds.py::test_trajectory_kind_parity_bit_exact[micro_funnel_dim11-lane]" > $O/pytest_rest.log 2>&1; tail -8 $O/pytest_rest.log
