#!/bin/bash
# round 5: bisecting the ExactNormal (4 doubles per lane) parity failure: library / without packed sums (kA) / with the rounds-1-4 special functions (kB)
export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O
for L in libnuts_amd.so libnuts_amd_kA.so libnuts_amd_kB.so; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/$L
  echo "== $L" >> $O/out.txt
  timeout 900 python -m pytest tests/test_gpu_trajectory_kinds.py tests/test_gpu_mclmc.py -q 2>&1 | tail -6 >> $O/out.txt
done
cat $O/out.txt
unset NUTS_AMD_LIB
timeout 2400 python -m pytest tests -m gpu -q --deselect "tests/test_gpu_trajectory_kinds.py::test_trajectory_kind_parity_bit_exact[exact_diag_dim130_dpl4-wave]" > $O/pytest_rest.log 2>&1; tail -8 $O/pytest_rest.log
