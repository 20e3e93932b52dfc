#!/bin/bash
# K2 on tuning builds of kern_iid_normal with other instruction-scheduling strategies (python tools/build_variant.py <tag> "-mllvm -amdgpu-sched-strategy=..."):
# rate (tools/quick_k2.py, twice) and the iid parity cases on each
export TMPDIR=/tmp; O=gpurun_out/sched; mkdir -p $O
for tag in base ${VARIANTS:-ilp memcl itilp}; do
  if [ $tag = base ]; then unset NUTS_AMD_LIB; else export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$tag.so; [ -f $NUTS_AMD_LIB ] || { echo "$tag: no library"; continue; }; fi
  for i in 1 2; do timeout 300 python tools/quick_k2.py 2>&1 | grep "^M1" | sed "s/^/$tag run $i: /"; done | tee -a $O/k2_rates.txt
  if [ $tag != base ]; then timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py -q -x -k "not k3 and not k4 and not k5" 2>&1 | tail -2 | sed "s/^/$tag parity: /" | tee -a $O/k2_rates.txt; fi
done
