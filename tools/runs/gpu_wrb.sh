#!/bin/bash
# write_row through a buffer descriptor (NM_WRITE_ROW_BUF): K2 with / without recording on tuning builds of kern_iid_normal, the iid parity cases (odd dims too)
export TMPDIR=/tmp; O=gpurun_out/wrb; mkdir -p $O
for tag in base ${VARIANTS:-wrb wrbnt}; do
  if [ $tag = base ]; then unset NUTS_AMD_LIB; else export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$tag.so; [ -f $NUTS_AMD_LIB ] || { echo "$tag: no library"; continue; }; fi
  timeout 300 python tools/probe_record_cost.py 2>&1 | grep kernel_ms | sed "s/^/$tag: /" | tee -a $O/record_cost.txt
  if [ $tag != base ]; then timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_units.py tests/test_gpu_math_seam.py -q -x -k "not k3 and not k4 and not k5" 2>&1 | tail -1 | sed "s/^/$tag parity: /" | tee -a $O/record_cost.txt; fi
done
