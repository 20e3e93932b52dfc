#!/bin/bash
# round 5: LrWrap<IidNormal> (4,1) mismatch: bisecting builds (tools/probes/repro_lr_iid129.py prints draw-0 differences)
export TMPDIR=/tmp; O=gpurun_out/r05t; mkdir -p $O
for L in "$@"; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$L.so
  timeout 300 python tools/probes/repro_lr_iid129.py 130 lr 2>&1 | grep "^lib\|draw [01] \|Error\|error" | cut -c1-420 >> $O/out.txt
done
cat $O/out.txt
