#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r05s; mkdir -p $O
for A in "129 lr" "129 diag 4" "130 lr" "200 lr" "256 lr" "129 diag 2"; do
  timeout 300 python tools/probes/repro_lr_iid129.py $A >> $O/out.txt 2>&1
done
cat $O/out.txt
