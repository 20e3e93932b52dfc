#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r05w; mkdir -p $O
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_y1.so
timeout 300 python tools/probes/repro_lr_iid129.py 130 lr 0 2 nojit 2>&1 | grep "^lib\|draw [0123] \|rror" | cut -c1-900 >> $O/out3.txt
cat $O/out3.txt
