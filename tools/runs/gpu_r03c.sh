#!/bin/bash
# round 3: differential timing of one chain's leaf (which part of the 2.2 us is what?) + the profiler's PC-sampling configurations
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03c; mkdir -p $O
for tag in "" _xinline _xmerge _xstores _xleap; do
  NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd$tag.so timeout 120 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --draws 20 --chains 1,1024 >> $O/diff.jsonl 2>> $O/diff.err
done
timeout 120 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --draws 20 --chains 1,1024 --no-turn >> $O/diff.jsonl 2>> $O/diff.err
NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_xmerge.so timeout 120 python tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --draws 20 --chains 1,1024 --no-turn >> $O/diff.jsonl 2>> $O/diff.err
cat $O/diff.jsonl
(cd /tmp && timeout 60 rocprofv3 -L > $O/rocprof_list.txt 2>&1); grep -n -i -B2 -A12 "pc.sampl" $O/rocprof_list.txt | head -80
