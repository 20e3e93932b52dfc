#!/bin/bash
# round 6: the causal experiment for DESIGN §22's root cause.  The library of 8e9c172 as it was built then (nuts_draw_kernel<4,1,LrWrap<IidNormal>> has
# VGPR spill stores above the exec restore of the `if (!reuse_edge)` join block: tools/check_exec_spill.py) against the SAME library with that one
# s_or_b64 moved up 35 instructions in the assembly (llvm-objdump diff of the two objects: one instruction moved, nothing else).
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06a; mkdir -p $O
cd tools/probes/r06_wt8e9
for L in failing repaired; do
  export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$L.so
  for args in "129 lr 0 10" "130 lr 0 2 nojit" "200 lr 0 10"; do
    echo "== lib $L args $args" >> $O/repro.txt
    timeout 300 python tools/probes/repro_lr_iid129.py $args 2>&1 | grep "^lib\|^draw\|rror" | cut -c1-230 >> $O/repro.txt
  done
done
cat $O/repro.txt
cd $GRAFT_REPO_ROOT; unset NUTS_AMD_LIB
NUTS_AMD_SELFTEST=0 timeout 600 python bench.py > $O/bench_head.txt 2>$O/bench_head.err; tail -c 600 $O/bench_head.txt
