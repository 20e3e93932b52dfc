#!/bin/bash
# round 5: the user-module test that fails on the r05x headers, with the module built three ways
export TMPDIR=/tmp; O=gpurun_out/r05z; mkdir -p $O
for F in "" "-DNM_DETMATH_INLINE=1" "-mllvm -enable-ipra=false"; do
  rm -rf tests/_modules/*.so
  echo "== module flags: '$F'" >> $O/module_variants.txt
  NM_MODULE_EXTRA_FLAGS="$F" timeout 900 python -m pytest tests/test_density_module.py -q -k "low_rank_adaptation_trajectory_kinds" 2>&1 | grep "AssertionError\|passed\|failed" | cut -c1-250 >> $O/module_variants.txt
done
cat $O/module_variants.txt
