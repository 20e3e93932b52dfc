#!/bin/bash
# round 3: where does ONE wavefront spend its cycles?  PC sampling of the funnel draw kernel, one chain per CU, deep trees
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03b; mkdir -p $O
cd /tmp
for method in stochastic host_trap; do
  if [ $method = stochastic ]; then unit=cycles; iv=65536; else unit=time; iv=100; fi
  ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $iv \
     --kernel-trace --output-format csv -d $O/pcs_$method -- python $GRAFT_REPO_ROOT/tools/leaf_latency.py --logp funnel --dim 101 --maxdepth 8 --draws 40 --chains 256 > $O/pcs_$method.log 2>&1
  echo "$method rc=$?"; tail -3 $O/pcs_$method.log
  find $O/pcs_$method -type f | head; du -sh $O/pcs_$method
done
# keep the output small: the sampling csv only
find $O -name "*.csv" -size +60M -delete
