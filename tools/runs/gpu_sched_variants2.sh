#!/bin/bash
# K3 / K4 (8192: 8 chains per wavefront; 65536: one chain per lane) on tuning builds of kern_funnel / kern_eight_schools / kern_lane with other
# instruction-scheduling strategies; the lane / group parity tests on each
export TMPDIR=/tmp; O=gpurun_out/sched; mkdir -p $O
for tag in base ${VARIANTS:-memcl ilp}; do
  if [ $tag = base ]; then unset NUTS_AMD_LIB; else export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_$tag.so; [ -f $NUTS_AMD_LIB ] || { echo "$tag: no library"; continue; }; fi
  for cfg in "k3" "k4 --chains 8192" "k4 --chains 65536"; do
    timeout 300 python tools/bench_configs.py $cfg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$tag', '$cfg', 'lf/s %.4g' % d['leapfrogs_per_s'], 'kernel_ms %.2f' % d['kernel_ms'], 'warm_ms %.1f' % d['warmup_kernel_ms'], 'lane', d['lane_launches'], 'group', d['group_launches'])
" | tee -a $O/k34_rates.txt
  done
  if [ $tag != base ]; then timeout 900 python -m pytest tests/test_gpu_lane_chains.py tests/test_gpu_parity.py -q -x 2>&1 | tail -1 | sed "s/^/$tag parity: /" | tee -a $O/k34_rates.txt; fi
done
