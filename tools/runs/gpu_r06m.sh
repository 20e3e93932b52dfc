#!/bin/bash
# round 6: instruction-cache counters of the K2 launch (counters only, one pass each)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r06m; mkdir -p $O
export NUTS_AMD_SELFTEST=0
cd /tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQ_INSTS_SMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set -d $O/$tag -o p --output-format csv -- python $R/tools/quick_k2.py 4096 1024 400 200 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" >> $O/icache.txt <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "nuts_draw_kernel" in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k, v in tot.items(): print(k, v, "dispatches", n[k])
PY
  rm -rf $O/$tag
done
cat $O/icache.txt
