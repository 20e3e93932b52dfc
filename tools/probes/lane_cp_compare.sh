export TMPDIR=/tmp
for v in "" _cp1 _cp2 _cp3; do
  export NUTS_AMD_LIB=nuts_rs_amd/libnuts_amd$v.so
  echo "== lib $v"
  timeout 300 python tools/bench_configs.py k4 --chains 65536 --lane-chains 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K4 ms',round(d['kernel_ms'],2),'warm',round(d['warmup_kernel_ms'],1),'lf/s %.3g'%d['leapfrogs_per_s'])"
  timeout 200 python tools/leaf_latency.py --logp schools --dim 10 --maxdepth 8 --draws 20 --chains 65536 --lane-chains 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deep schools %.3g'%d['leapfrogs_per_s'])"
  timeout 200 python tools/leaf_latency.py --logp iid --dim 4 --maxdepth 6 --draws 20 --chains 65536 --lane-chains 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('iid dim4 depth6 %.3g'%d['leapfrogs_per_s'])"
done
