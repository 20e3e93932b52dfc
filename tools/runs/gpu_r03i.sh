export TMPDIR=/tmp; O=gpurun_out/r03i; mkdir -p $O
NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_t2.so timeout 300 python -m pytest tests/test_gpu_lowrank.py tests/test_gpu_tile_diag.py -q -m gpu 2>&1 | tail -n 6 | grep -E "passed|failed|FAILED" > $O/tile_t2.log 2>&1
cat $O/tile_t2.log
