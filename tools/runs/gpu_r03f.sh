export TMPDIR=/tmp; O=gpurun_out/r03f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_lowrank.py -x -q -m gpu > $O/lowrank.log 2>&1; tail -5 $O/lowrank.log | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_host_callback.py -q -m gpu -k "not low_rank" > $O/hostcb_other.log 2>&1; tail -3 $O/hostcb_other.log | cut -c1-200
NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_nobatch_lrcb.so timeout 300 python -m pytest tests/test_gpu_host_callback.py -q -m gpu -k "low_rank" > $O/hostcb_lr_nobatch.log 2>&1; tail -3 $O/hostcb_lr_nobatch.log | cut -c1-200
