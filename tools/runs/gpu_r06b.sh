#!/bin/bash
# round 6: the per-instantiation known answers on the library of record (run_all timing, first-use hook), then the whole GPU suite with the hook on
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06b; mkdir -p $O
( time NUTS_AMD_SELFTEST=0 python -c "import nuts_rs_amd.selftest as s; print(s.run(), 'small runs ok'); print(s.run_all(), 'instantiations ok')" ) > $O/run_all.txt 2>&1
tail -5 $O/run_all.txt
( time python -c "
import time, nuts_rs_amd as N, numpy as np
t=time.time(); b=N.ChainBatch(N.DiagNutsSettings(num_chains=64, seed=1), N.LogpSpec.iid_normal(1000, 3.0), 64); print('first engine of (iid, nuts, 16x1):', round(time.time()-t,3), 's'); b.close()
t=time.time(); b=N.ChainBatch(N.DiagNutsSettings(num_chains=64, seed=1), N.LogpSpec.iid_normal(1000, 3.0), 64); print('second:', round(time.time()-t,3), 's'); b.close()
t=time.time(); b=N.ChainBatch(N.LowRankNutsSettings(num_chains=64, seed=1), N.LogpSpec.funnel(101), 64); print('first engine of (funnel, lr, 2x1):', round(time.time()-t,3), 's'); b.close()
" ) > $O/first_use.txt 2>&1
cat $O/first_use.txt
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
