#!/bin/bash
# round 6 ("done" clause): a library with ONE deliberately wrong instantiation — kern_lr_iid_normal.hip@small rebuilt with
# -DNM_X_SABOTAGE_LR41 (nuts_draw_kernel<4,1,LrWrap<IidNormal>> halves the main tree's log size after every doubling: the failing kernel and the
# lost quantity of DESIGN §22's fourth incident), every other object the regular one — must be rejected by build()'s check AND at first use.
export TMPDIR=/tmp; O=$PWD/gpurun_out/r06p; mkdir -p $O
export NUTS_AMD_LIB=$PWD/nuts_rs_amd/libnuts_amd_sab.so
echo "== build()'s check (nuts_rs_amd.selftest.run_all) on the sabotaged library" > $O/sabotage.txt
NUTS_AMD_SELFTEST=0 python -c "
import nuts_rs_amd.selftest as s
try:
    print(s.run_all(), 'instantiations ok  <-- NOT REJECTED')
except s.SelfTestError as e:
    print('REJECTED:', str(e)[:1500])
" >> $O/sabotage.txt 2>&1
echo "== first use: an engine of the sabotaged instantiation, then engines of its neighbours" >> $O/sabotage.txt
python -c "
import nuts_rs_amd as N
from nuts_rs_amd.selftest import SelfTestError
def mk(settings, dim):
    try:
        b = N.ChainBatch(settings(num_chains=4, seed=1), N.LogpSpec.iid_normal(dim, 1.0), 4); b.close(); return 'engine created'
    except SelfTestError as e:
        return 'REJECTED at first use: ' + str(e)[:300]
print('LowRankNutsSettings, iid dim 200 (4,1):', mk(N.LowRankNutsSettings, 200))
print('LowRankNutsSettings, iid dim 100 (2,1):', mk(N.LowRankNutsSettings, 100))
print('DiagNutsSettings,    iid dim 200 (4,1):', mk(N.DiagNutsSettings, 200))
" >> $O/sabotage.txt 2>&1
cat $O/sabotage.txt
