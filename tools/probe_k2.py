import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nuts_rs_amd as N
C, D = 4096, 1024
def run(label, **kw):
    s = N.DiagNutsSettings(num_chains=C, seed=20260928, num_tune=400, num_draws=200, **kw)
    b = N.ChainBatch(s, N.LogpSpec.iid_normal(D, 3.0), C)
    b.set_position(b.init_positions_uniform())
    b.draw_device(400); b.reset_counters()
    t = time.time(); b.draw_device(200); dt = time.time() - t
    c = b.counters()
    print('%-28s %.4g step*dims/s  steps/draw %.2f  us/leaf/block %.2f' % (label, c['total_leapfrogs']*D/dt, c['total_leapfrogs']/200/C, dt*1e6*1024/c['total_leapfrogs']))
    b.close()
run('default')
run('no_check maxdepth4', check_turning=False, maxdepth=4)
run('no_check maxdepth6', check_turning=False, maxdepth=6)
run('no_check maxdepth2', check_turning=False, maxdepth=2)
run('check maxdepth2', maxdepth=2)
