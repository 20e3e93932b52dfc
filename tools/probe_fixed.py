import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nuts_rs_amd as N
def run(label, C, D, **kw):
    s = N.DiagNutsSettings(num_chains=C, seed=20260928, num_tune=100, num_draws=200, **kw)
    b = N.ChainBatch(s, N.LogpSpec.iid_normal(D, 3.0), C)
    nb = 1024
    b.set_position(b.init_positions_uniform())
    b.draw_device(100); b.reset_counters()
    t = time.time(); b.draw_device(200); dt = time.time() - t
    c = b.counters()
    rounds = -(-C // nb)
    print('%-34s us/draw/block %.2f  steps/draw %.2f' % (label, dt*1e6/200/rounds, c['total_leapfrogs']/200/C))
    b.close()
for D in (128, 256, 512, 1024):
    run('D=%d maxdepth1' % D, 4096, D, maxdepth=1)
    run('D=%d maxdepth2' % D, 4096, D, maxdepth=2, check_turning=False)
