import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nuts_rs_amd as N
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tune = int(sys.argv[3]) if len(sys.argv) > 3 else 400
draws = int(sys.argv[4]) if len(sys.argv) > 4 else 200
dpl = int(sys.argv[5]) if len(sys.argv) > 5 else 0
wpc = int(sys.argv[6]) if len(sys.argv) > 6 else 0
gb = int(sys.argv[7]) if len(sys.argv) > 7 else 0
s = N.DiagNutsSettings(num_chains=C, seed=20260928, num_tune=tune, num_draws=draws)
b = N.ChainBatch(s, N.LogpSpec.iid_normal(D, 3.0), C, dims_per_lane=dpl, waves_per_chain=wpc, grid_blocks=gb)
x0 = b.init_positions_uniform()
t = time.time(); st = b.set_position(x0); print('init', time.time()-t, (st != 0).sum())
t = time.time(); b.draw_device(tune); t1 = time.time()-t
c = b.counters(); print('warmup s', t1, c, 'M1 warm', c['total_leapfrogs']*D/t1)
b.reset_counters()
t = time.time(); b.draw_device(draws); t2 = time.time()-t
c = b.counters(); print('sample s', t2, c)
print('M1 = %.4g step*dims/s ; steps/draw/chain %.2f ; roofline frac (64B, 8TB/s) %.3f' % (c['total_leapfrogs']*D/t2, c['total_leapfrogs']/draws/C, c['total_leapfrogs']*D/t2*64/8e12))
print('step sizes', b.step_sizes()[:4])
