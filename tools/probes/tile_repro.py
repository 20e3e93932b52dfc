import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import nuts_rs_amd as N
from test_gpu_lowrank import correlated_precision
dim, rank, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(1)
prec, sigma = correlated_precision(rng, dim, 4)
w, u = np.linalg.eigh(sigma)
keep = np.argsort(np.abs(np.log(w)))[::-1][:rank]
tr = (np.ones(dim), np.zeros(dim), w[keep], np.ascontiguousarray(u[:, keep].T), np.zeros(dim))
s = N.LowRankNutsSettings(num_chains=n, seed=17, num_tune=20, freeze_transform=True)
b = N.ChainBatch(s, N.LogpSpec.mvn_precision(prec), n)
b.set_position(b.init_positions_uniform())
b.set_transform(*tr)
print("launching", flush=True)
pos, st = b.draw_many(int(sys.argv[4]) if len(sys.argv) > 4 else 3)
print("tile launches", b.tile_launches(), st["depth"].mean(), st["n_steps"].sum(), np.isfinite(pos).all())
