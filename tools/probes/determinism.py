"""development: run the same engine twice and compare every chain (statistics, final mass matrix) — chains beyond the resident
blocks are handled in a block's later loop iterations"""
import sys, os
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
import nuts_rs_amd as N
from bench_configs import _k5_precision
kind, n, dim, draws = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
grid = int(sys.argv[5]) if len(sys.argv) > 5 else 0
s = N.DiagNutsSettings(num_chains=n, seed=20260928, num_tune=400)
logp = {"iid": lambda: N.LogpSpec.iid_normal(dim, 3.0), "mvn": lambda: N.LogpSpec.mvn_precision(_k5_precision(dim)),
        "diag": lambda: N.LogpSpec.diag_normal(np.exp(np.linspace(-2, 2, dim))), "funnel": lambda: N.LogpSpec.funnel(dim), "schools": lambda: N.LogpSpec.eight_schools()}[kind]()
groups = int(sys.argv[6]) if len(sys.argv) > 6 else 1
traj = int(sys.argv[7]) if len(sys.argv) > 7 else 0
if traj == 3:
    s = N.DiagMclmcSettings(num_chains=n, seed=20260928, num_tune=400)
elif traj:
    s = N.DiagNutsSettings(num_chains=n, seed=20260928, num_tune=400, trajectory_kind=traj)
res = []
for rep in range(2):
    b = N.ChainBatch(s, logp, n, chain_tiles=1, lane_groups=groups, grid_blocks=grid)
    b.set_position(b.init_positions_uniform())
    pos, st = b.draw_many(draws)
    sd, mu = b.mass_matrix()
    res.append((st, pos, sd, mu))
    b.close()
(st_a, pos_a, sd_a, mu_a), (st_b, pos_b, sd_b, mu_b) = res
bad = np.argwhere((st_a["n_steps"] != st_b["n_steps"]) | (st_a["energy"] != st_b["energy"]))
chains = sorted(set(int(c) for _, c in bad))
print(kind, n, dim, "grid", grid, "groups", groups, "traj", traj, "stats differ for", len(chains), "chains, min", chains[:1], "first draw", int(bad[:, 0].min()) if len(bad) else None,
      "| sigma differs:", int((sd_a != sd_b).any(axis=1).sum()), "| draw-0 positions differ:", int((pos_a[0] != pos_b[0]).any(axis=1).sum()))
