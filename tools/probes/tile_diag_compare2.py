"""development: where do two runs of the matrix-core DiagNutsSettings kernel part?  statistics + mass matrix after `draws` draws"""
import sys, os
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
import nuts_rs_amd as N
from bench_configs import _k5_precision
n, dim, draws = int(sys.argv[1]), 256, int(sys.argv[2])
s = N.DiagNutsSettings(num_chains=n, seed=20260928, num_tune=400)
logp = N.LogpSpec.mvn_precision(_k5_precision(dim))
res = []
for tiles in (2, 2, 1, 1):
    b = N.ChainBatch(s, logp, n, chain_tiles=tiles)
    b.set_position(b.init_positions_uniform())
    pos, st = b.draw_many(draws)
    sd, mu = b.mass_matrix()
    res.append((st, pos, sd, mu, b.step_sizes()))
    b.close()
def cmp(name, A, B):
    st_a, pos_a, sd_a, mu_a, ss_a = A; st_b, pos_b, sd_b, mu_b, ss_b = B
    bad = np.argwhere((st_a["n_steps"] != st_b["n_steps"]) | (st_a["energy"] != st_b["energy"]))
    chains = sorted(set(int(c) for _, c in bad))
    print(name, "stats differ for chains", chains[:12], "first draw", int(bad[:, 0].min()) if len(bad) else None,
          "| sigma differs for chains", np.nonzero((sd_a != sd_b).any(axis=1))[0][:12].tolist(),
          "| pos differs first at draw", int(np.argwhere((pos_a != pos_b).any(axis=2))[:, 0].min()) if (pos_a != pos_b).any() else None)
    if chains:
        c = chains[0]; t = int(bad[bad[:, 1] == c][:, 0].min())
        for f in ("depth", "n_steps", "energy", "logp", "step_size", "mean_tree_accept", "transformation_update_id"):
            print("   ", f, st_a[f][max(t - 2, 0):t + 1, c], st_b[f][max(t - 2, 0):t + 1, c])
cmp("tile/tile", res[0], res[1]); cmp("one/one", res[2], res[3]); cmp("tile/one", res[0], res[2])
