"""development: the matrix-core kernel of DiagNutsSettings x full-precision normal against the one-chain kernels at scale, twice"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nuts_rs_amd as N
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
from bench_configs import _k5_precision

n, dim, draws = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 256, int(sys.argv[2]) if len(sys.argv) > 2 else 450
s = N.DiagNutsSettings(num_chains=n, seed=20260928, num_tune=400)
logp = N.LogpSpec.mvn_precision(_k5_precision(dim))
res = []
for tiles in (2, 2, 1):
    b = N.ChainBatch(s, logp, n, chain_tiles=tiles)
    b.set_position(b.init_positions_uniform())
    _, st = b.draw_many(draws, positions=False)
    res.append(st)
    b.close()
for name, a, c in (("tile vs tile", res[0], res[1]), ("tile vs one-chain", res[0], res[2])):
    bad = np.argwhere((a["n_steps"] != c["n_steps"]) | (a["energy"] != c["energy"]))
    print(name, "differing (draw, chain) entries:", len(bad))
    if len(bad):
        first = {}
        for t, ch in bad:
            first.setdefault(int(ch), int(t))
        print(" chains:", len(first), "first differences:", sorted(first.items(), key=lambda kv: kv[1])[:10])
        ch = np.array(sorted(first))
        print(" chain min/max", ch.min(), ch.max(), "tiles", sorted(set((ch // 16).tolist()))[:40], "waves", np.bincount(ch % 16, minlength=16).tolist())
        print(" first-diff draws", np.bincount(np.array(list(first.values()))).tolist()[:20])
        t, ch = sorted(first.items(), key=lambda kv: kv[1])[0][1], sorted(first.items(), key=lambda kv: kv[1])[0][0]
        for f in ("depth", "n_steps", "energy", "logp", "step_size", "diverging", "mean_tree_accept", "divergence_energy_error"):
            print("  ", f, a[f][t, ch], c[f][t, ch], "prev:", a[f][t - 1, ch], c[f][t - 1, ch])
