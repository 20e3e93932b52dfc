"""where does the Euclidean group kernel leave the wave kernel?  8 schools, 64 chains"""
import sys, os
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nuts_rs_amd as N
n = 64
for tune in (40,):
    s = N.DiagNutsSettings(num_chains=n, seed=5, num_tune=tune)
    out = {}
    for lg in (1, 2):
        b = N.ChainBatch(s, N.LogpSpec.eight_schools(), n, lane_groups=lg)
        b.set_position(b.init_positions_uniform())
        out[lg] = b.draw_many(tune + 30)
        print("tune", tune, "lane_groups", lg, "group launches", b.group_launches(), flush=True)
        b.close()
    (p1, s1), (p2, s2) = out[1], out[2]
    bad = np.argwhere((p1 != p2).any(axis=2))
    print("tune", tune, "first position mismatch (draw, chain):", bad[0] if len(bad) else None)
    for f in ("depth", "n_steps", "step_size", "energy", "logp", "diverging", "mean_tree_accept"):
        d = np.argwhere(s1[f] != s2[f])
        if len(d):
            t, c = d[0]
            print("  stat", f, "first at", (t, c), s1[f][t, c], "vs", s2[f][t, c])
