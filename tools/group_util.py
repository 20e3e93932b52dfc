"""Lockstep utilisation of the 8-chains-per-wavefront kernel: chain-leapfrogs over 8 x (the longest tree of each wavefront)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch, time
import nuts_rs_amd as N
for name, logp in (("schools", N.LogpSpec.eight_schools()), ("iid10", N.LogpSpec.iid_normal(10, 3.0))):
    C = 65536
    s = N.DiagNutsSettings(num_chains=C, seed=1, num_tune=400)
    b = N.ChainBatch(s, logp, C)
    b.set_position(b.init_positions_uniform())
    _, sw = b.draw_many(400, positions=False); tw = b.counters()["kernel_ms"] / 1e3; b.reset_counters()
    _, st = b.draw_many(100, positions=False); ts = b.counters()["kernel_ms"] / 1e3
    for nm, a in (("warm", sw), ("samp", st)):
        n = a["n_steps"].astype(np.float64).reshape(a.shape[0], C // 8, 8)
        print(name, nm, "steps/draw %.2f" % n.mean(), "lockstep util %.3f" % (n.sum() / (8 * n.max(axis=2).sum())),
              "leapfrogs %.3g" % n.sum(), "kernel s %.4f" % (tw if nm == "warm" else ts), "leapfrogs/s %.3g" % (n.sum() / (tw if nm == "warm" else ts)))
    b.close()
