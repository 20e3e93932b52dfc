"""banana behind LowRankNutsSettings through the host callback: the moments with the estimator on the host and on the device, several seeds"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import nuts_rs_amd as N
from test_gpu_host_callback import banana
for seed in (4, 5, 6):
    for place in ("host", "device"):
        s = N.LowRankNutsSettings(num_chains=8, seed=seed, num_tune=200)
        b = N.ChainBatch(s, N.LogpSpec.host_callback(6, banana, threads=4), 8)
        b.set_lowrank_estimator_place(place)
        b.init_with_retries()
        pos, st = b.draw_many(400)
        n_eig = b.lowrank()[0]
        b.close()
        x = pos[200:].reshape(-1, 6)
        print(seed, place, "status", int((st["chain_status"] != 0).sum()), "mean0 %.3f var0 %.3f resid %.3f" % (x[:, 0].mean(), x[:, 0].var(), (x[:, 1] - x[:, 0] ** 2).mean()),
              "div", int(st["diverging"][200:].sum()), "steps %.1f" % st["n_steps"][200:].mean(), "n_eig", n_eig.tolist(), flush=True)
