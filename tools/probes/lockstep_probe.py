"""Lockstep kernel vs oracle, field by field for the first draws (development probe, round 4)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch  # noqa
import nuts_rs_amd as N
from oracle import oracle as O
from helpers import oracle_settings
from test_gpu_lowrank import correlated_precision

dim, rank, n_chains, draws = (int(a) for a in (sys.argv[1:5] + [200, 200, 16, 3][len(sys.argv) - 1:]))
tune = int(sys.argv[5]) if len(sys.argv) > 5 else 40
rng = np.random.default_rng(dim + rank)
prec, sigma = correlated_precision(rng, dim, 4)
w, u = np.linalg.eigh(sigma)
keep = np.argsort(np.abs(np.log(w)))[::-1][:rank]
tr = (np.exp(rng.normal(0, 0.2, dim)), rng.normal(0, 0.5, dim), w[keep], np.ascontiguousarray(u[:, keep].T), rng.normal(0, 0.1, dim))
s = N.LowRankNutsSettings(num_chains=n_chains, seed=17, num_tune=tune, freeze_transform=True)
logp = N.LogpSpec.mvn_precision(prec)
x0 = O.init_positions_uniform(s.seed, 0, n_chains, dim)
b = N.ChainBatch(s, logp, n_chains)
assert (b.set_position(x0) == 0).all()
b.set_transform(*tr)
pos, st = b.draw_many(draws)
tpc, order = b.threads_per_chain(), b.reduce_order()
print("order", order, "lockstep launches", b.lockstep_launches())
b.close()
pos_o, st_o, _, failed = O.run(oracle_settings(O, s), logp.kind, dim, logp.params, O.gpu_cfg(tpc, lr_seq_dots=order), n_chains, x0, draws, n_threads=8, transform=tr)
for t in range(min(draws, 3)):
    for c in range(min(n_chains, 3)):
        print(f"--- draw {t} chain {c}")
        for f in st.dtype.names:
            a, bb = st[f][t, c], st_o[f][t, c]
            flag = "" if (a == bb or (a != a and bb != bb)) else "   <<<<"
            print(f"   {f:28s} {a!r:28} {bb!r:28}{flag}")
        d = np.abs(pos[t, c] - pos_o[t, c])
        print("   pos max abs diff", d.max(), "at", int(d.argmax()), "gpu", pos[t, c][:4], "oracle", pos_o[t, c][:4])
bad = np.argwhere(pos.view(np.uint64) != pos_o.view(np.uint64))
print("first pos mismatch", tuple(bad[0]) if bad.size else None, "of", pos.shape)
