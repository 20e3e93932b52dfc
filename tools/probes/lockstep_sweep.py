"""Lockstep kernel vs oracle over small settings (development probe, round 4): which is the smallest configuration that parts?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch  # noqa
import nuts_rs_amd as N
from oracle import oracle as O
from helpers import oracle_settings, STAT_FIELDS_EXACT, STAT_FIELDS_FLOAT
from test_gpu_lowrank import correlated_precision

def run(dim, rank, n_chains, draws, **kw):
    rng = np.random.default_rng(dim + rank)
    prec, sigma = correlated_precision(rng, dim, 4)
    w, u = np.linalg.eigh(sigma)
    keep = np.argsort(np.abs(np.log(w)))[::-1][:rank]
    tr = (np.exp(rng.normal(0, 0.2, dim)), rng.normal(0, 0.5, dim), w[keep], np.ascontiguousarray(u[:, keep].T), rng.normal(0, 0.1, dim))
    s = N.LowRankNutsSettings(num_chains=n_chains, seed=17, num_tune=40, freeze_transform=True, **kw)
    logp = N.LogpSpec.mvn_precision(prec)
    x0 = O.init_positions_uniform(s.seed, 0, n_chains, dim)
    b = N.ChainBatch(s, logp, n_chains)
    assert (b.set_position(x0) == 0).all()
    b.set_transform(*tr)
    pos, st = b.draw_many(draws)
    tpc, order = b.threads_per_chain(), b.reduce_order()
    b.close()
    pos_o, st_o, _, failed = O.run(oracle_settings(O, s), logp.kind, dim, logp.params, O.gpu_cfg(tpc, lr_seq_dots=order), n_chains, x0, draws, n_threads=8, transform=tr)
    bad = np.argwhere(pos.view(np.uint64) != pos_o.view(np.uint64))
    fields = [f for f in STAT_FIELDS_EXACT + STAT_FIELDS_FLOAT if not (((st[f] == st_o[f]) | ((st[f] != st[f]) & (st_o[f] != st_o[f]))).all())]
    first = {}
    for f in fields:
        m = np.argwhere(~((st[f] == st_o[f]) | ((st[f] != st[f]) & (st_o[f] != st_o[f]))))
        first[f] = (tuple(int(v) for v in m[0]), st[f][tuple(m[0])], st_o[f][tuple(m[0])])
    print(f"dim {dim} rank {rank} chains {n_chains} draws {draws} {kw}: order {order}; first pos mismatch {tuple(int(v) for v in bad[0]) if bad.size else None}; depth hist {np.bincount(st_o['depth'].ravel().astype(int)).tolist()}", flush=True)
    for f, v in first.items():
        print("      ", f, v)

for md in (1, 2, 3, 4, 10):
    run(64, 16, 16, 6, maxdepth=md)
run(64, 16, 16, 6, maxdepth=3, check_turning=False)
run(64, 16, 1, 6, maxdepth=3)
