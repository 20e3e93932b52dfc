"""Where does the matrix-core kernel (DPL 2: dims 64 / 128) part from the oracle?  (round 4: test_matrix_core_kernel_bit_exact[64-16-21] failed after a rebuild)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch  # noqa
import nuts_rs_amd as N
from oracle import oracle as O
from helpers import oracle_settings
from test_gpu_lowrank import correlated_precision

def run(dim, rank, n_chains, splits, chain_tiles=0, draws=60):
    rng = np.random.default_rng(dim + rank)
    prec, sigma = correlated_precision(rng, dim, 4)
    w, u = np.linalg.eigh(sigma)
    keep = np.argsort(np.abs(np.log(w)))[::-1][:rank]
    tr = (np.exp(rng.normal(0, 0.2, dim)), rng.normal(0, 0.5, dim), w[keep], np.ascontiguousarray(u[:, keep].T), rng.normal(0, 0.1, dim))
    s = N.LowRankNutsSettings(num_chains=n_chains, seed=17, num_tune=40, freeze_transform=True)
    logp = N.LogpSpec.mvn_precision(prec)
    x0 = O.init_positions_uniform(s.seed, 0, n_chains, dim)
    b = N.ChainBatch(s, logp, n_chains, chain_tiles=chain_tiles)
    assert (b.set_position(x0) == 0).all()
    b.set_transform(*tr)
    cuts = [0] + [c for c in splits if 0 < c < draws] + [draws]
    parts = [b.draw_many(hi - lo) for lo, hi in zip(cuts[:-1], cuts[1:])]
    pos, st = np.concatenate([p for p, _ in parts]), np.concatenate([q for _, q in parts])
    tpc, tiles = b.threads_per_chain(), b.tile_launches()
    b.close()
    pos_o, st_o, _, failed = O.run(oracle_settings(O, s), logp.kind, dim, logp.params, O.gpu_cfg(tpc, lr_seq_dots=int(tiles > 0)), n_chains, x0, draws, n_threads=8, transform=tr)
    bad = np.argwhere(pos.view(np.uint64) != pos_o.view(np.uint64))
    badn = np.argwhere(st["n_steps"] != st_o["n_steps"])
    first = tuple(bad[0]) if bad.size else None
    chains_bad = sorted(set(bad[:, 1].tolist())) if bad.size else []
    print(f"dim {dim} rank {rank} chains {n_chains} splits {splits} chain_tiles {chain_tiles}: tile launches {tiles}, first pos mismatch {first}, first n_steps mismatch {tuple(badn[0]) if badn.size else None}, bad chains {chains_bad[:20]} ({len(chains_bad)})", flush=True)
    if bad.size:
        d, c, _ = bad[0]
        for f in ("depth", "n_steps", "step_size", "energy", "logp"):
            print("   ", f, st[f][d, c], st_o[f][d, c])

for args in [(64, 16, 21, (1, 33)), (64, 16, 21, ()), (64, 16, 21, (33,)), (64, 16, 21, (1,)), (64, 16, 16, ()), (64, 16, 1, ()), (64, 16, 21, (1, 33), 1),
             (128, 64, 40, ()), (128, 128, 16, ()), (200, 200, 16, (1, 33)), (64, 64, 16, ())]:
    run(*args)
