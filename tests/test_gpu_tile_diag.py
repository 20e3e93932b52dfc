"""DiagNutsSettings on the full-precision normal with the density's product on the matrix cores (nuts_tile_diag_kernel: 16 chains
per block, every chain with its own adapting diagonal mass matrix, one rendezvous per density evaluation): the same draws as the
one-chain kernels, i.e. the oracle's, bit for bit — warm-up, step-size searches inside the adaptation, ragged trees, partial tiles."""
import numpy as np
import pytest

import nuts_rs_amd as N
from helpers import assert_bit_exact, assert_vectors_bit_exact, oracle_settings

pytestmark = pytest.mark.gpu


def _prec(dim, seed):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(dim, dim))
    p = a @ a.T / dim + np.eye(dim)
    return (p + p.T) / 2


CASES = [
    # (id, dim, n_chains, num_tune, n_draws, settings kwargs)
    ("dim64_3chains", 64, 3, 80, 130, {}),
    ("dim104_20chains", 104, 20, 60, 100, {}),
    ("dim128_37chains", 128, 37, 50, 80, {}),
    ("dim256_k5_17chains", 256, 17, 40, 64, {}),
    ("dim200_options", 200, 16, 40, 70, dict(maxdepth=5, max_energy_error=2.0)),
    ("dim136_adam", 136, 9, 40, 70, dict(adam=True)),
    ("dim100_padded", 100, 18, 50, 80, {}),
    ("dim251_padded", 251, 16, 30, 50, {}),
    ("dim7_small", 7, 33, 60, 100, {}),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_tile_diag_parity_bit_exact(oracle, case):
    name, dim, n, tune, n_draws, opt = case
    kw = {k: v for k, v in opt.items() if k != "adam"}
    s = N.DiagNutsSettings(num_chains=n, seed=300 + dim, num_tune=tune, store_divergences=True, store_gradient=True, **kw)
    if opt.get("adam"):
        s.adapt_options.step_size_settings.method = N.STEP_ADAM
    logp = N.LogpSpec.mvn_precision(_prec(dim, dim))
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    b = N.ChainBatch(s, logp, n, chain_tiles=2, lane_groups=1)
    b.set_position(x0)
    h = n_draws // 2
    pos_a, st_a, vec_a = b.expanded_draw_many(h)
    pos_b, st_b, vec_b = b.expanded_draw_many(n_draws - h)
    assert b.tile_launches() >= 2 and b.threads_per_chain() == 64
    b.close()
    pos_g, st_g = np.concatenate([pos_a, pos_b]), np.concatenate([st_a, st_b])
    vec_g = {k: np.concatenate([vec_a[k], vec_b[k]]) for k in vec_a}
    vec_o = {}
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, dim, logp.params, oracle.gpu_cfg(64), n, x0, n_draws,
                                        n_threads=8, vectors=vec_o)
    assert failed == 0
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
    assert_vectors_bit_exact(vec_g, vec_o)
    if name == "dim200_options":
        assert st_g["diverging"].sum() > 0


def test_tile_diag_equals_one_chain_kernels_at_scale():
    """512 chains x dim 256 (BASELINE config 5's density with DiagNutsSettings): the matrix-core kernel (automatic from 256
    chains on) and the one-chain kernels give the same draws."""
    dim, n = 256, 512
    s = N.DiagNutsSettings(num_chains=n, seed=8, num_tune=60)
    logp = N.LogpSpec.mvn_precision(_prec(dim, 1))
    out = []
    for tiles in (0, 1):
        b = N.ChainBatch(s, logp, n, chain_tiles=tiles)
        b.set_position(b.init_positions_uniform())
        pos, st = b.draw_many(90)
        out.append((pos, st, b.tile_launches()))
        b.close()
    assert out[0][2] > 0 and out[1][2] == 0
    assert (out[0][0].view(np.uint64) == out[1][0].view(np.uint64)).all()
    for f in ("depth", "n_steps", "step_size", "energy", "logp", "diverging"):
        assert (out[0][1][f] == out[1][1][f]).all(), f
