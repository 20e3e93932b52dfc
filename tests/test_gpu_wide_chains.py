"""Chains wider than one block (dim > 4096; the reference has no dimension limit): ceil(dim / 4096) co-resident blocks per
chain, every block sum exchanged between them (csrc/kern_cluster.hip, dev_math.hpp).  Engine (C ABI) against the oracle with
the matching reduction order (gpu_cfg(256, gpu_slice=4096)): draw for draw, bit for bit."""
import numpy as np
import pytest

import nuts_rs_amd as N
from helpers import assert_bit_exact, oracle_settings

pytestmark = pytest.mark.gpu

CASES = [
    # (id, dim, n_chains, num_tune, n_draws, density, settings kwargs)
    ("dim4097", 4097, 3, 30, 45, "iid", {}),
    ("dim5000", 5000, 3, 30, 45, "diag", {}),
    ("dim8192_exact_slices", 8192, 2, 30, 45, "iid", {}),
    ("dim10000", 10000, 3, 25, 40, "diag", {}),
    ("dim20000_adam", 20000, 2, 20, 32, "iid", dict(adam=True)),
    ("dim40000", 40000, 2, 16, 26, "diag", {}),
    ("dim65536", 65536, 1, 12, 20, "iid", {}),
    ("dim131072_max", 131072, 2, 8, 14, "diag", {}),
    ("dim9000_many_chains", 9000, 37, 16, 26, "iid", {}),
    ("dim6000_options", 6000, 3, 30, 45, "diag", dict(maxdepth=4, store=True)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_wide_chain_parity_bit_exact(oracle, case):
    name, dim, n, tune, n_draws, dens, opt = case
    s = N.DiagNutsSettings(num_chains=n, seed=100 + dim % 97, num_tune=tune, maxdepth=opt.get("maxdepth", 10),
                           store_divergences=bool(opt.get("store")), store_gradient=bool(opt.get("store")))
    if opt.get("adam"):
        s.adapt_options.step_size_settings.method = N.STEP_ADAM
    rng = np.random.default_rng(dim)
    logp = N.LogpSpec.iid_normal(dim, 3.0) if dens == "iid" else N.LogpSpec.diag_normal(np.exp(rng.uniform(-2, 2, dim)))
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    b = N.ChainBatch(s, logp, n)
    assert b.blocks_per_chain() == -(-dim // 4096) and b.threads_per_chain() == 256
    status = b.set_position(x0, raise_on_error=False)
    assert (status == 0).all()
    pos_g, st_g = b.draw_many(n_draws // 2)
    pos_g2, st_g2 = b.draw_many(n_draws - n_draws // 2)           # the chains' state survives the end of a launch
    pos_g, st_g = np.concatenate([pos_g, pos_g2]), np.concatenate([st_g, st_g2])
    sd, mu = b.mass_matrix()
    xs, gs, steps_g = b.positions(), b.gradients(), b.counters()["total_leapfrogs"]
    b.close()
    cfg = oracle.gpu_cfg(256, gpu_slice=4096)
    pos_o, st_o, steps, failed = oracle.run(oracle_settings(oracle, s), logp.kind, dim, logp.params, cfg, n, x0, n_draws, n_threads=8)
    assert failed == 0
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
    assert steps_g == steps
    assert (xs == pos_o[-1]).all()                                 # the gathered per-chain vectors
    assert np.isfinite(sd).all() and np.isfinite(mu).all() and np.isfinite(gs).all()


def test_wide_chain_posterior_and_unsupported(oracle):
    dim, n = 6000, 16
    s = N.DiagNutsSettings(num_chains=n, seed=5, num_tune=150)
    b = N.ChainBatch(s, N.LogpSpec.iid_normal(dim, 3.0), n)
    b.set_position(b.init_positions_uniform())
    pos, st = b.draw_many(230)
    b.close()
    post = pos[150:]
    assert abs(post.mean() - 3.0) < 0.01 and abs(post.var() - 1.0) < 0.02 and st["diverging"][150:].sum() == 0
    with pytest.raises(N.NutsAmdError):
        N.ChainBatch(s, N.LogpSpec.funnel(5000), 2)                # not an element-wise density
    with pytest.raises(N.NutsAmdError):
        N.ChainBatch(N.LowRankNutsSettings(num_chains=2), N.LogpSpec.iid_normal(5000, 0.0), 2)
    with pytest.raises(N.NutsAmdError):
        N.ChainBatch(s, N.LogpSpec.iid_normal(131073, 0.0), 2)     # > 32 blocks per chain



def _coupled(chain, x):
    """a non-element-wise density: a random-walk prior (every coordinate tied to its neighbour, across the blocks' slice borders)
    plus a weak quartic well; deterministic numpy arithmetic so that the engine's and the oracle's calls agree bit for bit"""
    d = np.diff(x)
    lp = -0.5 * float(np.dot(d, d)) * 4.0 - 0.5 * float(np.dot(x, x)) * 0.25 - 0.01 * float(np.sum(x ** 4))
    g = -0.25 * x - 0.04 * x ** 3
    g[:-1] += 4.0 * d
    g[1:] -= 4.0 * d
    return lp, g


@pytest.mark.parametrize("dim,kind", [(4500, "nuts"), (9000, "nuts"), (12289, "nuts"), (4600, "micro"), (5000, "mclmc")],
                         ids=["dim4500", "dim9000", "dim12289", "dim4600_microcanonical", "dim5000_mclmc"])
def test_wide_chain_host_callback_bit_exact(oracle, dim, kind):
    """Any density for a wide chain: the members of a chain share its mailbox (each writes its slice of the position, the first
    rings, each reads its slice of the gradient) — against the oracle's callback chain on the same Python function."""
    n, tune, draws = 3, 12, 20
    if kind == "mclmc":
        s = N.DiagMclmcSettings(num_chains=n, seed=dim, num_tune=tune, step_size=0.05, momentum_decoherence_length=1.0)
    else:
        s = N.DiagNutsSettings(num_chains=n, seed=dim, num_tune=tune, maxdepth=5,
                               trajectory_kind=N.KineticEnergyKind.MICROCANONICAL if kind == "micro" else N.KineticEnergyKind.EUCLIDEAN)
    calls = {"n": 0}

    def counted(chain, x):
        calls["n"] += 1
        return _coupled(chain, x)
    logp = N.LogpSpec.host_callback(dim, counted, threads=2)
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    b = N.ChainBatch(s, logp, n)
    assert b.blocks_per_chain() == -(-dim // 4096)
    assert (b.set_position(x0, raise_on_error=False) == 0).all()
    pos_a, st_a = b.draw_many(draws // 2)
    pos_b, st_b = b.draw_many(draws - draws // 2)
    pos_g, st_g = np.concatenate([pos_a, pos_b]), np.concatenate([st_a, st_b])
    steps_g, host_calls = b.counters()["total_leapfrogs"], b.host_logp_calls()
    b.close()
    assert host_calls == calls["n"] and host_calls >= steps_g + n          # one call per chain per gradient, not one per block

    def tramp(ctx, d, px, pg, plogp):
        lp, g = _coupled(0, np.ctypeslib.as_array(px, shape=(d,)).copy())
        np.ctypeslib.as_array(pg, shape=(d,))[:] = g
        plogp[0] = lp
        return 0
    cb = oracle.HOST_LOGP_FN(tramp)
    so = oracle_settings(oracle, s)
    cfg = oracle.gpu_cfg(256, gpu_slice=4096)
    for c in range(n):
        ch = oracle.Chain(so, 0, dim, np.zeros(1), cfg, chain_id=c, callback=cb)
        assert ch.set_position(x0[c]) == 0
        for t in range(draws):
            p, q, rc = ch.draw()
            assert rc == 0
            assert (p.view(np.uint64) == pos_g[t, c].view(np.uint64)).all(), (c, t)
            for f in ("depth", "n_steps", "diverging", "step_size", "energy", "logp", "mean_tree_accept"):
                assert q[f] == st_g[f][t, c], (f, c, t)


def test_wide_chain_host_callback_errors(oracle):
    """The reference's error taxonomy survives the shared mailbox: a recoverable error is a divergence, an unrecoverable one stops
    the chain (all of its blocks), the other chains go on."""
    dim, n = 5000, 3
    s = N.DiagNutsSettings(num_chains=n, seed=8, num_tune=10, maxdepth=4)
    state = {"calls": 0}

    def flaky(chain, x):
        state["calls"] += 1
        if chain == 1 and state["calls"] > 40:
            raise ValueError("unrecoverable")
        if chain == 2 and abs(x[4999]) > 2.5:
            raise N.RecoverableLogpError()
        return -0.5 * float(np.dot(x, x)), -x
    b = N.ChainBatch(s, N.LogpSpec.host_callback(dim, flaky, threads=1), n)
    assert (b.set_position(b.init_positions_uniform(), raise_on_error=False) == 0).all()
    pos, st = b.draw_many(16, raise_on_error=False)
    b.close()
    cs = st["chain_status"]
    assert (cs[:, 0] == 0).all() and (cs[:, 2] == 0).all()
    stop = np.flatnonzero(cs[:, 1] == 2)                           # NM_CHAIN_LOGP_FATAL in the draw where the chain stops ...
    assert len(stop) == 1 and (st["n_steps"][stop[0] + 1:, 1] == 0).all()      # ... and nothing runs afterwards
    assert np.isfinite(pos[:, 0]).all() and (st["n_steps"][:, 0] > 0).all()


KIND_CASES = [
    # (id, settings, dim, density)
    ("exact_dim5000", lambda n: N.DiagNutsSettings(num_chains=n, seed=71, num_tune=24, trajectory_kind=N.KineticEnergyKind.EXACT_NORMAL), 5000, "diag"),
    ("micro_dim5000", lambda n: N.DiagNutsSettings(num_chains=n, seed=72, num_tune=24, trajectory_kind=N.KineticEnergyKind.MICROCANONICAL, maxdepth=6), 5000, "diag"),
    ("micro_dim12289", lambda n: N.DiagNutsSettings(num_chains=n, seed=73, num_tune=20, trajectory_kind=N.KineticEnergyKind.MICROCANONICAL, maxdepth=5), 12289, "iid"),
    ("mclmc_default_dim5000", lambda n: N.DiagMclmcSettings(num_chains=n, seed=74, num_tune=24, step_size=0.5), 5000, "iid"),
    ("mclmc_micro_dim9000", lambda n: N.DiagMclmcSettings(num_chains=n, seed=75, num_tune=20, step_size=0.4,
                                                          trajectory_kind=N.MclmcTrajectoryKind.MICROCANONICAL), 9000, "diag"),
    ("mclmc_euclid_ladder_dim4500", lambda n: N.DiagMclmcSettings(num_chains=n, seed=76, num_tune=20, step_size=1.5, max_energy_error=3.0,
                                                                  trajectory_kind=N.MclmcTrajectoryKind.EUCLIDEAN), 4500, "diag"),
]


@pytest.mark.parametrize("case", KIND_CASES, ids=[c[0] for c in KIND_CASES])
def test_wide_chain_trajectory_kinds_and_mclmc_bit_exact(oracle, case):
    """`trajectory_kind` (ExactNormal, Microcanonical) and the MCLMC sampler for chains wider than one block: the geodesic / ESH
    leapfrogs' sums, the normalisations and the partial momentum refresh go through the same exchange (csrc/kern_cluster_kin.hip)."""
    name, make, dim, dens = case
    n, n_draws = 3, 36
    s = make(n)
    rng = np.random.default_rng(dim)
    logp = N.LogpSpec.iid_normal(dim, 3.0) if dens == "iid" else N.LogpSpec.diag_normal(np.exp(rng.uniform(-2, 2, dim)))
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    b = N.ChainBatch(s, logp, n)
    assert b.blocks_per_chain() == -(-dim // 4096)
    assert (b.set_position(x0, raise_on_error=False) == 0).all()
    pos_a, st_a = b.draw_many(n_draws // 3)
    pos_b, st_b = b.draw_many(n_draws - n_draws // 3)
    pos_g, st_g = np.concatenate([pos_a, pos_b]), np.concatenate([st_a, st_b])
    steps_g = b.counters()["total_leapfrogs"]
    b.close()
    cfg = oracle.gpu_cfg(256, gpu_slice=4096)
    pos_o, st_o, steps, failed = oracle.run(oracle_settings(oracle, s), logp.kind, dim, logp.params, cfg, n, x0, n_draws, n_threads=8)
    assert failed == 0
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
    assert steps_g == steps
    if name.startswith("mclmc"):
        for f in ("energy_change", "average_step_size"):
            assert ((st_g[f] == st_o[f]) | (np.isnan(st_g[f]) & np.isnan(st_o[f]))).all(), f
    if "ladder" in name:
        assert (st_g["average_step_size"] < 1.5 * (1 - 1e-12)).sum() > 0        # the halve-and-retry ladder ran


def test_general_exchange_protocol_gives_the_same_draws(oracle, monkeypatch):
    """The fallback of the lock-free exchange — release / acquire on an arrival counter, taken where a chain's blocks do not share
    an XCD — forced for a whole run (NM_CLUSTER_GENERAL=1): the same bits."""
    dim, n, tune, draws = 9000, 5, 16, 26
    s = N.DiagNutsSettings(num_chains=n, seed=17, num_tune=tune)
    logp = N.LogpSpec.diag_normal(np.exp(np.random.default_rng(2).uniform(-2, 2, dim)))
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    monkeypatch.setenv("NM_CLUSTER_GENERAL", "1")
    b = N.ChainBatch(s, logp, n)
    b.set_position(x0)
    pos_g, st_g = b.draw_many(draws)
    b.close()
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, dim, logp.params, oracle.gpu_cfg(256, gpu_slice=4096), n, x0, draws, n_threads=8)
    assert failed == 0
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
