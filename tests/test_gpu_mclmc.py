"""GPU parity for `MclmcChain` (reference src/mclmc.rs; `DiagMclmcSettings`, src/sampler.rs:266-470 — experimental upstream):
the engine's NM_SAMPLER_MCLMC (C ABI) against the oracle's restatement on the same seeds, draw for draw and bit for bit —
the three MclmcTrajectoryKinds, the Euclidean -> Microcanonical switch, the halve-and-retry ladder of dynamic_step_size, real
divergences (the chain stays and resamples its momentum), the momentum that survives from draw to draw."""
import numpy as np
import pytest

import nuts_rs_amd as N
from helpers import assert_bit_exact, assert_vectors_bit_exact, oracle_settings, run_engine, run_oracle

pytestmark = pytest.mark.gpu

K = N.MclmcTrajectoryKind

CASES = [
    # (id, settings kwargs, dim, n_chains, n_draws, density, tiling)
    ("default_iid_dim10", dict(seed=1, num_tune=200), 10, 6, 320, "iid", 0),
    ("micro_iid_dim50", dict(seed=2, num_tune=100, trajectory_kind=K.MICROCANONICAL), 50, 5, 160, "iid", 0),
    ("euclid_iid_dim50", dict(seed=3, num_tune=100, trajectory_kind=K.EUCLIDEAN, step_size=0.3), 50, 5, 160, "iid", 0),
    ("switch_at_zero", dict(seed=4, num_tune=60, trajectory_switch_fraction=0.0), 12, 4, 100, "iid", 0),
    ("dim2", dict(seed=5, num_tune=60, trajectory_kind=K.MICROCANONICAL), 2, 4, 120, "iid", 0),
    ("subsample_every_step", dict(seed=6, num_tune=80, subsample_frequency=0.0), 20, 4, 150, "iid", 0),
    ("long_trajectories", dict(seed=7, num_tune=60, momentum_decoherence_length=12.0, step_size=0.25), 20, 4, 100, "diag", 0),
    ("funnel_ladder", dict(seed=8, num_tune=120, step_size=0.8, max_energy_error=2.0), 11, 10, 220, "funnel", 0),
    ("funnel_ladder_micro", dict(seed=9, num_tune=120, step_size=0.8, max_energy_error=2.0, trajectory_kind=K.MICROCANONICAL), 11, 10, 220, "funnel", 0),
    ("funnel_static_divergences", dict(seed=10, num_tune=120, step_size=0.8, max_energy_error=40.0, dynamic_step_size=False), 11, 10, 220, "funnel", 0),
    ("schools", dict(seed=11, num_tune=150, step_size=0.4, max_energy_error=30.0), 10, 12, 250, "schools", 0),
    ("mvn_dim64", dict(seed=12, num_tune=80), 64, 4, 130, "mvn", 0),
    ("diag_dim130_dpl4", dict(seed=13, num_tune=150, step_size=0.4), 130, 4, 220, "diag", 4),
    ("dim1024_dpl16", dict(seed=14, num_tune=30), 1024, 3, 45, "iid", 16),
    ("dim700_w2", dict(seed=15, num_tune=30, trajectory_kind=K.MICROCANONICAL), 700, 3, 45, "diag", (8, 2)),
    ("dim2000_w2_dpl16", dict(seed=16, num_tune=20), 2000, 2, 30, "iid", (16, 2)),
]


def _density(dens, dim, rng):
    if dens == "iid":
        return N.LogpSpec.iid_normal(dim, 3.0)
    if dens == "funnel":
        return N.LogpSpec.funnel(dim)
    if dens == "schools":
        return N.LogpSpec.eight_schools()
    if dens == "mvn":
        a = rng.normal(size=(dim, dim))
        p = a @ a.T / dim + np.eye(dim)
        return N.LogpSpec.mvn_precision((p + p.T) / 2)
    return N.LogpSpec.diag_normal(np.exp(rng.uniform(-3, 3, dim)))


def _runs():
    out = []
    for c in CASES:
        out.append(pytest.param(c, 1, id=c[0] + "-wave"))
        if c[2] <= 64 and c[6] == 0:                 # the small-chain kernels: 8 / 4 / 2 chains per wavefront (nuts_group.hpp), since round 4
            out.append(pytest.param(c, 2, id=c[0] + "-group"))
    return out


@pytest.mark.parametrize("case,lane_groups", _runs())
def test_mclmc_parity_bit_exact(oracle, case, lane_groups):
    name, kw, dim, n_chains, n_draws, dens, tiling = case
    s = N.DiagMclmcSettings(num_chains=n_chains, **kw)
    logp = _density(dens, dim, np.random.default_rng(kw["seed"]))
    x0 = oracle.init_positions_uniform(s.seed, 0, n_chains, dim)
    dpl, wpc = tiling if isinstance(tiling, tuple) else (tiling, 0)
    pos_g, st_g, ex = run_engine(s, logp, n_chains, x0, n_draws, dims_per_lane=dpl, waves_per_chain=wpc, lane_groups=lane_groups,
                                 splits=(n_draws // 3,))                 # the momentum survives the end of a launch too
    assert (ex["group_launches"] >= 1) == (lane_groups == 2)
    pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, n_chains, x0, n_draws, gpu_threads=ex["threads_per_chain"])
    assert failed == 0 and (ex["status"] == 0).all()
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
    for f in ("energy_change", "average_step_size"):
        a, b = st_g[f], st_o[f]
        assert ((a == b) | (np.isnan(a) & np.isnan(b))).all(), f
    assert ex["counters"]["total_leapfrogs"] == steps
    if name.startswith("funnel_ladder"):
        # the ladder ran: draws whose average step size is below the base step, and (max_halvings exhausted) real divergences
        assert (st_g["average_step_size"] < 0.8 * (1 - 1e-12)).sum() > 0
        assert (st_g["depth"] > np.round(3.0 / 0.8)).sum() > 0
    if name == "funnel_static_divergences":
        assert st_g["diverging"].sum() > 0


def test_mclmc_reference_envelope(oracle):
    """The reference's own tests (src/mclmc.rs:566-680): 10-dim N(3, 1), the three trajectory kinds, no divergence, the chain
    near the mean — here on 256 chains, with the moments checked."""
    dim, n = 10, 256
    for kind, step in ((K.MICROCANONICAL, 0.5), (K.EUCLIDEAN, 0.3), (K.EUCLIDEAN_EARLY_THEN_MICROCANONICAL, 0.5)):
        s = N.DiagMclmcSettings(num_chains=n, seed=21 + kind, num_tune=200, num_draws=500, step_size=step,
                                momentum_decoherence_length=3.0, trajectory_kind=kind)
        b = N.ChainBatch(s, N.LogpSpec.iid_normal(dim, 3.0), n)
        b.set_position(np.zeros((n, dim)))
        pos, st = b.draw_many(700)
        b.close()
        assert st["diverging"].sum() == 0
        post = pos[200:]
        assert abs(post.mean() - 3.0) < 0.05
        assert abs(post.var() - 1.0) < (0.25 if kind == K.EUCLIDEAN else 0.08)     # unadjusted: a step-size bias remains


def test_mclmc_vector_statistics(oracle):
    dim, n = 11, 6
    s = N.DiagMclmcSettings(num_chains=n, seed=31, num_tune=60, step_size=0.8, max_energy_error=2.0, dynamic_step_size=False,
                            store_gradient=True, store_unconstrained=True, store_transformed=True, store_divergences=True)
    s.adapt_options.mass_matrix_options.store_mass_matrix = True
    logp = N.LogpSpec.funnel(dim)
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    b = N.ChainBatch(s, logp, n, lane_groups=1)
    b.set_position(x0)
    pos_g, st_g, vec_g = b.expanded_draw_many(120)
    b.close()
    vec_o = {}
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, dim, logp.params, oracle.gpu_cfg(64), n, x0, 120,
                                        n_threads=4, vectors=vec_o)
    assert failed == 0
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
    assert_vectors_bit_exact(vec_g, vec_o)
    assert st_g["diverging"].sum() > 0


def test_mclmc_unsupported_combinations_fail_loudly():
    with pytest.raises(N.NutsAmdError):
        N.ChainBatch(N.DiagMclmcSettings(num_chains=2), N.LogpSpec.iid_normal(1, 0.0), 2)
    with pytest.raises(N.NutsAmdError):
        N.ChainBatch(N.DiagMclmcSettings(num_chains=2, step_size=0.0), N.LogpSpec.iid_normal(4, 0.0), 2)
    with pytest.raises(N.NutsAmdError):
        N.ChainBatch(N.DiagMclmcSettings(num_chains=2, trajectory_kind=9), N.LogpSpec.iid_normal(4, 0.0), 2)


@pytest.mark.parametrize("kind", [K.EUCLIDEAN_EARLY_THEN_MICROCANONICAL, K.MICROCANONICAL], ids=["default_kind", "microcanonical"])
def test_lowrank_mclmc_bit_exact(oracle, kind):
    """`LowRankMclmcSettings` (src/sampler.rs:325-328, :376-384): MclmcChain with the low-rank adaptation — the pause for the host's
    estimator falls in the middle of MclmcChain::draw's adapt call — with the oracle's estimator injected on both sides."""
    import ctypes as C
    from nuts_rs_amd import _lib
    from oracle import lowrank as LR
    rng = np.random.default_rng(3)
    dim, n, draws = 16, 4, 170
    u = np.linalg.qr(rng.normal(size=(dim, 2)))[0]
    sigma = np.eye(dim) + u @ np.diag([40.0, 15.0]) @ u.T
    p = np.linalg.inv(sigma)
    logp = N.LogpSpec.mvn_precision((p + p.T) / 2)
    s = N.LowRankMclmcSettings(num_chains=n, seed=12, num_tune=120, trajectory_kind=kind, step_size=0.4)
    rec = []
    cb_o = LR.estimator_callback(rec)
    cb_e = C.cast(cb_o, _lib.LOWRANK_ESTIMATOR_FN)
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    b = N.ChainBatch(s, logp, n)
    assert (b.set_position(x0) == 0).all()
    b.set_lowrank_estimator(cb_e, n_threads=1)
    pos_a, st_a = b.draw_many(60)
    pos_b, st_b = b.draw_many(draws - 60)
    pos_g, st_g = np.concatenate([pos_a, pos_b]), np.concatenate([st_a, st_b])
    tpc = b.threads_per_chain()
    b.close()
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, dim, logp.params, oracle.gpu_cfg(tpc), n, x0, draws, estimator=cb_o)
    assert failed == 0
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
    for f in ("energy_change", "average_step_size"):
        assert ((st_g[f] == st_o[f]) | (np.isnan(st_g[f]) & np.isnan(st_o[f]))).all(), f
    assert (st_g["num_eigenvalues"] == st_o["num_eigenvalues"]).all() and st_g["num_eigenvalues"].max() >= 1
