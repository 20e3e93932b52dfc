"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on the same seeds — draw for draw,
bit for bit (positions and every sampler statistic).  north_star's tolerance is 1e-9 relative; the engine is
built to do better: identical arithmetic contract => identical bits => identical discrete decisions."""
import os

import numpy as np
import pytest

import nuts_rs_amd as N
from helpers import assert_bit_exact, assert_vectors_bit_exact, oracle_settings, run_engine, run_oracle

pytestmark = pytest.mark.gpu


def test_k1_readme_config_bit_exact(oracle):
    """BASELINE configs[0]: 10-dim iid N(3,1), 4 chains, DiagNutsSettings::default(), x0 = zeros (README.md:44-79)."""
    s = N.DiagNutsSettings(num_chains=4, seed=0)
    logp = N.LogpSpec.iid_normal(10, 3.0)
    x0 = np.zeros((4, 10))
    n = s.num_tune + s.num_draws
    pos_g, st_g, ex = run_engine(s, logp, 4, x0, n)
    pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, 4, x0, n)
    assert failed == 0 and (ex["status"] == 0).all()
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
    assert ex["counters"]["total_leapfrogs"] == steps
    # behavioural envelope of the reference's own tests (src/adapt_strategy.rs:367-435): converged, no divergences
    assert st_g["diverging"].sum() == 0
    assert abs(pos_g[s.num_tune:].mean() - 3.0) < 0.1


def test_north_star_tolerance_against_reference_arithmetic(oracle):
    """north_star: "results match the reference CpuMath path draw-for-draw on fixed seeds within 1e-9 relative f64".  Every other
    GPU test compares with oracle.gpu_cfg (the engine's own exp / ln and summation order, bit for bit); THIS one compares the engine
    with oracle.ref_cfg() = the reference's arithmetic (libm exp / ln, pulp-order SIMD sums, src/math/util.rs), directly:
      K1 (README config: 10-dim iid N(3,1), 4 chains, DiagNutsSettings::default(), x0 = 0) and chains 0-7 of K2 (dim 1024 of the
      4096-chain job, bench.py's seed): identical depth / n_steps on EVERY draw of the run, positions within 1e-9 relative
      (max |dx| / max |x| per draw) for the first N draws — N = 100 for K1, 30 for K2, chosen below the earliest departures measured over
      seeds and densities (profiles/r02_arithmetic_bridge.json: 29-30 at dim 1024, later at dim 10) — and what ends the band, where it
      ends, is DRIFT (the rounding difference of two summation orders growing through the chain's own dynamics: the first
      out-of-band draw is off by < 1e-8, with the same tree), not a flipped discrete decision.  The reference shows the same spread
      between its own SIMD widths (tests/test_oracle_golden.py::test_arithmetic_bridge_is_no_wider_than_the_references_own_simd_spread)."""
    TOL = 1e-9
    cases = [("k1", N.DiagNutsSettings(num_chains=4, seed=0), N.LogpSpec.iid_normal(10, 3.0), np.zeros((4, 10)), 4, 600, 100),
             ("k2", N.DiagNutsSettings(num_chains=4096, seed=20260928, num_tune=400), N.LogpSpec.iid_normal(1024, 3.0), None, 8, 120, 30)]
    for name, s, logp, x0, n, draws, band in cases:
        if x0 is None:
            x0 = oracle.init_positions_uniform(s.seed, 0, n, logp.dim)
        pos_g, st_g, ex = run_engine(s, logp, n, x0, draws)
        assert (ex["status"] == 0).all()
        pos_r, st_r, _, failed = run_oracle(oracle, s, logp, n, x0, draws, cfg=oracle.ref_cfg())
        assert failed == 0
        assert (st_g["depth"] == st_r["depth"]).all() and (st_g["n_steps"] == st_r["n_steps"]).all(), name      # no flipped decision anywhere
        assert (st_g["diverging"] == st_r["diverging"]).all()
        rel = np.abs(pos_g - pos_r).max(axis=2) / np.abs(pos_r).max(axis=2)            # [draw][chain]
        assert rel[:band].max() <= TOL, (name, float(rel[:band].max()))
        assert np.abs(st_g["step_size"][:band] - st_r["step_size"][:band]).max() <= TOL * st_r["step_size"][:band].max()
        for c in range(n):
            out = np.nonzero(rel[:, c] > TOL)[0]
            if len(out):                                   # the band ends by drift: just over the line, with the same tree
                assert out[0] >= band and rel[out[0], c] < 1e-8, (name, c, int(out[0]), float(rel[out[0], c]))
        assert rel.max() < 1e-3, name                      # and stays a drift for the whole run (measured 4e-6 in K2's warm-up; a parted trajectory is O(1))


def _diag_settings(**kw):
    return N.DiagNutsSettings(**kw)


PARITY_CASES = [
    # (id, settings kwargs, dim, n_chains, n_draws, density, dims_per_lane)
    ("dim100_defaults", dict(seed=1, num_tune=120, num_draws=60), 100, 6, 180, "iid", 0),
    ("dim64_edge", dict(seed=2, num_tune=60), 64, 3, 90, "iid", 0),
    ("dim65_pad", dict(seed=3, num_tune=60), 65, 3, 90, "iid", 0),
    ("dim1_scalar", dict(seed=4, num_tune=60), 1, 5, 120, "iid", 0),
    ("dim130_dpl4", dict(seed=5, num_tune=60), 130, 3, 90, "iid", 4),
    ("dim10_dpl16", dict(seed=6, num_tune=60), 10, 3, 90, "iid", 16),
    ("dim300_dpl8", dict(seed=7, num_tune=50), 300, 3, 70, "iid", 8),
    ("dim1024_dpl16", dict(seed=8, num_tune=40), 1024, 4, 55, "iid", 16),
    ("maxdepth3_readme", dict(seed=9, num_tune=100, maxdepth=3), 10, 4, 150, "iid", 0),
    ("mindepth2", dict(seed=10, num_tune=50, mindepth=2), 20, 4, 80, "iid", 0),
    ("extra_doublings", dict(seed=11, num_tune=50, extra_doublings=2, maxdepth=6), 20, 4, 80, "iid", 0),
    # extra_doublings > 2: sub-trees of level > maxdepth are built; the scratch layout must follow maxdepth + extra_doublings
    ("extra_doublings4_md3", dict(seed=31, num_tune=60, extra_doublings=4, maxdepth=3), 20, 4, 100, "iid", 0),
    ("extra_doublings3_md2_small", dict(seed=32, num_tune=60, extra_doublings=3, maxdepth=2), 6, 5, 100, "diag", 0),
    ("extra_doublings5_md5_w2", dict(seed=33, num_tune=40, extra_doublings=5, maxdepth=5), 700, 2, 60, "iid", (8, 2)),
    ("no_check_turning", dict(seed=12, num_tune=30, check_turning=False, maxdepth=4), 12, 3, 50, "iid", 0),
    ("target_time", dict(seed=13, num_tune=50, target_integration_time=2.0), 20, 4, 80, "iid", 0),
    ("ragged_scales", dict(seed=14, num_tune=150), 40, 8, 220, "diag", 0),
    ("low_energy_threshold", dict(seed=15, num_tune=50, max_energy_error=0.3), 30, 6, 80, "iid", 0),
    ("funnel_k3_dim101", dict(seed=16, num_tune=150), 101, 12, 230, "funnel", 0),
    ("funnel_dim11", dict(seed=17, num_tune=100), 11, 10, 200, "funnel", 0),
    ("eight_schools_k4", dict(seed=18, num_tune=200), 10, 16, 320, "schools", 0),
    # several wavefronts per chain (cross-wave sums through LDS): (dims_per_lane, waves_per_chain)
    ("dim1024_w2_dpl8", dict(seed=19, num_tune=40), 1024, 3, 55, "iid", (8, 2)),
    ("dim1024_w4_dpl4", dict(seed=20, num_tune=40), 1024, 3, 55, "iid", (4, 4)),
    ("dim700_w2_dpl8_pad", dict(seed=23, num_tune=40), 700, 3, 55, "diag", (8, 2)),
    ("dim2000_w2_dpl16", dict(seed=24, num_tune=30), 2000, 2, 40, "iid", (16, 2)),
    ("dim4096_w4_dpl16", dict(seed=25, num_tune=25), 4096, 2, 32, "iid", (16, 4)),
    ("funnel_dim300_w4", dict(seed=26, num_tune=60), 300, 4, 90, "funnel", (4, 4)),
    # full precision matrix (BASELINE config K5's density): per-chain GEMV, position published through LDS
    ("mvn_dim64", dict(seed=27, num_tune=80), 64, 4, 130, "mvn", 0),
    ("mvn_dim100_pad", dict(seed=28, num_tune=80), 100, 3, 120, "mvn", 0),
    ("mvn_k5_dim256", dict(seed=29, num_tune=60), 256, 3, 90, "mvn", 0),
    ("mvn_dim300_w2", dict(seed=30, num_tune=40), 300, 2, 60, "mvn", (8, 2)),
]


@pytest.mark.parametrize("case", PARITY_CASES, ids=[c[0] for c in PARITY_CASES])
def test_chain_parity_bit_exact(oracle, case):
    """Draw-for-draw bit parity across register tilings, padding edges, tree options and ragged depths."""
    _, kw, dim, n_chains, n_draws, dens, dpl = case
    s = _diag_settings(num_chains=n_chains, **kw)
    rng = np.random.default_rng(kw["seed"])
    if dens == "iid":
        logp = N.LogpSpec.iid_normal(dim, 3.0)
    elif dens == "funnel":
        logp = N.LogpSpec.funnel(dim)
    elif dens == "schools":
        logp = N.LogpSpec.eight_schools()
    elif dens == "mvn":
        a = rng.normal(size=(dim, dim))
        p = a @ a.T / dim + np.eye(dim)
        logp = N.LogpSpec.mvn_precision((p + p.T) / 2)
    else:
        logp = N.LogpSpec.diag_normal(np.exp(rng.uniform(-6, 6, dim)))        # scales e^-3 .. e^3: deep, ragged trees
    x0 = oracle.init_positions_uniform(s.seed, 0, n_chains, dim)
    dpl, wpc = dpl if isinstance(dpl, tuple) else (dpl, 0)
    pos_g, st_g, ex = run_engine(s, logp, n_chains, x0, n_draws, dims_per_lane=dpl, waves_per_chain=wpc)
    if wpc:
        assert ex["threads_per_chain"] == 64 * wpc and ex["dims_per_lane"] == dpl
    # the reduction order over dim is a function of the threads that share a chain; the oracle reproduces it
    pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, n_chains, x0, n_draws, gpu_threads=ex["threads_per_chain"])
    assert failed == 0 and (ex["status"] == 0).all()
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
    assert ex["counters"]["total_leapfrogs"] == steps
    if case[0] == "low_energy_threshold":
        assert st_g["diverging"].sum() > 0          # the divergence path was exercised
    if dens == "funnel":
        # BASELINE config K3's purpose: divergences and ragged tree depths (chains stop at different depths)
        assert st_g["diverging"].sum() > 0 and len(np.unique(st_g["depth"])) >= 5
    if case[0] == "ragged_scales":
        assert len(np.unique(st_g["depth"])) >= 4   # ragged depths across chains/draws


def test_step_size_options_parity(oracle):
    """Fixed step size, jitter None (stepsize/adapt.rs:27, :259-266) and the Adam adaptor (stepsize/adam.rs:42-112)."""
    for st_kw in (dict(method=N.STEP_FIXED, fixed_step_size=0.4), dict(jitter=None), dict(target_accept=0.95),
                  dict(method=N.STEP_ADAM), dict(method=N.STEP_ADAM, jitter=None,
                                                 adam=N.AdamOptions(beta1=0.8, beta2=0.99, learning_rate=0.1))):
        a = N.EuclideanAdaptOptions(step_size_settings=N.StepSizeSettings(**st_kw))
        s = N.DiagNutsSettings(num_chains=3, seed=21, num_tune=60, adapt_options=a)
        logp = N.LogpSpec.iid_normal(16, 3.0)
        x0 = oracle.init_positions_uniform(21, 0, 3, 16)
        pos_g, st_g, ex = run_engine(s, logp, 3, x0, 90)
        pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, 3, x0, 90)
        assert failed == 0
        assert_bit_exact(pos_g, st_g, pos_o, st_o)


def test_draw_variance_estimator_parity(oracle):
    """use_grad_based_estimate = false (update_diag_draw, transform/diagonal.rs:85-105)."""
    a = N.EuclideanAdaptOptions(mass_matrix_options=N.DiagAdaptExpSettings(use_grad_based_estimate=False))
    s = N.DiagNutsSettings(num_chains=3, seed=22, num_tune=100, adapt_options=a)
    logp = N.LogpSpec.diag_normal(np.array([1.0, 100.0, 0.01, 4.0, 1.0, 9.0]))
    x0 = oracle.init_positions_uniform(22, 0, 3, 6)
    pos_g, st_g, ex = run_engine(s, logp, 3, x0, 130)
    pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, 3, x0, 130)
    assert failed == 0
    assert_bit_exact(pos_g, st_g, pos_o, st_o)


def test_chain_sharding_is_invisible(oracle):
    """Chains are independent units: chain 5 gives the same draws whether it is local chain 5 of one
    engine or local chain 1 of an engine that starts at global id 4 — the multi-GPU partition cannot change results."""
    s = N.DiagNutsSettings(num_chains=8, seed=31, num_tune=40)
    logp = N.LogpSpec.iid_normal(24, 3.0)
    x0 = oracle.init_positions_uniform(31, 0, 8, 24)
    pos_a, st_a, _ = run_engine(s, logp, 8, x0, 60)
    pos_b, st_b, _ = run_engine(s, logp, 4, x0[4:], 60, chain_id_offset=4)
    assert (pos_a[:, 4:].view(np.uint64) == pos_b.view(np.uint64)).all()
    assert (st_a["chain"][:, 4:] == st_b["chain"]).all() and (st_a["n_steps"][:, 4:] == st_b["n_steps"]).all()


def test_incremental_draw_calls_equal_one_call(oracle):
    """nm_engine_draw(n) then (m) == nm_engine_draw(n+m): all chain state lives on the device between calls."""
    s = N.DiagNutsSettings(num_chains=4, seed=32, num_tune=30)
    logp = N.LogpSpec.iid_normal(12, 3.0)
    x0 = oracle.init_positions_uniform(32, 0, 4, 12)
    b = N.ChainBatch(s, logp, 4)
    b.set_position(x0)
    parts = [b.draw_many(k)[0] for k in (1, 7, 20, 22)]
    p2, progress = b.draw()
    assert progress[0].draw == 50 and not progress[0].tuning and progress[2].chain == 2
    b.close()
    pos, _, _ = run_engine(s, logp, 4, x0, 51)
    assert (np.concatenate(parts + [p2[None]]).view(np.uint64) == pos.view(np.uint64)).all()


def test_bad_init_reported_per_chain(oracle):
    """BadInitGrad (src/nuts.rs:21): a chain started exactly at the mode has a zero gradient."""
    s = N.DiagNutsSettings(num_chains=3, seed=33)
    logp = N.LogpSpec.iid_normal(8, 3.0)
    x0 = oracle.init_positions_uniform(33, 0, 3, 8)
    x0[1] = 3.0
    b = N.ChainBatch(s, logp, 3)
    status = b.set_position(x0, raise_on_error=False)
    assert list(status) == [0, 1, 0]
    with pytest.raises(N.NutsAmdError) as e:
        b.set_position(x0)
    assert e.value.status == 5
    x0[1, 0] = np.nan
    assert list(b.set_position(x0, raise_on_error=False)) == [0, 1, 0]
    b.close()


def test_k2_full_size_properties():
    """BASELINE configs[1] at full size (4096 chains x dim 1024): size-independent properties instead of the oracle —
    stationarity of N(3, I) (mean, variance, per-chain energy), warm-up reaching the target acceptance, no
    post-warm-up divergences, and invariance of every chain to the batch it runs in."""
    C_, D = 4096, 1024
    s = N.DiagNutsSettings(num_chains=C_, seed=20260928, num_tune=400, num_draws=50)
    b = N.ChainBatch(s, N.LogpSpec.iid_normal(D, 3.0), C_)
    x0 = b.init_positions_uniform()
    assert (b.set_position(x0) == 0).all()
    b.draw_device(400)
    pos, st = b.draw_many(50)
    assert st["diverging"].sum() == 0 and (st["tuning"] == 0).all() and (st["chain_status"] == 0).all()
    assert abs(pos.mean() - 3.0) < 2e-3 and abs(pos.var() - 1.0) < 5e-3
    assert np.abs(pos.mean(axis=(0, 2)) - 3.0).max() < 0.05            # every chain is centred
    assert abs(st["mean_tree_accept"].mean() - 0.8) < 0.05
    sd, mu = b.mass_matrix()
    assert abs(np.median(sd) - 1.0) < 0.1 and abs(np.median(mu) - 3.0) < 0.1
    # logp of a stationary N(3, I_D) draw is -chi2_D / 2
    assert abs(st["logp"].mean() + D / 2) < 1.0
    sub_pos = pos[:, 100:104]
    b.close()
    s2 = N.DiagNutsSettings(num_chains=4, seed=20260928, num_tune=400, num_draws=50)
    b2 = N.ChainBatch(s2, N.LogpSpec.iid_normal(D, 3.0), 4, chain_id_offset=100)
    b2.set_position(x0[100:104])
    b2.draw_device(400)
    pos2, _ = b2.draw_many(50)
    b2.close()
    assert (sub_pos.view(np.uint64) == pos2.view(np.uint64)).all()


def test_k3_funnel_full_size_properties():
    """BASELINE configs[2]: Neal's funnel dim 101 x 8192 chains — the ragged/divergent stress at full size.
    Properties: every chain survives, depths are ragged, divergences happen, v stays in a sane range and the
    conditional scale of x follows e^{v/2} (sign + rough magnitude), independent of the oracle."""
    C_, D = 8192, 101
    s = N.DiagNutsSettings(num_chains=C_, seed=3, num_tune=200, num_draws=50)
    b = N.ChainBatch(s, N.LogpSpec.funnel(D), C_)
    assert (b.set_position(b.init_positions_uniform()) == 0).all()
    b.draw_device(200)
    pos, st = b.draw_many(50)
    c = b.counters()
    b.close()
    assert (st["chain_status"] == 0).all() and np.isfinite(pos).all()
    depth_hist = np.bincount(st["depth"].ravel().astype(int), minlength=11)
    assert (depth_hist > 0).sum() >= 5 and st["diverging"].mean() > 1e-4
    v = pos[..., 0]
    assert -9 < v.mean() < 3 and 1.0 < v.std() < 4.0
    hi, lo = v > 1.0, v < -1.0
    assert np.abs(pos[..., 1:][hi]).mean() > 3 * np.abs(pos[..., 1:][lo]).mean()
    # lane-utilisation figure of merit of a lockstep design; this engine does not lock-step
    util = st["n_steps"].sum() / (st["n_steps"].max(axis=1).sum() * C_)
    assert 0 < util <= 1.0


def test_k4_eight_schools_posterior():
    """BASELINE configs[3] (one GPU's share: 8192 chains x dim 10): posterior summaries of the classic 8-schools
    model agree with the known values (mu ~ 4.4 +- 3.3, tau median ~ 2.7) — independent of the oracle."""
    C_ = 8192
    s = N.DiagNutsSettings(num_chains=C_, seed=4, num_tune=300, num_draws=40)
    b = N.ChainBatch(s, N.LogpSpec.eight_schools(), C_)
    assert (b.set_position(b.init_positions_uniform()) == 0).all()
    b.draw_device(300)
    pos, st = b.draw_many(40)
    assert b.group_launches() == 2          # more chains than resident wavefronts: drawn 8 per wavefront (nuts_group.hpp)
    b.close()
    assert (st["chain_status"] == 0).all()
    mu, tau = pos[..., 0], np.exp(pos[..., 1])
    assert abs(mu.mean() - 4.4) < 0.3 and abs(mu.std() - 3.3) < 0.3
    assert 2.0 < np.median(tau) < 3.6
    assert abs(pos[..., 2:].mean()) < 0.15 and abs(pos[..., 2:].std() - 1.0) < 0.1      # theta~ close to N(0,1)
    assert st["diverging"].mean() < 0.02


VECTOR_CASES = [
    # (id, settings kwargs, density, dim, n_chains, n_draws, (dims_per_lane, waves_per_chain))
    ("funnel_divergences", dict(seed=31, num_tune=80), "funnel", 31, 12, 160, (0, 0)),
    ("low_threshold_iid", dict(seed=32, num_tune=40, max_energy_error=0.25), "iid", 70, 6, 70, (0, 0)),
    ("schools", dict(seed=33, num_tune=120), "schools", 10, 8, 200, (0, 0)),
    ("iid_w2", dict(seed=34, num_tune=40, max_energy_error=0.5), "iid", 600, 3, 55, (8, 2)),
    # several chains per wavefront (nuts_group.hpp): 8 / 4 / 2 chains, the last wavefront partly filled
    ("grouped_funnel_dim11", dict(seed=35, num_tune=80), "funnel", 11, 21, 160, "group"),
    ("grouped_schools", dict(seed=36, num_tune=120), "schools", 10, 13, 200, "group"),
    ("grouped_low_threshold_dim30", dict(seed=37, num_tune=40, max_energy_error=0.25), "iid", 30, 9, 70, "group"),
    ("grouped_funnel_dim50", dict(seed=38, num_tune=80), "funnel", 50, 5, 160, "group"),
]


@pytest.mark.parametrize("case", VECTOR_CASES, ids=[c[0] for c in VECTOR_CASES])
def test_expanded_draw_vector_statistics_bit_exact(oracle, case):
    """`expanded_draw` (src/chain.rs:190-204): gradient / transformed point / mass-matrix events / divergence
    locations, against the oracle's copies of the same reference fields."""
    _, kw, dens, dim, n_chains, n_draws, tiling = case
    grouped = tiling == "group"
    dpl, wpc = (0, 0) if grouped else tiling
    s = N.DiagNutsSettings(num_chains=n_chains, store_gradient=True, store_unconstrained=True, store_transformed=True,
                           store_divergences=True, **kw)
    s.adapt_options.mass_matrix_options.store_mass_matrix = True
    logp = {"funnel": lambda: N.LogpSpec.funnel(dim), "iid": lambda: N.LogpSpec.iid_normal(dim, 3.0),
            "schools": N.LogpSpec.eight_schools}[dens]()
    x0 = oracle.init_positions_uniform(s.seed, 0, n_chains, dim)
    b = N.ChainBatch(s, logp, n_chains, dims_per_lane=dpl, waves_per_chain=wpc, lane_groups=2 if grouped else 1)
    assert sorted(b.stored_vectors()) == sorted(k for k in N.VECTOR_STATS if k != "mass_matrix_eigvals")   # (LowRankSettings only)
    b.set_position(x0)
    pos_g, st_g, vec_g = b.expanded_draw_many(n_draws)
    tpc = b.threads_per_chain()
    assert b.group_launches() == (1 if grouped else 0)
    b.close()
    vec_o = {}
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, logp.dim, logp.params, oracle.gpu_cfg(tpc),
                                        n_chains, x0, n_draws, n_threads=8, vectors=vec_o)
    assert failed == 0
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
    assert_vectors_bit_exact(vec_g, vec_o)
    # the events really occur in these cases, and only their rows are written
    div = st_g["diverging"] != 0
    upd = st_g["transformation_update_id"] >= 0
    assert div.sum() > 0 and upd.sum() > n_chains
    assert (np.isnan(vec_g["divergence_start"]).all(axis=2) == ~div).all()
    assert (np.isnan(vec_g["mass_matrix_inv"]).all(axis=2) == ~upd).all()
    assert upd[0].all() and not upd[-1].any()                    # first extraction compares with -1; none after tuning


def test_multi_wave_tilings_seed_sweep(oracle):
    """Chains that span 2 or 4 wavefronts (cross-wave sums through LDS, shared RNG cache, barriers): a sweep over
    seeds, dims with and without padding and both densities, each draw for draw against the oracle."""
    rng = np.random.default_rng(99)
    cases = []
    for i in range(14):
        wpc = 2 if i % 2 == 0 else 4
        dpl = int(rng.choice([8, 16] if wpc == 2 else [4, 16]))
        cap = 64 * wpc * dpl
        dim = int(rng.integers(cap // 2 + 1, cap + 1)) if i % 3 else cap
        cases.append((dpl, wpc, dim, 1000 + i, "diag" if i % 4 == 1 else "iid"))
    for dpl, wpc, dim, seed, dens in cases:
        s = N.DiagNutsSettings(num_chains=2, seed=seed, num_tune=25)
        logp = (N.LogpSpec.iid_normal(dim, 3.0) if dens == "iid"
                else N.LogpSpec.diag_normal(np.exp(np.random.default_rng(seed).uniform(-4, 4, dim))))
        x0 = oracle.init_positions_uniform(seed, 0, 2, dim)
        pos_g, st_g, ex = run_engine(s, logp, 2, x0, 36, dims_per_lane=dpl, waves_per_chain=wpc)
        assert ex["threads_per_chain"] == 64 * wpc
        pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, 2, x0, 36, gpu_threads=64 * wpc)
        assert failed == 0
        try:
            assert_bit_exact(pos_g, st_g, pos_o, st_o)
        except AssertionError as e:
            raise AssertionError(f"tiling (dpl {dpl}, waves {wpc}) dim {dim} seed {seed} {dens}: {e}") from None


def test_random_settings_sweep(oracle):
    """Randomised sweep over the sampler's option space (tree limits, U-turn switches, extra doublings, target
    integration time, energy threshold, step-size method / jitter, estimator kind, window schedule) on small
    problems: every combination must agree with the oracle draw for draw."""
    rng = np.random.default_rng(2024)
    for i in range(int(os.environ.get("NM_SWEEP_CASES", "80"))):
        maxdepth = int(rng.integers(1, 8))
        st = N.StepSizeSettings(
            target_accept=float(rng.choice([0.6, 0.8, 0.9])), initial_step=float(rng.choice([0.01, 0.1, 1.0])),
            jitter=None if rng.random() < 0.3 else float(rng.choice([0.05, 0.1, 0.3])),
            method=int(rng.choice([N.STEP_DUAL_AVERAGE, N.STEP_DUAL_AVERAGE, N.STEP_ADAM, N.STEP_FIXED])),
            fixed_step_size=float(rng.choice([0.2, 0.7])))
        a = N.EuclideanAdaptOptions(
            step_size_settings=st,
            mass_matrix_options=N.DiagAdaptExpSettings(use_grad_based_estimate=bool(rng.random() < 0.7)),
            early_window=float(rng.choice([0.1, 0.3, 0.5])), step_size_window=float(rng.choice([0.1, 0.15, 0.3])),
            mass_matrix_switch_freq=int(rng.choice([10, 30, 80])), early_mass_matrix_switch_freq=int(rng.choice([5, 10])),
            mass_matrix_update_freq=int(rng.choice([1, 1, 3])), mass_matrix_window_growth=float(rng.choice([1.0, 1.5, 2.0])))
        kw = dict(seed=int(rng.integers(0, 2 ** 31)), num_tune=int(rng.integers(20, 90)), maxdepth=maxdepth,
                  mindepth=int(rng.integers(0, maxdepth + 1)) if rng.random() < 0.3 else 0,
                  check_turning=bool(rng.random() < 0.85), extra_doublings=int(rng.integers(0, 3)) if rng.random() < 0.3 else 0,
                  max_energy_error=float(rng.choice([1000.0, 1000.0, 2.0, 0.3])),
                  target_integration_time=None if rng.random() < 0.75 else float(rng.choice([0.5, 2.0, 8.0])),
                  adapt_options=a)
        dens = rng.choice(["iid", "diag", "funnel", "mvn", "schools"], p=[0.3, 0.3, 0.2, 0.1, 0.1])
        dim = 10 if dens == "schools" else int(rng.integers(2, 40)) if dens == "mvn" else int(rng.integers(2, 150))
        n_chains = int(rng.integers(1, 5))
        s = N.DiagNutsSettings(num_chains=n_chains, **kw)

        def mvn():
            a_ = np.random.default_rng(i).normal(size=(dim, dim))
            p_ = a_ @ a_.T / dim + np.eye(dim)
            return N.LogpSpec.mvn_precision((p_ + p_.T) / 2)
        logp = {"iid": lambda: N.LogpSpec.iid_normal(dim, 3.0), "funnel": lambda: N.LogpSpec.funnel(dim), "mvn": mvn,
                "schools": N.LogpSpec.eight_schools,
                "diag": lambda: N.LogpSpec.diag_normal(np.exp(np.random.default_rng(i).uniform(-3, 3, dim)))}[dens]()
        # now and then a wider tiling than the dim needs (more doubles per lane, or several waves per chain)
        dpl, wpc = 0, 0
        if dens != "schools" and rng.random() < 0.3:
            dpl, wpc = [(4, 1), (8, 1), (16, 1), (8, 2), (4, 4)][int(rng.integers(0, 5))]
        x0 = oracle.init_positions_uniform(s.seed, 0, n_chains, dim)
        n_draws = s.num_tune + 25
        pos_g, st_g, ex = run_engine(s, logp, n_chains, x0, n_draws, dims_per_lane=dpl, waves_per_chain=wpc)
        pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, n_chains, x0, n_draws, gpu_threads=ex["threads_per_chain"])
        if failed or not (ex["status"] == 0).all():
            assert failed == int((ex["status"] != 0).sum()), f"case {i}: init failures differ"
            continue
        try:
            assert_bit_exact(pos_g, st_g, pos_o, st_o)
        except AssertionError as e:
            raise AssertionError(f"case {i} ({dens}, dim {dim}, {kw}): {e}") from None


def test_more_chains_than_resident_blocks(oracle):
    """Blocks stride over the chains (chain = block, block + grid, ...): tree scratch belongs to the block, state to the
    chain.  A grid of 2 or 3 blocks serving 7 chains must give what 7 resident blocks give."""
    s = N.DiagNutsSettings(num_chains=7, seed=41, num_tune=50)
    for logp, wpc, dpl in ((N.LogpSpec.funnel(20), 0, 0), (N.LogpSpec.iid_normal(300, 3.0), 2, 8)):
        x0 = oracle.init_positions_uniform(41, 0, 7, logp.dim)
        pos_o, st_o, _, failed = run_oracle(oracle, s, logp, 7, x0, 80, gpu_threads=64 * (wpc or 1))
        assert failed == 0
        for grid in (2, 3):
            b = N.ChainBatch(s, logp, 7, grid_blocks=grid, waves_per_chain=wpc, dims_per_lane=dpl)
            b.set_position(x0)
            pos_a, st_a = b.draw_many(30)          # two launches: the state of a non-resident chain survives in HBM
            pos_b, st_b = b.draw_many(50)
            b.close()
            assert_bit_exact(np.concatenate([pos_a, pos_b]), np.concatenate([st_a, st_b]), pos_o, st_o)


def test_lane_group_kernel_sweep(oracle):
    """Chains with dim <= 16 / 32 / 64 can be drawn 8 / 4 / 2 per wavefront (nuts_group.hpp): randomised settings, ragged chain counts
    (partial wavefronts, more chains than resident groups), launches cut at and after the end of the warm-up — the same
    bits as the oracle, which knows nothing of the grouping."""
    rng = np.random.default_rng(77)
    for i in range(int(os.environ.get("NM_GROUP_SWEEP_CASES", "40"))):
        maxdepth = int(rng.integers(1, 9))
        st = N.StepSizeSettings(
            target_accept=float(rng.choice([0.6, 0.8, 0.9])), initial_step=float(rng.choice([0.01, 0.1, 1.0])),
            jitter=None if rng.random() < 0.3 else float(rng.choice([0.05, 0.1, 0.3])),
            method=int(rng.choice([N.STEP_DUAL_AVERAGE, N.STEP_DUAL_AVERAGE, N.STEP_ADAM, N.STEP_FIXED])),
            fixed_step_size=float(rng.choice([0.2, 0.7])))
        kw = dict(seed=int(rng.integers(0, 2 ** 31)), num_tune=int(rng.integers(20, 70)), maxdepth=maxdepth,
                  mindepth=int(rng.integers(0, maxdepth + 1)) if rng.random() < 0.3 else 0,
                  check_turning=bool(rng.random() < 0.85), extra_doublings=int(rng.integers(0, 3)) if rng.random() < 0.3 else 0,
                  max_energy_error=float(rng.choice([1000.0, 1000.0, 2.0, 0.3])),
                  target_integration_time=None if rng.random() < 0.75 else float(rng.choice([0.5, 2.0, 8.0])),
                  adapt_options=N.EuclideanAdaptOptions(
                      step_size_settings=st,
                      mass_matrix_options=N.DiagAdaptExpSettings(use_grad_based_estimate=bool(rng.random() < 0.7)),
                      early_window=float(rng.choice([0.1, 0.3, 0.5])), step_size_window=float(rng.choice([0.1, 0.15, 0.3])),
                      mass_matrix_switch_freq=int(rng.choice([10, 30, 80])), early_mass_matrix_switch_freq=int(rng.choice([5, 10])),
                      mass_matrix_update_freq=int(rng.choice([1, 1, 3])), mass_matrix_window_growth=float(rng.choice([1.0, 1.5, 2.0]))))
        dens = rng.choice(["iid", "diag", "schools", "funnel", "mvn"], p=[0.25, 0.3, 0.15, 0.15, 0.15])
        dim = 10 if dens == "schools" else int(rng.integers(1, 17)) if rng.random() < 0.5 else int(rng.integers(17, 65))
        if dens == "funnel":
            dim = max(dim, 2)
        n_chains = int(rng.integers(1, 40))
        s = N.DiagNutsSettings(num_chains=n_chains, **kw)

        def mvn():
            a_ = np.random.default_rng(i).normal(size=(dim, dim))
            p_ = a_ @ a_.T / dim + np.eye(dim)
            return N.LogpSpec.mvn_precision((p_ + p_.T) / 2)
        logp = {"iid": lambda: N.LogpSpec.iid_normal(dim, 3.0), "schools": N.LogpSpec.eight_schools,
                "funnel": lambda: N.LogpSpec.funnel(dim), "mvn": mvn,
                "diag": lambda: N.LogpSpec.diag_normal(np.exp(np.random.default_rng(i).uniform(-3, 3, dim)))}[dens]()
        x0 = oracle.init_positions_uniform(s.seed, 0, n_chains, dim)
        n_draws = s.num_tune + 40
        grid = int(rng.integers(1, 4)) if rng.random() < 0.3 else 0
        pos_g, st_g, ex = run_engine(s, logp, n_chains, x0, n_draws, lane_groups=2, grid_blocks=grid,
                                     splits=(s.num_tune, s.num_tune + 1, s.num_tune + 17))
        pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, n_chains, x0, n_draws, gpu_threads=64)
        if failed or not (ex["status"] == 0).all():
            assert failed == int((ex["status"] != 0).sum()), f"case {i}: init failures differ"
            continue
        try:
            assert_bit_exact(pos_g, st_g, pos_o, st_o)
            assert ex["counters"]["total_leapfrogs"] == steps
            assert ex["group_launches"] == 4
        except AssertionError as e:
            raise AssertionError(f"case {i} ({dens}, dim {dim}, {n_chains} chains, grid {grid}, {kw}): {e}") from None


def test_lane_group_kernel_is_the_default_for_many_small_chains(oracle):
    """Automatic choice (more chains than resident wavefronts): same draws as with the grouping switched off, and the
    final state the host reads back (positions, gradients, step sizes, mass matrix) is the same too."""
    n = 6000
    s = N.DiagNutsSettings(num_chains=n, seed=5, num_tune=60)
    logp = N.LogpSpec.eight_schools()
    x0 = oracle.init_positions_uniform(s.seed, 0, n, 10)
    a = run_engine(s, logp, n, x0, 90, lane_groups=0, splits=(60,))
    b = run_engine(s, logp, n, x0, 90, lane_groups=1, splits=(60,))
    assert_bit_exact(a[0], a[1], b[0], b[1])
    assert a[2]["group_launches"] == 2 and b[2]["group_launches"] == 0
    for k in ("x", "gx", "step_sizes", "stds", "mean"):
        assert (a[2][k].view(np.uint64) == b[2][k].view(np.uint64)).all(), k
    assert a[2]["counters"]["total_leapfrogs"] == b[2]["counters"]["total_leapfrogs"]
    few = run_engine(s, logp, 100, x0[:100], 10)
    assert few[2]["group_launches"] == 0        # few chains: one wavefront per chain finishes sooner


def test_lane_group_kernel_skips_failed_chains(oracle):
    """A chain whose set_position failed (BadInitGrad) shares a wavefront with seven healthy ones: they draw what they
    draw with one wavefront per chain; the failed chain stays untouched."""
    n, dim = 19, 8
    s = N.DiagNutsSettings(num_chains=n, seed=34, num_tune=40)
    logp = N.LogpSpec.iid_normal(dim, 3.0)
    x0 = oracle.init_positions_uniform(34, 0, n, dim)
    x0[1] = 3.0
    x0[9] = 3.0
    good = np.array([c not in (1, 9) for c in range(n)])
    out = {}
    for lg in (1, 2):
        b = N.ChainBatch(s, logp, n, lane_groups=lg)
        assert list(np.nonzero(b.set_position(x0, raise_on_error=False))[0]) == [1, 9]
        with pytest.raises(N.NutsAmdError):
            b.draw_many(1)                            # a failed chain is an error, as Chain::draw's Result is
        pos, st = b.draw_many(70, raise_on_error=False)
        out[lg] = (pos[:, good], st[:, good], b.group_launches(), b.positions()[~good])
        b.close()
    assert out[1][2] == 0 and out[2][2] == 2
    assert_bit_exact(out[2][0], out[2][1], out[1][0], out[1][1])
    assert (out[2][3] == out[1][3]).all() and (out[2][1]["chain_status"] == 0).all()


def test_lane_group_kernel_late_starter(oracle):
    """A chain whose first set_position failed starts its warm-up when the others are already sampling: the launch must
    still take the kernel with the adaptation in it (the engine keys that on the chain that is furthest behind)."""
    n, dim = 12, 6
    s = N.DiagNutsSettings(num_chains=n, seed=35, num_tune=30)
    logp = N.LogpSpec.iid_normal(dim, 3.0)
    x0 = oracle.init_positions_uniform(35, 0, n, dim)
    bad = x0.copy()
    bad[4] = 3.0                                             # zero gradient: BadInitGrad
    out = {}
    for lg in (1, 2):
        b = N.ChainBatch(s, logp, n, lane_groups=lg)
        assert list(np.nonzero(b.set_position(bad, raise_on_error=False))[0]) == [4]
        b.draw_many(45, raise_on_error=False)                # the other 11 chains finish their warm-up
        cur = b.positions()
        cur[4] = x0[4]
        assert (b.set_position(cur, raise_on_error=False) == 0).all()
        out[lg] = b.draw_many(50)
        b.close()
    assert_bit_exact(out[2][0], out[2][1], out[1][0], out[1][1])
    assert (out[2][1]["tuning"][:30, 4] == 1).all() and (out[2][1]["tuning"][:, 0] == 0).all()


def test_k4_full_size_one_gpu():
    """BASELINE configs[3] at its full size on ONE GPU (65536 chains x dim 10, one chain per lane = 64 chains per wavefront): every chain
    healthy, the pooled posterior where it belongs, chains statistically exchangeable (no chain-position artefacts of
    the grouping: the 8 lanes-groups of a wavefront see the same distribution)."""
    C_ = 65536
    s = N.DiagNutsSettings(num_chains=C_, seed=5, num_tune=200, num_draws=20)
    b = N.ChainBatch(s, N.LogpSpec.eight_schools(), C_)
    assert (b.set_position(b.init_positions_uniform()) == 0).all()
    b.draw_device(200)
    pos, st = b.draw_many(20)
    assert b.lane_launches() == 2 and b.group_launches() == 1        # 65536 chains of dim 10: one chain per lane (nuts_lane.hpp) is the default, after the warm-up's first 20 draws
    b.close()
    assert (st["chain_status"] == 0).all() and (st["tuning"] == 0).all()
    mu = pos[..., 0]
    assert abs(mu.mean() - 4.4) < 0.15 and abs(mu.std() - 3.3) < 0.15
    by_group = mu.reshape(20, C_ // 8, 8).mean(axis=(0, 1))          # chains by their position inside the wavefront
    assert np.ptp(by_group) < 0.25
    assert 0.7 < st["mean_tree_accept"].mean() < 0.9 and st["diverging"].mean() < 0.02


def test_dim_zero_draws_return_the_initial_point(oracle):
    """src/nuts.rs:322-326: with dim == 0 `draw` returns the initial state and an info of depth 0 without touching the direction
    stream; the rest of the chain (momentum refresh of no elements, collectors, the adaptation fed with a mean acceptance of 0 / 0)
    runs as for any other dim.  Engine against oracle, every statistic; NaNs compare as NaNs (0 / 0 has the sign bit set on x86 and clear
    on the GPU: the one place where this repo's bit-for-bit contract cannot hold, documented in DESIGN section 4)."""
    n = 3
    s = N.DiagNutsSettings(num_chains=n, seed=12, num_tune=25)
    logp = N.LogpSpec.iid_normal(0, 3.0)
    x0 = np.zeros((n, 0))
    pos_g, st_g, ex = run_engine(s, logp, n, x0, 40, splits=(25,))
    pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, n, x0, 40, gpu_threads=64)
    assert failed == 0 and (ex["status"] == 0).all()
    assert pos_g.shape == (40, n, 0) and steps == 0 and ex["counters"]["total_leapfrogs"] == 0
    assert (st_g["depth"] == 0).all() and (st_g["n_steps"] == 0).all() and (st_g["maxdepth_reached"] == 0).all() and (st_g["diverging"] == 0).all()
    for f in st_g.dtype.names:
        a, b = st_g[f], st_o[f]
        if a.dtype.kind == "f":
            both_nan = (a != a) & (b != b)
            assert (both_nan | (a.view(np.uint64) == b.view(np.uint64))).all(), f
        else:
            assert (a == b).all(), f


def test_init_retry_loop_matches_reference_semantics(oracle):
    """The ChainProcess init loop (src/sampler.rs:1133-1147): a chain whose first initial point is rejected (BadInitGrad:
    an iid normal started exactly at its mean has a zero whitened gradient) takes its next init_position; the chains that
    started are not touched.  The oracle does the same two `set_position` calls on that chain."""
    dim, n = 20, 6
    s = N.DiagNutsSettings(num_chains=n, seed=13, num_tune=60)
    logp = N.LogpSpec.iid_normal(dim, 3.0)
    b = N.ChainBatch(s, logp, n)
    x0 = b.init_positions_uniform()
    x0[1] = 3.0
    x0[4, 7] = np.nan                                   # non-finite start: BadInitGrad as well
    status, tries = b.init_with_retries(x0)
    assert (status == 0).all() and list(tries) == [1, 2, 1, 1, 2, 1]
    L = N.load_library()
    x1 = np.empty_like(x0)
    L.nm_init_positions_uniform_at(s.seed, 0, n, dim, 0, x1.ctypes.data)     # the retry of a chain = its init_position #0
    pos, st = b.draw_many(90)
    b.close()
    so = oracle_settings(oracle, s)
    for c in range(n):
        ch = oracle.Chain(so, logp.kind, dim, logp.params, oracle.gpu_cfg(64), chain_id=c)
        rc = ch.set_position(x0[c])
        if c in (1, 4):
            assert rc == 1                               # ST_BAD_INIT
            assert ch.set_position(x1[c]) == 0
        else:
            assert rc == 0
        for t in range(90):
            p, q, rc = ch.draw()
            assert rc == 0
            assert (p.view(np.uint64) == pos[t, c].view(np.uint64)).all(), (c, t)
            assert q["n_steps"] == st["n_steps"][t, c] and q["step_size"] == st["step_size"][t, c]


@pytest.mark.parametrize("config", ["k2", "k3", "k4", "k5_diag"])
def test_baseline_configs_first_chains_match_oracle(oracle, config):
    """BASELINE.json's configurations AT THEIR OWN SIZE AND SETTINGS (4096 x 1024 iid normal; funnel 101 x 8192; 8 schools x
    65536 chains on the several-chains-per-wavefront kernel; the full-precision normal 256 x 4096 with the diagonal
    adaptation, on the matrix-core kernel): DiagNutsSettings defaults (400 tuning draws) + 50 draws, and chains 0-3, two from the
    middle and the last two of that very run (late blocks of the grid) compared with the oracle draw for draw, bit for bit —
    positions, tree sizes, step sizes, energies."""
    if config == "k2":
        logp, C_ = N.LogpSpec.iid_normal(1024, 3.0), 4096
    elif config == "k3":
        logp, C_ = N.LogpSpec.funnel(101), 8192
    elif config == "k4":
        logp, C_ = N.LogpSpec.eight_schools(), 65536
    else:
        rng = np.random.default_rng(1)
        u = np.linalg.qr(rng.normal(size=(256, 4)))[0]
        sigma = np.eye(256) + u @ np.diag([30.0, 20.0, 10.0, 5.0]) @ u.T
        p = np.linalg.inv(sigma)
        logp, C_ = N.LogpSpec.mvn_precision((p + p.T) / 2), 4096
    tune, draws = 400, 50
    s = N.DiagNutsSettings(num_chains=C_, seed=20260928, num_tune=tune, num_draws=draws)
    b = N.ChainBatch(s, logp, C_)
    x0 = b.init_positions_uniform()
    status, _ = b.init_with_retries(x0)
    assert (status == 0).all()
    sel = [0, 1, 2, 3, C_ // 2, C_ // 2 + 1, C_ - 2, C_ - 1]
    import torch
    total = tune + draws
    pos = torch.empty((total, C_, logp.dim), dtype=torch.float64, device="cuda") if C_ * logp.dim * total * 8 < 4e9 else None
    if pos is not None:
        st = torch.zeros((total, C_, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()        # the fill runs on torch's stream, the engine on its own: it must not overtake the draws
        b.draw_device(total, pos.data_ptr(), st.data_ptr())
        pos_g = pos[:, sel].cpu().numpy()
        st_g = np.frombuffer(st[:, sel].contiguous().cpu().numpy().tobytes(), dtype=N.STATS_DTYPE).reshape(total, len(sel))
    else:       # K2 / K3 traces are tens of GB: chunks of 50 draws
        parts = []
        for lo in range(0, total, 50):
            n = min(50, total - lo)
            p_ = torch.empty((n, C_, logp.dim), dtype=torch.float64, device="cuda")
            q_ = torch.zeros((n, C_, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            b.draw_device(n, p_.data_ptr(), q_.data_ptr())
            parts.append((p_[:, sel].cpu().numpy(), np.frombuffer(q_[:, sel].contiguous().cpu().numpy().tobytes(), dtype=N.STATS_DTYPE).reshape(n, len(sel))))
            del p_, q_
        pos_g, st_g = np.concatenate([a for a, _ in parts]), np.concatenate([q for _, q in parts])
    tpc = b.threads_per_chain()
    if config == "k4":
        assert b.group_launches() + b.lane_launches() > 0      # the many-chain job runs 8 (nuts_group.hpp) or 64 (nuts_lane.hpp) chains per wavefront
    if config == "k5_diag":
        assert b.tile_launches() > 0                     # P x on the matrix cores, per-chain mass matrices
    b.close()
    for lo, cnt, col in ((0, 4, 0), (C_ // 2, 2, 4), (C_ - 2, 2, 6)):
        pos_o, st_o, _, failed = run_oracle(oracle, s, logp, cnt, x0[lo:lo + cnt], total, chain_id_offset=lo, gpu_threads=tpc, n_threads=4)
        assert failed == 0
        assert_bit_exact(pos_g[:, col:col + cnt], st_g[:, col:col + cnt], pos_o, st_o)


@pytest.mark.parametrize("dens,dim,n", [("funnel", 101, 4200), ("iid", 256, 2100), ("mvn", 64, 1300)])
def test_late_blocks_match_oracle_and_are_deterministic(oracle, dens, dim, n):
    """Chains handled by blocks that start late in a big grid (index >= 1024) against the oracle, and two runs against each other.
    (These kernels cap their registers and spill; a 128-bit buffer store directly followed by a write of its data registers
    corrupted single lanes there — nuts_kernels.hpp buf_store2 — visible as chains >= 1024 whose first mass-matrix update came
    out different from run to run.)"""
    s = N.DiagNutsSettings(num_chains=n, seed=77, num_tune=400)
    if dens == "mvn":
        a = np.random.default_rng(5).normal(size=(dim, dim))
        logp = N.LogpSpec.mvn_precision((a @ a.T / dim + np.eye(dim) + (a @ a.T / dim + np.eye(dim)).T) / 2)
    else:
        logp = N.LogpSpec.funnel(dim) if dens == "funnel" else N.LogpSpec.iid_normal(dim, 3.0)
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    draws = 16
    runs = []
    for _ in range(2):
        b = N.ChainBatch(s, logp, n, lane_groups=1, chain_tiles=1)
        b.set_position(x0)
        pos, st = b.draw_many(draws)
        runs.append((pos, st, b.mass_matrix()[0], b.threads_per_chain()))
        b.close()
    assert (runs[0][0].view(np.uint64) == runs[1][0].view(np.uint64)).all()
    assert (runs[0][2].view(np.uint64) == runs[1][2].view(np.uint64)).all()
    picks = [0, 1023] + list(range(1024, 1040)) + list(range(n - 8, n))
    for c in picks:
        pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, dim, logp.params, oracle.gpu_cfg(runs[0][3]), 1,
                                            x0[c:c + 1], draws, chain_offset=c, n_threads=1)
        assert failed == 0
        assert_bit_exact(runs[0][0][:, c:c + 1], runs[0][1][:, c:c + 1], pos_o, st_o)


def test_to_host_pipeline_equals_device_buffers(oracle):
    """`nm_engine_draw_to_host` cuts a launch into chunks that cross PCIe while the next chunk's kernel runs: the same draws
    as one launch into device buffers (several chunks, a partial last one), also into reused and into pinned host arrays."""
    import torch
    from nuts_rs_amd import _lib
    dim, n, tune, draws = 1024, 4096, 12, 21                       # 32 MiB per draw of all chains: chunks of 8, 8, 5 draws
    s = N.DiagNutsSettings(num_chains=n, seed=91, num_tune=tune, maxdepth=4)
    logp = N.LogpSpec.iid_normal(dim, 3.0)
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    a = N.ChainBatch(s, logp, n)
    a.set_position(x0)
    pos_d = torch.empty((draws, n, dim), dtype=torch.float64, device="cuda")
    st_d = torch.empty((draws, n, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    a.draw_device(draws, pos_d.data_ptr(), st_d.data_ptr())
    kl_a = a.counters()["kernel_launches"]
    a.close()
    b = N.ChainBatch(s, logp, n)
    b.set_position(x0)
    pos_h, st_h = b.draw_many(draws)
    assert b.counters()["kernel_launches"] == kl_a + 2              # three chunks
    assert (pos_h.view(np.uint64) == pos_d.cpu().numpy().view(np.uint64)).all()
    assert (st_h.view(np.uint8).reshape(draws, n, -1) == st_d.cpu().numpy()).all()
    # the next draws into the same arrays (touched memory), then into registered (pinned) ones: a continuation of the chains
    c = N.ChainBatch(s, logp, n)
    c.set_position(x0)
    c.draw_many(draws)
    ref_pos, ref_st = c.draw_many(draws)
    c.close()
    L = _lib.load()
    _lib.check(L.nm_host_register(pos_h.ctypes.data, pos_h.nbytes))
    try:
        b.draw_many(draws, out=(pos_h, st_h))
    finally:
        _lib.check(L.nm_host_unregister(pos_h.ctypes.data))
    b.close()
    assert (pos_h.view(np.uint64) == ref_pos.view(np.uint64)).all() and (st_h.view(np.uint8) == ref_st.view(np.uint8)).all()


@pytest.mark.parametrize("dim", [1, 3, 127, 129, 1023, 1025, 2047, 2049, 4095, 4097, 8191, 9001])
def test_position_rows_of_odd_dims_in_every_tiling(oracle, dim):
    """The position row of a draw is written through a per-row buffer descriptor whose range ends on the last whole PAIR of elements; an
    odd last element is stored by the lane that holds it (`write_row`).  Odd dims on every tiling (2 / 4 / 8 / 16 doubles per
    lane, 1 / 2 / 4 wavefronts per chain, several blocks per chain beyond 4096), rows of neighbouring chains and draws packed without
    padding: the engine's positions and statistics are the oracle's, and the cell behind a row's last element is the next row's first."""
    n, nd = 3, 14
    s = N.DiagNutsSettings(num_chains=n, seed=100 + dim, num_tune=8, num_draws=6, maxdepth=4)
    logp = N.LogpSpec.iid_normal(dim, 3.0)
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    pos_g, st_g, ex = run_engine(s, logp, n, x0, nd)
    cfg = oracle.gpu_cfg(256, gpu_slice=4096) if dim > 4096 else oracle.gpu_cfg(ex["threads_per_chain"])      # (beyond 4096: the blocks' slices, tests/test_gpu_wide_chains.py)
    pos_o, st_o, _, failed = run_oracle(oracle, s, logp, n, x0, nd, cfg=cfg)
    assert failed == 0
    assert_bit_exact(pos_g, st_g, pos_o, st_o)
