"""One chain per LANE (nuts_rs_amd/csrc/nuts_lane.hpp; north_star's mapping): chains with dim <= 16 drawn 64 per
wavefront.  The oracle knows nothing of the mapping: positions and every statistic must agree bit for bit — randomised settings,
every built-in density the kernels cover, ragged chain counts (partial wavefronts, more chains than resident wavefronts), launches
cut at and after the end of the warm-up, vector statistics, divergences; and launches of this kernel alternate with the others'
on one engine's state."""
import os

import numpy as np
import pytest

import nuts_rs_amd as N
from helpers import assert_bit_exact, oracle_settings, run_engine, run_oracle

pytestmark = pytest.mark.gpu


def _case(rng, i):
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
    dens = rng.choice(["iid", "diag", "schools", "funnel"], p=[0.3, 0.3, 0.2, 0.2])
    dim = 10 if dens == "schools" else int(rng.integers(1, 11))
    if dens == "funnel":
        dim = max(dim, 2)
    logp = {"iid": lambda: N.LogpSpec.iid_normal(dim, 3.0), "schools": N.LogpSpec.eight_schools, "funnel": lambda: N.LogpSpec.funnel(dim),
            "diag": lambda: N.LogpSpec.diag_normal(np.exp(np.random.default_rng(i).uniform(-3, 3, dim)))}[dens]()
    return dens, dim, kw, logp


def test_lane_kernel_sweep(oracle, lane_chains=2):
    """lane_chains = 2: the 64 chains of a wavefront start every draw together.  (Round 4's second launch form, unsynchronised draws —
    lane_chains = 3 — and the 8-pair kernel for dim 11 .. 16 were measured losers and are removed: dims above 10 run on the 8-lane kernels,
    test_removed_lane_forms.)"""
    rng = np.random.default_rng(177)
    for i in range(int(os.environ.get("NM_LANE_SWEEP_CASES", "120"))):
        dens, dim, kw, logp = _case(rng, i)
        n_chains = int(rng.choice([1, 3, 17, 64, 65, 100, 200]))
        s = N.DiagNutsSettings(num_chains=n_chains, **kw)
        x0 = oracle.init_positions_uniform(s.seed, 0, n_chains, dim)
        n_draws = s.num_tune + 40
        grid = int(rng.integers(1, 3)) if rng.random() < 0.3 else 0
        pos_g, st_g, ex = run_engine(s, logp, n_chains, x0, n_draws, lane_chains=lane_chains, grid_blocks=grid,
                                     splits=(s.num_tune, s.num_tune + 1, s.num_tune + 17))
        pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, n_chains, x0, n_draws, gpu_threads=64)
        if failed or not (ex["status"] == 0).all():
            assert failed == int((ex["status"] != 0).sum()), f"case {i}: init failures differ"
            continue
        try:
            assert ex["lane_launches"] == 4 and ex["group_launches"] == 0
            assert_bit_exact(pos_g, st_g, pos_o, st_o)
            assert ex["counters"]["total_leapfrogs"] == steps
        except AssertionError as e:
            raise AssertionError(f"case {i} ({dens}, dim {dim}, {n_chains} chains, grid {grid}, {kw}): {e}") from None


@pytest.mark.parametrize("maxdepth,extra", [(3, 4), (2, 3), (5, 5), (1, 6), (7, 3)])
def test_extra_doublings_beyond_two_lane_and_group_kernels(oracle, maxdepth, extra):
    """extra_doublings >= 3 (src/nuts.rs:350-371): sub-trees of level > maxdepth are built, so every slot family must be laid
    out for maxdepth + extra_doublings levels (with the layout of maxdepth alone the L[] slots ran into the candidate
    pool and draws were silently wrong).  One chain per lane and 8 lanes per chain, against the oracle."""
    n, dim = 70, 9
    s = N.DiagNutsSettings(num_chains=n, seed=500 + 10 * maxdepth + extra, num_tune=50, maxdepth=maxdepth, extra_doublings=extra)
    logp = N.LogpSpec.diag_normal(np.exp(np.random.default_rng(maxdepth).uniform(-2, 2, dim)))
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    pos_o, st_o, steps, failed = run_oracle(oracle, s, logp, n, x0, 90, gpu_threads=64)
    assert failed == 0
    assert st_o["depth"].max() > maxdepth                    # the extra doublings really extend trees beyond maxdepth
    for kw, field in ((dict(lane_chains=2), "lane_launches"), (dict(lane_chains=1, lane_groups=2), "group_launches")):
        pos_g, st_g, ex = run_engine(s, logp, n, x0, 90, splits=(50,), **kw)
        assert ex[field] == 2, (kw, ex)
        assert_bit_exact(pos_g, st_g, pos_o, st_o)
        assert ex["counters"]["total_leapfrogs"] == steps


def test_lane_kernel_k4_many_chains_equals_the_other_kernels(oracle):
    """K4's shape at scale: 20000 chains of the 8-schools model.  The lane kernel, the 8-lanes-per-chain kernel and the wave kernel
    give the same draws (whole run, every chain); sampled chains against the oracle; the state the host reads back agrees too."""
    n = 20000
    s = N.DiagNutsSettings(num_chains=n, seed=9, num_tune=50)
    logp = N.LogpSpec.eight_schools()
    x0 = oracle.init_positions_uniform(s.seed, 0, n, 10)
    res = {}
    for name, kw in (("lane", dict(lane_chains=2)), ("group", dict(lane_chains=1, lane_groups=2))):
        res[name] = run_engine(s, logp, n, x0, 80, splits=(50,), **kw)
    assert res["lane"][2]["lane_launches"] == 2 and res["group"][2]["group_launches"] == 2
    assert (res["lane"][0].view(np.uint64) == res["group"][0].view(np.uint64)).all()
    for f in res["lane"][1].dtype.names:
        a, b = res["lane"][1][f], res["group"][1][f]
        assert ((a == b) | ((a != a) & (b != b))).all(), f
    for k in ("stds", "mean", "step_sizes", "x", "gx"):
        assert (res["lane"][2][k].view(np.uint64) == res["group"][2][k].view(np.uint64)).all(), k


def test_lane_kernel_is_automatic_for_very_many_small_chains(oracle):
    n = 50000
    s = N.DiagNutsSettings(num_chains=n, seed=3, num_tune=30)
    b = N.ChainBatch(s, N.LogpSpec.iid_normal(10, 3.0), n)
    b.set_position(b.init_positions_uniform())
    pos, st = b.draw_many(40)
    # the engine's own choice: the warm-up's first 20 draws 8 chains per wavefront, everything after one chain per lane — the
    # draws compared below cross that switch
    assert b.lane_launches() == 2 and b.group_launches() == 1      # (draw_many cuts the call at the end of the warm-up: two lane launches)
    b.close()
    pick = [0, 77, 8191, 32767, n - 1]
    x0 = oracle.init_positions_uniform(s.seed, 0, n, 10)
    for c in pick:
        so = oracle_settings(oracle, s)
        ch = oracle.Chain(so, oracle.LOGP_IID_NORMAL, 10, np.array([3.0]), oracle.gpu_cfg(64), chain_id=c)
        assert ch.set_position(x0[c]) == 0
        for t in range(40):
            p, q, rc = ch.draw()
            assert rc == 0 and (p.view(np.uint64) == pos[t, c].view(np.uint64)).all(), (c, t)
            assert q["n_steps"] == st["n_steps"][t, c] and q["step_size"] == st["step_size"][t, c]


def test_lane_kernel_vector_statistics_and_divergences(oracle):
    """expanded_draw's vectors (gradient, transformed point, mass-matrix events, divergence locations) out of the lane kernel."""
    n, dim = 70, 7
    s = N.DiagNutsSettings(num_chains=n, seed=21, num_tune=40, max_energy_error=0.5, store_gradient=True, store_transformed=True,
                           store_divergences=True)
    s.adapt_options.mass_matrix_options.store_mass_matrix = True
    logp = N.LogpSpec.funnel(dim)
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    b = N.ChainBatch(s, logp, n, lane_chains=2)
    assert (b.set_position(x0) == 0).all()
    pos, st, vec = b.expanded_draw_many(70)
    assert b.lane_launches() == 1
    b.close()
    vo = {}
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, dim, logp.params, oracle.gpu_cfg(64), n, x0, 70,
                                        n_threads=8, vectors=vo)
    assert failed == 0
    assert_bit_exact(pos, st, pos_o, st_o)
    assert st["diverging"].sum() > 10
    for k in vec:
        both_nan = np.isnan(vec[k]) & np.isnan(vo[k])
        assert ((vec[k].view(np.uint64) == vo[k].view(np.uint64)) | both_nan).all(), k


def test_removed_lane_forms():
    """lane_chains = 3 is refused with a reason; dim 11 .. 16 has no one-chain-per-lane kernel any more (an explicit lane_chains = 2 runs the 8-lane kernels)."""
    s = N.DiagNutsSettings(num_chains=64, seed=3, num_tune=10)
    with pytest.raises(Exception) as e:
        N.ChainBatch(s, N.LogpSpec.iid_normal(8, 3.0), 64, lane_chains=3)
    assert "removed" in str(e.value)
    b = N.ChainBatch(s, N.LogpSpec.iid_normal(12, 3.0), 64, lane_chains=2)
    b.set_position(b.init_positions_uniform())
    b.draw_many(12)
    assert b.lane_launches() == 0
    b.close()
