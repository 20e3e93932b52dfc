"""End-to-end statistical checks of the sampler on distributions with known moments (many chains make the tolerances
tight).  Parity with the oracle is tested elsewhere; this guards against an error shared by engine and oracle."""
import numpy as np
import pytest

import nuts_rs_amd as N

pytestmark = pytest.mark.gpu


def run(logp, n_chains=1024, tune=300, draws=150, seed=11, **kw):
    s = N.DiagNutsSettings(num_chains=n_chains, seed=seed, num_tune=tune, num_draws=draws, **kw)
    b = N.ChainBatch(s, logp, n_chains)
    b.set_position(b.init_positions_uniform())
    b.draw_many(tune, positions=False)
    pos, st = b.draw_many(draws)
    b.close()
    return pos, st


def test_diag_normal_moments_and_acceptance():
    prec = np.exp(np.random.default_rng(0).uniform(-4, 4, 40))           # scales e^-2 .. e^2
    pos, st = run(N.LogpSpec.diag_normal(prec))
    x = pos.reshape(-1, 40)                                               # 153600 draws, ESS >> 1e4
    assert np.abs(x.mean(axis=0) * np.sqrt(prec)).max() < 0.03           # mean 0 in units of sigma
    assert np.abs(x.var(axis=0) * prec - 1.0).max() < 0.04               # variance 1/p
    assert st["diverging"].sum() == 0
    assert abs(st["mean_tree_accept"].mean() - 0.8) < 0.05               # dual averaging reached target_accept
    assert (st["tuning"] == 0).all() and (st["depth"] >= 1).all()


def test_full_precision_normal_covariance():
    rng = np.random.default_rng(1)
    a = rng.normal(size=(12, 12))
    p = a @ a.T / 12 + 0.5 * np.eye(12)
    p = (p + p.T) / 2
    pos, st = run(N.LogpSpec.mvn_precision(p), n_chains=2048, draws=100)
    x = pos.reshape(-1, 12)
    cov, want = np.cov(x.T), np.linalg.inv(p)
    assert np.abs(cov - want).max() < 0.05 * np.abs(want).max()
    assert np.abs(x.mean(axis=0)).max() < 0.03 * np.sqrt(np.diag(want)).max()


def test_funnel_neck_variable_and_eight_schools():
    hi = N.EuclideanAdaptOptions(step_size_settings=N.StepSizeSettings(target_accept=0.95))
    pos, st = run(N.LogpSpec.funnel(11), n_chains=4096, tune=400, draws=100, adapt_options=hi)
    v = pos[:, :, 0].ravel()
    # v ~ N(0, 9).  NUTS with a diagonal metric is known to under-explore the neck of a centered funnel (negative v;
    # the reference and Stan behave the same), so only the wide half is compared with the truth, loosely.
    q75, q90 = np.quantile(v, [0.75, 0.9])
    assert abs(q75 - 2.02) < 0.8 and abs(q90 - 3.84) < 0.8 and np.quantile(v, 0.1) < -1.5
    assert 0 < st["diverging"].mean() < 0.05
    pos, st = run(N.LogpSpec.eight_schools(), n_chains=4096, tune=300, draws=100)
    mu, tau = pos[:, :, 0].ravel(), np.exp(pos[:, :, 1].ravel())
    # reference values of the classic 8-schools posterior (Gelman et al. BDA3 §5.5; Stan case study): mu ~ 4.4 +- 3.3, tau median ~ 2.7
    assert abs(mu.mean() - 4.4) < 0.3 and abs(mu.std() - 3.3) < 0.4
    assert abs(np.median(tau) - 2.7) < 0.4
    assert st["diverging"].mean() < 0.01
