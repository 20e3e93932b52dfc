"""LowRankMassMatrixStrategy::compute_update, restated step by step with LAPACK (numpy / scipy).

TEST INFRASTRUCTURE ONLY (part of the oracle).  Follows reference src/transform/adapt/low_rank.rs:
  rescale_points        :161-226
  compute_update        :73-142   (thin SVDs, concat, column-pivoted QR, projections, eigenvalue filter, back projection)
  estimate_mass_matrix  :228-260
  spd_mean              :262-290
The reference runs these on faer; decompositions are unique only up to signs / bases of degenerate spaces, but what the
transformation uses — U diag(f(lambda)) U' and mu — is basis independent, so results agree to rounding (~1e-12), not
bit for bit.  The C++ oracle calls `estimator_callback()` through ctypes (oracle.ESTIMATOR_FN).
"""
import numpy as np
import scipy.linalg


def rescale_points(draws, grads):
    """draws, grads: [ndim][ndraws] (modified copies are returned).  -> draws, grads, stds, mu, draw_mean, grad_mean"""
    draws, grads = draws.copy(), grads.copy()
    ndim, n = draws.shape
    stds, mu, dmo, gmo = np.zeros(ndim), np.zeros(ndim), np.zeros(ndim), np.zeros(ndim)
    for row in range(ndim):
        dm = draws[row].sum() / n
        gm = grads[row].sum() / n
        dv = ((draws[row] - dm) * (draws[row] - dm)).sum() / n
        gv = ((grads[row] - gm) * (grads[row] - gm)).sum() / n
        with np.errstate(all="ignore"):
            sigma = np.sqrt(np.sqrt(dv / gv))
        mu[row] = dm + sigma * sigma * gm
        stds[row] = sigma
        with np.errstate(all="ignore"):
            draws[row] = (draws[row] - mu[row]) * (1.0 / sigma)
            grads[row] = grads[row] * sigma
        dmo[row] = draws[row].sum() / n
        gmo[row] = grads[row].sum() / n
        draws[row] -= dmo[row]
        grads[row] -= gmo[row]
    return draws, grads, stds, mu, dmo, gmo


def _sym_fn(mat, fn):
    w, u = np.linalg.eigh(mat)
    return (u * fn(w)) @ u.T, w, u


def spd_mean(cov_draws, cov_grads):
    w, u = np.linalg.eigh(cov_grads)
    g_sqrt = (u * np.sqrt(w)) @ u.T
    m = g_sqrt @ cov_draws @ g_sqrt
    mw, mu_ = np.linalg.eigh(m)
    m_sqrt = (mu_ * np.sqrt(mw)) @ mu_.T
    g_inv_sqrt = (u * (1.0 / np.sqrt(w))) @ u.T
    return g_inv_sqrt @ m_sqrt @ g_inv_sqrt


def estimate_mass_matrix(draws, grads, gamma):
    cov_d = (draws @ draws.T) * (1.0 / gamma)
    cov_g = (grads @ grads.T) * (1.0 / gamma)
    cov_d[np.diag_indices_from(cov_d)] += 1.0
    cov_g[np.diag_indices_from(cov_g)] += 1.0
    mean = spd_mean(cov_d, cov_g)
    if not np.isfinite(mean).all():
        return None
    vals, vecs = np.linalg.eigh(mean)
    return vals, vecs


def _basis(a, rank_revealing, tol=1e-10):
    """orthonormal basis of the columns of `a`: the thin SVD's U (literal), or only its numerically non-null part"""
    u, sv, _ = np.linalg.svd(a, full_matrices=False)
    if rank_revealing and len(sv):
        u = u[:, sv > tol * sv[0]]
    return u


def compute_update(draws, grads, gamma=1e-5, eigval_cutoff=2.0, rank_revealing=False):
    """draws, grads: [ndim][ndraws].  -> (stds, mean, vals, vecs [ndim][n_eig], mu_low_rank) or None.

    rank_revealing=False follows the reference literally (thin SVD U with LAPACK's completion of null directions, full
    thin Q of the pivoted QR).  rank_revealing=True drops numerically null directions — in exact arithmetic the same
    result (those directions carry eigenvalue 1 and are filtered), in floating point free of the eps / gamma^2 noise the
    literal form has there; it is what the engine's host estimator (csrc/lowrank_host.cpp) computes."""
    draws, grads, stds, mean, draw_mean, grad_mean = rescale_points(np.asarray(draws, float), np.asarray(grads, float))
    if not (np.isfinite(draws).all() and np.isfinite(grads).all()):
        return None                                    # faer's SVD fails on non-finite input (`.ok()?`)
    try:
        ud = _basis(draws, rank_revealing)
        ug = _basis(grads, rank_revealing)
    except np.linalg.LinAlgError:
        return None
    subspace = np.concatenate([ud, ug], axis=1)
    if rank_revealing:
        q = _basis(subspace, True)
    else:
        q = scipy.linalg.qr(subspace, mode="economic", pivoting=True)[0]
    dp, gp = q.T @ draws, q.T @ grads
    est = estimate_mass_matrix(dp, gp, gamma)
    if est is None:
        return None
    vals, vecs = est
    keep = (vals > eigval_cutoff) | (vals < 1.0 / eigval_cutoff)
    vals, vecs = vals[keep], vecs[:, keep]
    vecs = q @ vecs
    b = vecs @ ((vals - 1.0) * (vecs.T @ grad_mean))
    mu = draw_mean + grad_mean + b
    return stds, mean, vals, vecs, mu


def estimator_callback(record=None, rank_revealing=False):
    """An oracle.ESTIMATOR_FN around compute_update.  `record` (a list) receives every (draws, grads, result)."""
    from . import oracle as O

    def cb(ctx, ndim, ndraws, draws, grads, gamma, cutoff, stds, mean, n_eig, vals, vecs, mu):
        d = np.ctypeslib.as_array(draws, shape=(ndraws, ndim)).T.copy()
        g = np.ctypeslib.as_array(grads, shape=(ndraws, ndim)).T.copy()
        res = compute_update(d, g, gamma, cutoff, rank_revealing)
        if record is not None:
            record.append((d, g, res))
        if res is None:
            return 1
        s, m, v, u, ml = res
        np.ctypeslib.as_array(stds, shape=(ndim,))[:] = s
        np.ctypeslib.as_array(mean, shape=(ndim,))[:] = m
        np.ctypeslib.as_array(mu, shape=(ndim,))[:] = ml
        k = len(v)
        n_eig[0] = k
        if k:
            np.ctypeslib.as_array(vals, shape=(k,))[:] = v
            np.ctypeslib.as_array(vecs, shape=(k, ndim))[:] = u.T
        return 0

    return O.ESTIMATOR_FN(cb)
