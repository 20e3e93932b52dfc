"""How far is the engine's built-in low-rank estimator (csrc/lowrank_host.cpp, rank-revealing) from the LITERAL reference
algorithm (src/transform/adapt/low_rank.rs:73-290 restated on LAPACK, oracle/lowrank.py rank_revealing=False)?

Runs on the CPU (the estimator is host code; the windows come from an oracle run with the literal estimator, so they are
the windows a real LowRankNutsSettings warm-up sees: the early ones have fewer draws than dims).  For every window:
  d_sigma  max relative difference of the diagonal scales            (rescale_points: identical arithmetic expected)
  d_op     || A_builtin - A_literal ||_2 / || A_literal ||_2,  A = I + U (diag(lambda)^1/2 - I) U'   (what the sampler applies)
  d_mu     max | mu_builtin - mu_literal | / (1 + max | mu_literal |)
  n_eig    eigenpairs kept by the two forms
and the same against the rank-revealing LAPACK form (what the built-in is designed to equal).

  python tools/estimator_departure.py [--dims 64,128] [--out profiles/r04a_estimator_departure.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def builtin_update(L, d, g, gamma, cutoff):
    """d, g: [ndim][ndraws] -> (stds, mean, vals, vecs [ndim][k], mu) through nm_lowrank_compute_update"""
    dim, n = d.shape
    dr, gr = np.ascontiguousarray(d.T), np.ascontiguousarray(g.T)
    m = min(dim, 2 * n)
    stds, mean, vals, vecs, mu = np.empty(dim), np.empty(dim), np.empty(m), np.empty((m, dim)), np.empty(dim)
    ne = C.c_uint64()
    rc = L.nm_lowrank_compute_update(None, dim, n, dr.ctypes.data, gr.ctypes.data, gamma, cutoff, stds.ctypes.data, mean.ctypes.data,
                                     C.byref(ne), vals.ctypes.data, vecs.ctypes.data, mu.ctypes.data)
    if rc != 0:
        return None
    k = ne.value
    return stds, mean, vals[:k].copy(), vecs[:k].T.copy(), mu


def op_of(vals, vecs):
    return np.eye(vecs.shape[0]) + vecs @ np.diag(np.sqrt(vals) - 1.0) @ vecs.T


def compare(a, b):
    """a, b: (stds, mean, vals, vecs, mu) -> dict of departures of a from b"""
    A, B = op_of(a[2], a[3]), op_of(b[2], b[3])
    return dict(d_sigma=float(np.max(np.abs(a[0] - b[0]) / np.abs(b[0]))),
                d_mean=float(np.max(np.abs(a[1] - b[1])) / (1.0 + np.max(np.abs(b[1])))),
                d_op=float(np.linalg.norm(A - B, 2) / np.linalg.norm(B, 2)),
                d_mu=float(np.max(np.abs(a[4] - b[4])) / (1.0 + np.max(np.abs(b[4])))),
                n_eig=(int(len(a[2])), int(len(b[2]))))


def windows_of_a_run(dim, n_chains, tune, seed):
    import nuts_rs_amd as N
    from oracle import oracle as O
    from oracle import lowrank as LR
    from helpers import oracle_settings
    from test_gpu_lowrank import correlated_precision
    rng = np.random.default_rng(dim)
    prec, _ = correlated_precision(rng, dim, max(2, dim // 20))
    logp = N.LogpSpec.mvn_precision(prec)
    s = N.LowRankNutsSettings(num_chains=n_chains, seed=seed, num_tune=tune)
    rec = []
    cb = LR.estimator_callback(rec, rank_revealing=False)
    x0 = O.init_positions_uniform(s.seed, 0, n_chains, dim)
    O.run(oracle_settings(O, s), logp.kind, dim, logp.params, O.gpu_cfg(64), n_chains, x0, tune, estimator=cb, want_positions=False)
    return rec, s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", default="64,128")
    ap.add_argument("--chains", type=int, default=2)
    ap.add_argument("--tune", type=int, default=300)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from nuts_rs_amd import _lib
    from oracle import lowrank as LR
    L = _lib.load()
    rows = []
    for dim in [int(x) for x in a.dims.split(",")]:
        rec, s = windows_of_a_run(dim, a.chains, a.tune, 11)
        gamma, cutoff = s.adapt_options.mass_matrix_gamma if hasattr(s.adapt_options, "mass_matrix_gamma") else 1e-5, 2.0
        for i, (d, g, lit) in enumerate(rec):
            n = d.shape[1]
            bi = builtin_update(L, d, g, 1e-5, 2.0)
            rr = LR.compute_update(d, g, 1e-5, 2.0, rank_revealing=True)
            row = dict(dim=dim, window=i, n_draws=n, rank_deficient=bool(n - 1 < dim), literal_ok=lit is not None, builtin_ok=bi is not None)
            if bi is not None and lit is not None:
                row["vs_literal"] = compare(bi, lit)
            if bi is not None and rr is not None:
                row["vs_rank_revealing_lapack"] = compare(bi, rr)
            if lit is not None and rr is not None:
                row["lapack_rr_vs_literal"] = compare(rr, lit)
            rows.append(row)
            print(json.dumps(row))
    if a.out:
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
