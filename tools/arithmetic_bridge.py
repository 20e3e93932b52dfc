"""How far does the engine's arithmetic (restated exp / ln, lane-order sums: oracle gpu_cfg) follow the reference's
(libm, pulp-style SIMD-order sums: oracle ref_cfg) on seeded chains?  north_star asks for draws "within 1e-9 relative"
of the CpuMath path; NUTS trees are discontinuous in the last bits of a dot product, so a seed can depart at some draw
and never return.  This tool runs both arithmetics on BASELINE's densities over several seeds and reports, per
configuration: the fraction of chains that stay within 1e-9 for the whole run, the first draw at which any chain
departs (tree size or 1e-9), the worst relative difference before that, and the same for SIMD widths 2 / 4 / 8 of the
reference against each other (the reference is not bit-stable across CPUs either: pulp picks the width at run time).

  python tools/arithmetic_bridge.py [--seeds 6] [--draws 500] [--out profiles/r02_arithmetic_bridge.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def configs():
    rng = np.random.default_rng(1)
    u = np.linalg.qr(rng.normal(size=(256, 4)))[0]
    sigma = np.eye(256) + u @ np.diag([30.0, 20.0, 10.0, 5.0]) @ u.T
    p = np.linalg.inv(sigma)
    return {
        "iid_normal_1024": (O.LOGP_IID_NORMAL, 1024, [3.0], 64),
        "funnel_101": (O.LOGP_FUNNEL, 101, [], 64),
        "eight_schools_10": (O.LOGP_EIGHT_SCHOOLS, 10, [28., 8., -3., 7., -1., 1., 18., 12., 15., 10., 16., 11., 9., 11., 10., 18.], 64),
        "mvn_precision_256": (O.LOGP_MVN_PREC, 256, ((p + p.T) / 2).reshape(-1), 64),
        "iid_normal_50": (O.LOGP_IID_NORMAL, 50, [3.0], 64),
    }


def compare(a, b, tol=1e-9):
    """per chain: first draw where tree size differs or positions differ by more than tol (relative to the draw's scale)"""
    pa, sa = a[0], a[1]
    pb, sb = b[0], b[1]
    n_draws, n_chains = sa.shape
    first = np.full(n_chains, n_draws)
    worst = np.zeros(n_chains)
    for c in range(n_chains):
        for t in range(n_draws):
            scale = max(1.0, np.abs(pa[t, c]).max())
            rel = np.abs(pa[t, c] - pb[t, c]).max() / scale
            if sa["n_steps"][t, c] != sb["n_steps"][t, c] or sa["depth"][t, c] != sb["depth"][t, c] or not rel <= tol:
                first[c] = t
                break
            worst[c] = max(worst[c], rel)
    return first, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=6)
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--draws", type=int, default=500)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_arithmetic_bridge.json"))
    a = ap.parse_args()
    report = {"tolerance": 1e-9, "draws": a.draws, "chains_per_seed": a.chains, "seeds": a.seeds,
              "note": "draws 0-399 are the warm-up (mass matrix + step size adapting every draw), the rest sampling"}
    for name, (kind, dim, params, tpc) in configs().items():
        rows = {"gpu_vs_ref": [], "ref_simd2_vs_simd4": [], "ref_simd8_vs_simd4": []}
        for seed in range(1, a.seeds + 1):
            s = O.default_settings(seed=seed, num_chains=a.chains)
            x0 = O.init_positions_uniform(seed, 0, a.chains, dim)
            run = lambda cfg: O.run(s, kind, dim, params, cfg, a.chains, x0, a.draws, n_threads=8)
            ref4, gpu, ref2, ref8 = run(O.ref_cfg(4)), run(O.gpu_cfg(tpc)), run(O.ref_cfg(2)), run(O.ref_cfg(8))
            for key, other in (("gpu_vs_ref", gpu), ("ref_simd2_vs_simd4", ref2), ("ref_simd8_vs_simd4", ref8)):
                first, worst = compare(ref4, other)
                rows[key].append({"seed": seed, "first_departure": [int(f) if f < a.draws else None for f in first],
                                  "worst_rel_before": [float(w) for w in worst]})
        out = {}
        for key, rr in rows.items():
            firsts = [f for r in rr for f in r["first_departure"]]
            stay = [f is None for f in firsts]
            dep = [f for f in firsts if f is not None]
            out[key] = {"chains": len(firsts), "within_tolerance_whole_run": float(np.mean(stay)),
                        "earliest_departure_draw": min(dep) if dep else None, "median_departure_draw": float(np.median(dep)) if dep else None,
                        "worst_rel_diff_before_departure": max(w for r in rr for w in r["worst_rel_before"]), "per_seed": rr}
        report[name] = out
        print(name, {k: (v["within_tolerance_whole_run"], v["earliest_departure_draw"], v["worst_rel_diff_before_departure"]) for k, v in out.items()}, flush=True)
    json.dump(report, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
