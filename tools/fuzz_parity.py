"""Randomised parity sweep: engine (C ABI) against the oracle, bit for bit, over random densities, dims, chain counts, tilings,
settings, samplers and trajectory kinds.  Not part of the test suite (minutes on a GPU); prints one line per case and a summary.

  python tools/fuzz_parity.py [--cases 100] [--seed 1] [--scale]

--scale: thousands of chains per case (late blocks of big grids, several chains per wavefront, matrix-core kernels at their real
sizes); ten randomly picked chains of each run are compared with the oracle."""
import argparse
import os
import sys
import traceback

import numpy as np
import torch  # noqa: F401  (initialises the HIP runtime first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nuts_rs_amd as N  # noqa: E402
from oracle import oracle as O  # noqa: E402
from helpers import STAT_FIELDS_EXACT, oracle_settings  # noqa: E402


def make_case(rng, scale=False):
    dens = rng.choice(["iid", "diag", "funnel", "schools", "mvn"], p=[0.3, 0.25, 0.2, 0.1, 0.15])
    if dens == "schools":
        dim = 10
    elif dens == "mvn":
        dim = int(rng.choice([3, 17, 64, 100, 200, 256, 300]))
    elif dens == "funnel":
        dim = int(rng.choice([2, 5, 11, 40, 101, 130, 300]))
    else:
        dim = int(rng.choice([1, 2, 4, 7, 9, 10, 13, 16, 33, 64, 65, 128, 129, 257, 511, 1024, 1500, 2048, 3000, 4096, 5000, 9000]))
    sampler = rng.choice(["nuts", "exact", "micro", "mclmc"], p=[0.5, 0.15, 0.15, 0.2])
    if sampler in ("micro", "mclmc") and dim < 2:
        dim = 2
    if dim > 4096 and dens not in ("iid", "diag"):
        dim = 4096
    n = int(rng.choice([1, 2, 3, 5, 8, 17, 33]))
    if scale:
        n = int(rng.choice([1100, 2500, 4200, 9000, 20000]))
        dim = min(dim, 300)
    kw = dict(num_chains=n, seed=int(rng.integers(0, 2**31)), num_tune=int(rng.choice([20, 40, 60, 100])))
    if sampler == "mclmc":
        kw.update(step_size=float(rng.choice([0.2, 0.5, 0.9])), momentum_decoherence_length=float(rng.choice([1.0, 3.0, 7.0])),
                  trajectory_kind=int(rng.integers(0, 3)), dynamic_step_size=bool(rng.integers(0, 2)),
                  subsample_frequency=float(rng.choice([0.0, 0.5, 1.0])), max_energy_error=float(rng.choice([1000.0, 5.0, 0.5])),
                  trajectory_switch_fraction=float(rng.choice([0.0, 0.3, 0.9])))
        s = N.DiagMclmcSettings(**kw)
    else:
        kw.update(maxdepth=int(rng.choice([10, 10, 6, 3])), mindepth=int(rng.choice([0, 0, 1, 2])), extra_doublings=int(rng.choice([0, 0, 1])),
                  max_energy_error=float(rng.choice([1000.0, 1000.0, 3.0, 0.3])), check_turning=bool(rng.random() > 0.1),
                  target_integration_time=None if rng.random() > 0.15 else float(rng.choice([0.5, 2.0])),
                  trajectory_kind={"nuts": 0, "exact": 1, "micro": 2}[sampler])
        if kw["mindepth"] > kw["maxdepth"]:
            kw["mindepth"] = 0
        s = N.DiagNutsSettings(**kw)
        st = s.adapt_options.step_size_settings
        r = rng.random()
        if r < 0.15:
            st.method = N.STEP_ADAM
        elif r < 0.25:
            st.method, st.fixed_step_size = N.STEP_FIXED, float(rng.choice([0.1, 0.4]))
        if rng.random() < 0.2:
            st.jitter = None
        if rng.random() < 0.2:
            st.target_accept = float(rng.choice([0.6, 0.95]))
        if rng.random() < 0.2:
            s.adapt_options.mass_matrix_options.use_grad_based_estimate = False
    prng = np.random.default_rng(kw["seed"])
    if dens == "iid":
        logp = N.LogpSpec.iid_normal(dim, float(prng.normal()))
    elif dens == "diag":
        logp = N.LogpSpec.diag_normal(np.exp(prng.uniform(-4, 4, dim)))
    elif dens == "funnel":
        logp = N.LogpSpec.funnel(dim)
    elif dens == "schools":
        logp = N.LogpSpec.eight_schools()
    else:
        a = prng.normal(size=(dim, dim))
        p = a @ a.T / dim + np.eye(dim)
        logp = N.LogpSpec.mvn_precision((p + p.T) / 2)
    transform = None
    if sampler != "mclmc" and not scale and dim <= 1024 and rng.random() < 0.2:       # a given low-rank transformation (frozen)
        rank = int(rng.integers(0, min(dim, 12) + 1))
        per_chain = n if rng.random() < 0.5 else 0
        shp = (per_chain,) if per_chain else ()
        stds = np.exp(prng.normal(0, 0.5, shp + (dim,))); mean = prng.normal(0, 1, shp + (dim,)); mu = prng.normal(0, 0.3, shp + (dim,))
        vals = np.exp(prng.uniform(-2, 3, shp + (rank,)))
        mk = lambda: np.linalg.qr(prng.normal(size=(dim, max(rank, 1))))[0].T[:rank]
        vecs = np.stack([mk() for _ in range(per_chain)]) if per_chain else mk()
        transform = (stds, mean, vals, np.ascontiguousarray(vecs).reshape(shp + (rank, dim)), mu)
        s = N.LowRankNutsSettings(freeze_transform=True, **kw)
        sampler = "lowrank-" + sampler
    elif sampler != "mclmc" and not scale and dim <= 256 and rng.random() < 0.25:      # LowRankNutsSettings ADAPTING: the estimator kernel
        lk = dict(kw)                                                                    # (engine) against its host twin (oracle)
        lk["num_tune"] = int(rng.choice([60, 100, 150]))
        s = N.LowRankNutsSettings(**lk)
        s.adapt_options.mass_matrix_update_freq = int(rng.choice([1, 5, 20]))
        transform = "adapt"
        kw = lk
        sampler = "lowrank-adapt-" + sampler
    eng = {}
    if sampler == "nuts" and dens != "schools" and dim <= 4096:
        from nuts_rs_amd.build import pick_tiling
        combos = [(d, w) for d, w in ((2, 1), (4, 1), (8, 1), (16, 1), (8, 2), (16, 2), (4, 4), (16, 4)) if d * 64 * w >= dim]
        if dens == "mvn":
            combos = [c for c in combos if c != (16, 4)]
        d, w = combos[int(rng.integers(0, len(combos)))]
        eng.update(dims_per_lane=d, waves_per_chain=w)
    eng["lane_groups"] = int(rng.choice([0, 1, 2]))
    eng["chain_tiles"] = int(rng.choice([0, 1, 2]))
    eng["lane_chains"] = int(rng.choice([0, 1, 2, 2]))            # one chain per lane (dim <= 16, nuts_lane.hpp) whenever it applies
    draws = kw["num_tune"] + int(rng.choice([10, 30]))
    desc = f"{dens} dim {dim} n {n} {sampler} tune {kw['num_tune']} draws {draws} eng {eng}"
    return s, logp, n, draws, eng, desc, transform


def run_case(s, logp, n, draws, eng, rng=None, transform=None):
    x0 = O.init_positions_uniform(s.seed, 0, n, logp.dim)
    b = N.ChainBatch(s, logp, n, **eng)
    status = b.set_position(x0, raise_on_error=False)
    adapt = isinstance(transform, str)
    if adapt:
        transform = None
        b.set_lowrank_estimator_place("device")
    if transform is not None and (status == 0).all():
        b.set_transform(*transform)
    cut = draws // 2
    if (status == 0).all():
        pa, sa = b.draw_many(cut, raise_on_error=False)
        pb, sb = b.draw_many(draws - cut, raise_on_error=False)
        pos, st = np.concatenate([pa, pb]), np.concatenate([sa, sb])
    tpc, k, order = b.threads_per_chain(), b.blocks_per_chain(), b.reduce_order()
    b.close()
    cfg = O.gpu_cfg(tpc, gpu_slice=4096 if k > 1 else 0, lr_seq_dots=order if transform is not None else 0)
    so = oracle_settings(O, s)
    est = {}
    if adapt:
        import ctypes as C
        from nuts_rs_amd import _lib
        est = dict(estimator=C.cast(_lib.load().nm_lowrank_block_twin, O.ESTIMATOR_FN))
    # per chain through the step-wise interface so that a failed chain does not end the comparison
    if n > 100:          # a sample of the chains, each from its own global id
        picks = sorted(set([0, n - 1] + [int(c) for c in rng.integers(0, n, 8)]))
        outs = [O.run(so, logp.kind, logp.dim, logp.params, cfg, 1, x0[c:c + 1], draws, chain_offset=c, n_threads=1, **est) for c in picks]
        pos_o = np.concatenate([o[0] for o in outs], axis=1); st_o = np.concatenate([o[1] for o in outs], axis=1)
        failed = sum(o[3] for o in outs)
        if (status == 0).all():
            pos, st = pos[:, picks], st[:, picks]
        status = status[picks]
    else:
        pos_o, st_o, _, failed = O.run(so, logp.kind, logp.dim, logp.params, cfg, n, x0, draws, n_threads=8, transform=transform, **est)
    if not (status == 0).all():
        return "init-failed" if failed else "MISMATCH: engine refused an initial point the oracle accepts"
    if failed:
        return "MISMATCH: oracle chain failed, engine did not" if (st["chain_status"] == 0).all() else "both-failed"
    if not (pos.view(np.uint64) == pos_o.view(np.uint64)).all():
        t, c = np.argwhere((pos != pos_o).any(axis=2))[0]
        return f"MISMATCH: positions first at draw {t} chain {c}"
    for f in list(STAT_FIELDS_EXACT) + ["step_size", "energy", "logp"]:
        a, bb = st[f], st_o[f]
        same = (a == bb) | (np.isnan(a.astype(float)) & np.isnan(bb.astype(float))) if a.dtype.kind == "f" else (a == bb)
        if not same.all():
            t, c = np.argwhere(~same)[0]
            return f"MISMATCH: stat {f} first at draw {t} chain {c}: {a[t, c]} vs {bb[t, c]}"
    return "ok"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scale", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated case indices to run (every case is still generated: same cases as a full run; not with --scale)")
    a = ap.parse_args()
    only = {int(x) for x in a.only.split(",") if x}
    rng = np.random.default_rng(a.seed)
    tally = {}
    for i in range(a.cases):
        s, logp, n, draws, eng, desc, transform = make_case(rng, a.scale)
        if only and i not in only:
            continue
        try:
            res = run_case(s, logp, n, draws, eng, rng, transform)
        except N.NutsAmdError as e:
            res = "unsupported: " + str(e)[:80]
        except Exception:
            res = "ERROR: " + traceback.format_exc().splitlines()[-1][:160]
        key = res.split(":")[0]
        tally[key] = tally.get(key, 0) + 1
        print(f"[{i}] {res:14s} {desc}", flush=True)
    print("summary:", tally)
