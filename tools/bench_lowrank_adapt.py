"""LowRankNutsSettings end to end at scale: the whole warm-up with per-chain low-rank adaptation (draw kernels + the estimator
rounds between launches: on the device by default, --place host for the host threads), then sampling.  One JSON line: wall time
of the warm-up, the share spent outside the draw kernels, the estimator rounds' parts, ranks found, sampling throughput.

  python tools/bench_lowrank_adapt.py [--chains 1024] [--dim 128] [--tune 300] [--draws 100] [--place auto|host|device]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--tune", type=int, default=300)
    ap.add_argument("--draws", type=int, default=100)
    ap.add_argument("--place", default="auto")
    a = ap.parse_args()
    rng = np.random.default_rng(3)
    u = np.linalg.qr(rng.normal(size=(a.dim, 4)))[0]
    sigma = np.eye(a.dim) + u @ np.diag([100.0, 50.0, 20.0, 10.0]) @ u.T
    sc = np.exp(rng.normal(0, 0.5, a.dim))
    sigma = np.diag(sc) @ sigma @ np.diag(sc)
    prec = np.linalg.inv(sigma)
    prec = (prec + prec.T) / 2
    s = N.LowRankNutsSettings(num_chains=a.chains, seed=11, num_tune=a.tune, num_draws=a.draws)
    b = N.ChainBatch(s, N.LogpSpec.mvn_precision(prec), a.chains)
    b.set_lowrank_estimator_place(a.place)
    b.init_with_retries()
    t = time.time()
    _, st_w = b.draw_many(a.tune, positions=False)
    t_warm = time.time() - t
    c_w = b.counters()
    import ctypes as C
    tm = (C.c_double * 6)()
    N.load_library().nm_debug_lowrank_timing(b._h, tm)
    b.reset_counters()
    t = time.time()
    pos, st = b.draw_many(a.draws)
    t_s = time.time() - t
    c_s = b.counters()
    n_eig = b.lowrank()[0]
    sample = pos.reshape(-1, a.dim)
    cov_err = float(np.abs(np.cov(sample.T) - sigma).max() / np.abs(sigma).max())
    print(json.dumps({
        "config": f"LowRankNutsSettings, full-precision normal dim {a.dim} (4 strong directions) x {a.chains} chains, num_tune {a.tune}",
        "warmup_wall_s": t_warm, "warmup_kernel_s": c_w["kernel_ms"] * 1e-3, "warmup_launches": c_w["kernel_launches"],
        "estimator_place": a.place, "estimator_calls_on_device": b.lowrank_device_updates(),
        "warmup_share_outside_draw_kernels": 1.0 - c_w["kernel_ms"] * 1e-3 / t_warm,
        "warmup_host_share": (t_warm - c_w["kernel_ms"] * 1e-3 - (tm[2] if b.lowrank_device_updates() else 0.0)) / t_warm,
        "estimator_rounds_s": {"total": tm[0], "window_download": tm[1], "estimator_threads": tm[2], "upload_scatter": tm[3],
                               "rounds": int(tm[4]), "estimator_calls": int(tm[5])},
        "updates_per_chain": float((st_w["transformation_update_id"] >= 0).sum() / a.chains),
        "n_eig_median": float(np.median(n_eig)), "n_eig_max": int(n_eig.max()),
        "sampling_leapfrogs_per_s": c_s["total_leapfrogs"] / t_s, "sampling_leapfrogs_per_draw": float(st["n_steps"].mean()),
        "sampling_kernel_s": c_s["kernel_ms"] * 1e-3, "sampling_wall_s": t_s,
        "divergences": int(st["diverging"].sum()), "cov_rel_err": cov_err}))
    b.close()


if __name__ == "__main__":
    main()
