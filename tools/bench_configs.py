"""Throughput and tree statistics of BASELINE.json's other configurations on one GPU (K1 for scale).

  python tools/bench_configs.py [k3|k4|k5|k1|all] [--draws N] [--chains C] [--lane-groups 0|1|2] [--chain-tiles 0|1|2]

Prints one JSON line per configuration: M1 = leapfrog-steps*dims/s and M2 = draws/s/chain for the post-warm-up
draws, depth histogram, divergence rate, and the "lane utilisation" SURVEY asks for on the ragged config
(sum n_steps / (chains * max_c n_steps), averaged over draws) — the fraction of lockstep work a one-chain-per-lane
design would have kept busy; this engine's blocks are independent, so its own figure is the occupancy of the chip.
"""
import json
import os
import sys
import time

import numpy as np
import torch  # before the engine: both bring a HIP runtime and torch's must be the one that initialises

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402

def _k5_precision(d):
    """Sigma = I + 0.5 * 11' scaled like tests/sample_normal.rs:29-45 of the reference; its inverse is the precision."""
    sigma = np.eye(d) + 0.5 * np.ones((d, d)) / d
    p = np.linalg.inv(sigma)
    return (p + p.T) / 2


CONFIGS = {
    "k1": dict(name="K1 iid N(3,1) dim 10", logp=lambda: N.LogpSpec.iid_normal(10, 3.0), chains=4, tune=400),
    "k3": dict(name="K3 Neal's funnel dim 101", logp=lambda: N.LogpSpec.funnel(101), chains=8192, tune=400),
    "k5": dict(name="K5 normal with a full precision matrix dim 256, DiagNutsSettings (P x: matrix cores from 256 chains on, else per-chain GEMV)", logp=lambda: N.LogpSpec.mvn_precision(_k5_precision(256)),
               chains=4096, tune=400),
    "s24": dict(name="small chains: diag normal dim 24 (16 lanes per chain)", logp=lambda: N.LogpSpec.diag_normal(np.exp(np.linspace(-1, 1, 24))), chains=32768, tune=400),
    "s48": dict(name="small chains: diag normal dim 48 (32 lanes per chain)", logp=lambda: N.LogpSpec.diag_normal(np.exp(np.linspace(-1, 1, 48))), chains=32768, tune=400),
    "k4": dict(name="K4 8 schools non-centered dim 10 (one GPU's shard of 65536)", logp=N.LogpSpec.eight_schools,
               chains=8192, tune=400),
}


def run(key, draws, chains=0, lane_groups=0, chain_tiles=0, lane_chains=0):
    cfg = CONFIGS[key]
    logp = cfg["logp"]()
    C, D = chains or cfg["chains"], logp.dim
    s = N.DiagNutsSettings(num_chains=C, seed=20260928, num_tune=cfg["tune"], num_draws=draws)
    b = N.ChainBatch(s, logp, C, lane_groups=lane_groups, chain_tiles=chain_tiles, lane_chains=lane_chains)
    b.set_position(b.init_positions_uniform())
    t = time.time()
    _, st_w = b.draw_many(cfg["tune"], positions=False)
    t_warm = time.time() - t
    warm_kernel_ms = b.counters()["kernel_ms"]
    b.reset_counters()
    st_dev = torch.empty((draws, C, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t = time.time()
    b.draw_device(draws, 0, st_dev.data_ptr())
    dt = time.time() - t
    st = st_dev.cpu().numpy().view(N.STATS_DTYPE).reshape(draws, C)
    c = b.counters()
    steps = int(st["n_steps"].sum())
    assert steps == c["total_leapfrogs"]
    depth_hist = np.bincount(st["depth"].ravel().astype(np.int64), minlength=11)
    util = float((st["n_steps"].sum(axis=1) / (C * st["n_steps"].max(axis=1))).mean())
    out = {
        "config": cfg["name"], "chains": C, "dim": D, "draws": draws, "threads_per_chain": b.threads_per_chain(),
        "dims_per_lane": b.dims_per_lane(), "lane_groups": lane_groups, "lane_chains": lane_chains, "lane_launches": b.lane_launches(), "chain_tiles": chain_tiles, "matrix_core_launches": b.tile_launches(),
        "M1_steps_dims_per_s": steps * D / dt, "M2_draws_per_s_per_chain": draws / dt,
        "leapfrogs_per_s": steps / dt, "kernel_ms": c["kernel_ms"], "warmup_s": t_warm, "warmup_kernel_ms": warm_kernel_ms,
        "group_launches": b.group_launches(),
        "warmup_divergence_rate": float(st_w["diverging"].mean()),
        "mean_steps_per_draw": steps / (draws * C), "depth_histogram": depth_hist.tolist(),
        "divergence_rate": float(st["diverging"].mean()), "maxdepth_rate": float(st["maxdepth_reached"].mean()),
        "lockstep_lane_utilisation": util,
        "mean_step_size": float(st["step_size"][-1].mean()),
    }
    b.close()
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "all"
    draws = int(sys.argv[sys.argv.index("--draws") + 1]) if "--draws" in sys.argv else 200
    for k in (["k1", "k3", "k4", "k5"] if which == "all" else [which]):
        run(k, draws, chains=int(sys.argv[sys.argv.index("--chains") + 1]) if "--chains" in sys.argv else 0,
            lane_groups=int(sys.argv[sys.argv.index("--lane-groups") + 1]) if "--lane-groups" in sys.argv else 0,
            chain_tiles=int(sys.argv[sys.argv.index("--chain-tiles") + 1]) if "--chain-tiles" in sys.argv else 0,
            lane_chains=int(sys.argv[sys.argv.index("--lane-chains") + 1]) if "--lane-chains" in sys.argv else 0)
