"""Throughput of the other integrators on one GPU: NUTS with trajectory_kind ExactNormal / Microcanonical and the
MCLMC sampler, against Euclidean NUTS, on the K2 density (iid N(3,1) dim 1024 x 4096 chains) and on the funnel (dim 101 x 8192).

  python tools/bench_kinds.py [--draws N]

One JSON line per case: leapfrogs/s and steps*dims/s of the post-warm-up launch, mean leapfrogs per draw, posterior moments."""
import argparse
import json
import os
import sys

import numpy as np
import torch  # before the engine: torch's HIP runtime must be the one that initialises

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402


def run(name, settings, logp, chains, tune, draws, **eng):
    b = N.ChainBatch(settings, logp, chains, **eng)
    b.set_position(b.init_positions_uniform())
    b.draw_device(tune)
    b.reset_counters()
    pos = torch.empty((draws, chains, logp.dim), dtype=torch.float64, device="cuda")
    st = torch.empty((draws, chains, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    b.draw_device(draws, pos.data_ptr(), st.data_ptr())
    c = b.counters()
    s = st.cpu().numpy().view(N.STATS_DTYPE).reshape(draws, chains)
    out = {"case": name, "chains": chains, "dim": logp.dim, "draws": draws, "kernel_ms": c["kernel_ms"],
           "leapfrogs_per_s": c["total_leapfrogs"] / (c["kernel_ms"] * 1e-3),
           "steps_dims_per_s": c["total_leapfrogs"] * logp.dim / (c["kernel_ms"] * 1e-3),
           "draws_per_s_per_chain": draws / (c["kernel_ms"] * 1e-3),
           "leapfrogs_per_draw": c["total_leapfrogs"] / (draws * chains), "divergence_rate": float(s["diverging"].mean()),
           "mean_step_size": float(s["step_size"][-1].mean()),
           "first_coordinate_mean": float(pos[:, :, 0].mean()), "first_coordinate_var": float(pos[:, :, 0].var()),
           "group_launches": b.group_launches(), "lane_launches": b.lane_launches()}
    b.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=100)
    ap.add_argument("--chains", type=int, default=8192, help="--k4: chains (65536: the whole K4 job on one GPU, one chain per lane where that applies)")
    ap.add_argument("--k4", action="store_true", help="only the K4-shaped cases: 8 schools dim 10 x 8192 chains, the kinds on the small-chain kernels and without them")
    a = ap.parse_args()
    K, M = N.KineticEnergyKind, N.MclmcTrajectoryKind
    if a.k4:
        nch = a.chains
        base = dict(num_chains=nch, seed=20260928, num_tune=400)
        auto = "automatic (8 chains per wavefront)" if nch < 24576 else "automatic (one chain per lane)"
        for kname, kind in (("euclidean", K.EUCLIDEAN), ("exact_normal", K.EXACT_NORMAL), ("microcanonical", K.MICROCANONICAL)):
            for lg, lc, form in ((1, 1, "one chain per wavefront"), (2, 1, "8 chains per wavefront"), (0, 0, auto)):
                if (lg, lc) == (2, 1) and nch < 24576:
                    continue
                run(f"k4 8 schools nuts {kname}: {form}", N.DiagNutsSettings(trajectory_kind=kind, max_energy_error=50.0 if kind == K.MICROCANONICAL else 1000.0, **base),
                    N.LogpSpec.eight_schools(), nch, 400, a.draws, lane_groups=lg, lane_chains=lc)
        for lg, form in ((1, "one chain per wavefront"), (0, "automatic (8 chains per wavefront)")):
            run(f"k4 8 schools mclmc: {form}", N.DiagMclmcSettings(step_size=0.4, max_energy_error=30.0, **base), N.LogpSpec.eight_schools(), nch, 400, a.draws,
                lane_groups=lg, lane_chains=1)
        sys.exit(0)
    for dens, chains, mk in (("k2", 4096, lambda: N.LogpSpec.iid_normal(1024, 3.0)), ("k3", 8192, lambda: N.LogpSpec.funnel(101))):
        base = dict(num_chains=chains, seed=20260928, num_tune=400)
        run(dens + " nuts euclidean", N.DiagNutsSettings(**base), mk(), chains, 400, a.draws)
        run(dens + " nuts exact_normal", N.DiagNutsSettings(trajectory_kind=K.EXACT_NORMAL, **base), mk(), chains, 400, a.draws)
        run(dens + " nuts microcanonical", N.DiagNutsSettings(trajectory_kind=K.MICROCANONICAL, **base), mk(), chains, 400, a.draws)
        run(dens + " mclmc microcanonical", N.DiagMclmcSettings(trajectory_kind=M.MICROCANONICAL, **base), mk(), chains, 400, a.draws)
        run(dens + " mclmc euclidean", N.DiagMclmcSettings(trajectory_kind=M.EUCLIDEAN, step_size=0.3, **base), mk(), chains, 400, a.draws)
