"""Single-chain leapfrog latency — the figure of merit for ragged workloads (K3): a launch lasts as long as its slowest chain,
and one chain's draws are sequential, so what bounds K3 is microseconds per leapfrog of ONE chain, not chip throughput.

Deep trees on demand: a fixed, small step size and no jitter make every tree reach `maxdepth` (2^maxdepth - 1 leapfrogs per
draw, every merge level exercised), on any density.  Reports us per leapfrog for 1 chain (unloaded latency), for one chain per
CU, and for a full grid (the latency a chain sees inside a busy launch).

  python tools/leaf_latency.py [--logp funnel|iid|schools] [--dim 101] [--maxdepth 8] [--draws 20] [--step 0.01] [--chains 1,256,8192]
"""
import argparse
import json
import os
import sys

import torch  # noqa: F401  (its HIP runtime initialises first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logp", default="funnel")
    ap.add_argument("--dim", type=int, default=101)
    ap.add_argument("--maxdepth", type=int, default=8)
    ap.add_argument("--draws", type=int, default=20)
    ap.add_argument("--step", type=float, default=0.002)
    ap.add_argument("--chains", default="1,256,1024,8192")
    ap.add_argument("--lane-groups", type=int, default=0)
    ap.add_argument("--lane-chains", type=int, default=0)
    ap.add_argument("--no-turn", action="store_true", help="check_turning = false: the tree without its U-turn tests")
    a = ap.parse_args()
    out = []
    for nc in [int(x) for x in a.chains.split(",")]:
        logp = {"funnel": lambda: N.LogpSpec.funnel(a.dim), "iid": lambda: N.LogpSpec.iid_normal(a.dim, 3.0),
                "schools": N.LogpSpec.eight_schools}[a.logp]()
        s = N.DiagNutsSettings(num_chains=nc, seed=11, num_tune=1, num_draws=a.draws, maxdepth=a.maxdepth, check_turning=not a.no_turn)
        st = s.adapt_options.step_size_settings
        st.method, st.fixed_step_size, st.jitter = N.sampler.STEP_FIXED, a.step, None
        b = N.ChainBatch(s, logp, nc, lane_groups=a.lane_groups, lane_chains=a.lane_chains)
        b.set_position(b.init_positions_uniform())
        b.draw_device(2)                      # warm the caches / code
        b.reset_counters()
        b.draw_device(a.draws)
        c = b.counters()
        per_chain = c["total_leapfrogs"] / nc
        row = {"lib": os.path.basename(os.environ.get("NUTS_AMD_LIB", "libnuts_amd.so")), "no_turn": a.no_turn, "logp": a.logp, "dim": logp.dim, "chains": nc, "maxdepth": a.maxdepth, "draws": a.draws, "leapfrogs_per_chain": per_chain,
               "kernel_ms": c["kernel_ms"], "us_per_leapfrog_of_one_chain": c["kernel_ms"] * 1e3 / per_chain,
               "leapfrogs_per_s": c["total_leapfrogs"] / (c["kernel_ms"] * 1e-3), "threads_per_chain": b.threads_per_chain(),
               "dims_per_lane": b.dims_per_lane(), "group_launches": b.group_launches(), "lane_launches": b.lane_launches(), "lane_chains": a.lane_chains}
        b.close()
        print(json.dumps(row), flush=True)
        out.append(row)
    return out


if __name__ == "__main__":
    main()
