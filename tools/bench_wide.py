"""Throughput of chains wider than one block (dim > 4096, csrc/kern_cluster.hip) on one GPU: iid N(3,1), post-warm-up.

  python tools/bench_wide.py [dim ...]

One JSON line per dim: steps*dims/s, leapfrogs per draw, blocks per chain, chains in flight."""
import json
import os
import sys

import torch  # before the engine: torch's HIP runtime must be the one that initialises

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402

for dim in [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768, 65536]:
    k = -(-dim // 4096)
    chains = 256 if dim <= 4096 else (256 // (8 * k)) * 8          # one round of resident chains
    s = N.DiagNutsSettings(num_chains=chains, seed=20260928, num_tune=200)
    b = N.ChainBatch(s, N.LogpSpec.iid_normal(dim, 3.0), chains)
    b.set_position(b.init_positions_uniform())
    b.draw_device(200)
    b.reset_counters()
    draws = 50
    pos = torch.empty((draws, chains, dim), dtype=torch.float64, device="cuda")
    st = torch.empty((draws, chains, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    b.draw_device(draws, pos.data_ptr(), st.data_ptr())
    c = b.counters()
    print(json.dumps({"dim": dim, "blocks_per_chain": b.blocks_per_chain(), "chains": chains, "draws": draws, "kernel_ms": c["kernel_ms"],
                      "steps_dims_per_s": c["total_leapfrogs"] * dim / (c["kernel_ms"] * 1e-3),
                      "leapfrogs_per_draw": c["total_leapfrogs"] / (draws * chains),
                      "us_per_leapfrog_per_chain": c["kernel_ms"] * 1e3 / (c["total_leapfrogs"] / chains),
                      "mean": float(pos.mean()), "var": float(pos.var())}), flush=True)
    b.close()
