"""K4 (8 schools x 65536 chains) warm-up cut into launches of 10 draws: kernel milliseconds per chunk for the 8-lane kernels
(lane_chains 1) and the one-chain-per-lane kernels (lane_chains 2), with the tree statistics of each chunk — where the 400 warm-up
draws spend their time.  One JSON line per form."""
import json
import os
import sys

import numpy as np
import torch  # noqa: F401  (before the engine: torch's HIP runtime initialises first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402

C = int(sys.argv[sys.argv.index("--chains") + 1]) if "--chains" in sys.argv else 65536
CH = 10
for lc in (1, 2):
    s = N.DiagNutsSettings(num_chains=C, seed=20260928, num_tune=400, num_draws=200)
    b = N.ChainBatch(s, N.LogpSpec.eight_schools(), C, lane_chains=lc)
    b.set_position(b.init_positions_uniform())
    ms, mean_steps, max_steps, slots64, slots8 = [], [], [], [], []
    for k in range(600 // CH):
        b.reset_counters()
        _, st = b.draw_many(CH, positions=False)
        ms.append(round(b.counters()["kernel_ms"], 3))
        n = st["n_steps"].astype(np.int64)
        mean_steps.append(round(float(n.mean()), 2))
        max_steps.append(int(n.max()))
        slots64.append(round(float(n.reshape(CH, C // 64, 64).max(axis=2).mean()), 2))
        slots8.append(round(float(n.reshape(CH, C // 8, 8).max(axis=2).mean()), 2))
    print(json.dumps({"chains": C, "lane_chains": lc, "chunk_draws": CH, "warmup_ms": round(sum(ms[:400 // CH]), 2), "sampling_ms": round(sum(ms[400 // CH:]), 2),
                      "kernel_ms_per_chunk": ms, "mean_steps": mean_steps, "max_steps": max_steps, "slots_per_draw_64": slots64, "slots_per_draw_8": slots8,
                      "lane_launches": b.lane_launches(), "group_launches": b.group_launches()}))
    b.close()
