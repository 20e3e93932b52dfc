"""What recording costs on K4 (8 schools): 200 post-warm-up draws with no outputs, positions only, statistics only, both — 65536 chains (one chain
per lane) and 8192 chains (8 chains per wavefront); kernel ms, twice."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402

draws = 200
for C in (65536, 8192):
    s = N.DiagNutsSettings(num_chains=C, seed=20260928, num_tune=400, num_draws=draws)
    b = N.ChainBatch(s, N.LogpSpec.eight_schools(), C)
    b.set_position(b.init_positions_uniform())
    b.draw_device(400)
    pos = torch.empty((draws, C, 10), dtype=torch.float64, device="cuda")
    st = torch.empty((draws, C, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    for rep in range(2):
        for name, p, q in (("none", 0, 0), ("positions", pos.data_ptr(), 0), ("stats", 0, st.data_ptr()), ("both", pos.data_ptr(), st.data_ptr())):
            b.reset_counters()
            b.draw_device(draws, p, q)
            c = b.counters()
            print(f"chains {C:6d} {name:10s} kernel_ms {c['kernel_ms']:.2f}  leapfrogs/s {c['total_leapfrogs'] / (c['kernel_ms'] * 1e-3):.4g}")
    b.close()
