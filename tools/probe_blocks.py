"""Is the K2 kernel bound by per-block latency or by a shared resource?  Same per-block work, fewer resident blocks.

  python tools/probe_blocks.py      # chains = blocks, one chain per block, 200 post-warm-up draws
"""
import os
import sys
import time

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402

D = 1024
for blocks in (128, 256, 512, 768, 1024):
    s = N.DiagNutsSettings(num_chains=blocks, seed=20260928, num_tune=400, num_draws=200)
    b = N.ChainBatch(s, N.LogpSpec.iid_normal(D, 3.0), blocks, grid_blocks=blocks)
    b.set_position(b.init_positions_uniform())
    b.draw_device(400)
    b.reset_counters()
    t = time.time()
    b.draw_device(200)
    dt = time.time() - t
    c = b.counters()
    print("blocks %5d  kernel %.2f ms  us/draw/block %.1f  M1 %.4g  steps/draw %.2f" % (
        blocks, c["kernel_ms"], c["kernel_ms"] * 1e3 / 200, c["total_leapfrogs"] * D / dt, c["total_leapfrogs"] / 200 / blocks))
    b.close()
