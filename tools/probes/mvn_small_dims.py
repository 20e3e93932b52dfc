import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import nuts_rs_amd as N
from bench_configs import _k5_precision
for dim in (16, 32, 64, 128):
    for lg, ct, name in ((0, 1, "auto-group/one-chain"), (1, 2, "matrix cores"), (1, 1, "one chain per wave")):
        n = 8192
        s = N.DiagNutsSettings(num_chains=n, seed=3, num_tune=200)
        b = N.ChainBatch(s, N.LogpSpec.mvn_precision(_k5_precision(dim)), n, lane_groups=lg, chain_tiles=ct)
        b.set_position(b.init_positions_uniform())
        b.draw_device(200); b.reset_counters(); b.draw_device(100)
        c = b.counters()
        print(dim, name, "group", b.group_launches(), "tile", b.tile_launches(), "leapfrogs/s %.3g" % (c["total_leapfrogs"] / (c["kernel_ms"] * 1e-3)), flush=True)
        b.close()
