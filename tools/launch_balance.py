"""How evenly a launch's work is spread: blocks stride over chains (chain = block + k * grid), a launch ends with its
slowest block.  Prints sum(n_steps) per block: mean / max = the fraction of the launch the average block is busy."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
import nuts_rs_amd as N

key = sys.argv[1] if len(sys.argv) > 1 else "k3"
logp, C = {"k3": (N.LogpSpec.funnel(101), 8192), "k4": (N.LogpSpec.eight_schools(), 65536)}[key]
s = N.DiagNutsSettings(num_chains=C, seed=20260928, num_tune=400)
b = N.ChainBatch(s, logp, C, lane_groups=1)
b.set_position(b.init_positions_uniform())
b.draw_many(400, positions=False)
b.reset_counters()
_, st = b.draw_many(200, positions=False)
ms = b.counters()["kernel_ms"]
per_chain = st["n_steps"].sum(axis=0).astype(np.float64)
for grid in (2048, 4096):
    if C % grid == 0:
        per_block = per_chain.reshape(C // grid, grid).sum(axis=0)
        print(key, "grid", grid, "mean/max work per block %.3f" % (per_block.mean() / per_block.max()),
              "chain work min/mean/max %.0f %.0f %.0f" % (per_chain.min(), per_chain.mean(), per_chain.max()))
print("kernel_ms %.1f" % ms, "max single draw steps", int(st["n_steps"].max()))
b.close()
