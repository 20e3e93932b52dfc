#!/usr/bin/env python
"""Wave-level phase timeline of the one-chain-per-lane kernel (block 0), from a -DNM_LANE_PROF=1 build:
  python tools/build_variant.py lprof "-DNM_LANE_PROF=1" kern_lane.hip && NUTS_AMD_LIB=nuts_rs_amd/libnuts_amd_lprof.so python tools/prof_lane.py
Prints one JSON line: s_memtime ticks (100 MHz) per phase and per draw for the post-warm-up draws of the K4 model."""
import argparse, ctypes as Ct, json, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import nuts_rs_amd as N
from nuts_rs_amd import _lib

NAMES = ["begin(normals,z,g,ke)", "doubling head + ld_edge", "leapfrog", "account", "turning tests", "merges", "pair stores",
         "top-level tests + merge + edge store", "draw end: x/gx + P stores", "outputs + fd", "adapt", "stats row"]

def read(b):
    L = _lib.load()
    L.nm_debug_read_prof.argtypes = [Ct.c_void_p, Ct.c_void_p]
    buf = np.zeros(32, np.uint64)
    L.nm_debug_read_prof(b._h, buf.ctypes.data)
    return buf

ap = argparse.ArgumentParser()
ap.add_argument("--logp", default="schools"); ap.add_argument("--chains", type=int, default=65536)
ap.add_argument("--tune", type=int, default=400, help="warm-up draws (>= 20)"); ap.add_argument("--draws", type=int, default=200)
ap.add_argument("--maxdepth", type=int, default=10)
ap.add_argument("--fixed-step", type=float, default=0.0, help="> 0: no adaptation, this step size (deep trees with a small one)")
a = ap.parse_args()
logp = N.LogpSpec.eight_schools() if a.logp == "schools" else N.LogpSpec.iid_normal(10, 3.0)
s = N.DiagNutsSettings(num_chains=a.chains, seed=20260928, num_tune=a.tune, num_draws=a.draws, maxdepth=a.maxdepth)
if a.fixed_step > 0:
    a.tune = 1
    s = N.DiagNutsSettings(num_chains=a.chains, seed=11, num_tune=1, num_draws=a.draws, maxdepth=a.maxdepth)
    st_ = s.adapt_options.step_size_settings
    st_.method, st_.fixed_step_size, st_.jitter = N.sampler.STEP_FIXED, a.fixed_step, None
b = N.ChainBatch(s, logp, a.chains, lane_chains=2)
b.set_position(b.init_positions_uniform())
out = {}
for tag, n in (("warmup", a.tune), ("sampling", a.draws)):
    if n == 0: continue
    _, st = b.draw_many(n, positions=False)
    p = read(b).astype(np.float64)
    tot = p[:12].sum()
    out[tag] = {"draws": n, "ticks_per_draw": tot / n, "us_per_draw_at_100MHz": tot / n / 100.0,
                "mean_steps_per_draw_block0": float(st["n_steps"][:, :64].mean()), "max_steps_per_draw_block0": float(st["n_steps"][:, :64].max(axis=1).mean()),
                "phases": {NAMES[i]: {"share": round(p[i] / tot, 4), "ticks_per_draw": round(p[i] / n, 1), "marks_per_draw": round(p[16 + i] / n, 2)} for i in range(12)}}
print(json.dumps(out))
