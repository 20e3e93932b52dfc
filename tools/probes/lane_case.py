#!/usr/bin/env python
"""Debug: one case of tests/test_gpu_lane_chains.py::test_lane_kernel_sweep, stats of the GPU and the oracle side by side around the first
difference.  usage: python tools/probes/lane_case.py <case index> [--nosplit]"""
import sys, os
import numpy as np
R = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import nuts_rs_amd as N
from oracle import oracle as O
from helpers import run_engine, run_oracle
import test_gpu_lane_chains as T
O.lib()
want = int(sys.argv[1])
rng = np.random.default_rng(177)
for i in range(want + 1):
    dens, dim, kw, logp = T._case(rng, i)
    n_chains = int(rng.choice([1, 3, 17, 64, 65, 100, 200]))
    s = N.DiagNutsSettings(num_chains=n_chains, **kw)
    grid = int(rng.integers(1, 3)) if rng.random() < 0.3 else 0
x0 = O.init_positions_uniform(s.seed, 0, n_chains, dim)
n_draws = s.num_tune + 40
splits = () if "--nosplit" in sys.argv else (s.num_tune, s.num_tune + 1, s.num_tune + 17)
if os.environ.get("SPLITS"): splits = tuple(int(x) for x in os.environ["SPLITS"].split(","))
pos_g, st_g, ex = run_engine(s, logp, n_chains, x0, n_draws, lane_chains=2, grid_blocks=grid, splits=splits)
pos_o, st_o, steps, failed = run_oracle(O, s, logp, n_chains, x0, n_draws, gpu_threads=64)
print("case", want, dens, dim, n_chains, "tune", s.num_tune, "splits", splits, "launches", ex["lane_launches"])
for t in range(n_draws):
    for c in range(n_chains):
        d = [f for f in ("depth", "n_steps", "step_size", "energy", "logp", "mean_tree_accept", "index_in_trajectory") if st_g[f][t, c] != st_o[f][t, c] and not (st_g[f][t, c] != st_g[f][t, c])]
        if d:
            print("first difference at draw", t, "chain", c, d)
            for f in ("depth", "n_steps", "step_size", "energy", "logp", "mean_tree_accept", "index_in_trajectory", "diverging", "tuning"):
                print("  ", f, "gpu", st_g[f][t, c], "oracle", st_o[f][t, c], "| prev gpu", st_g[f][t - 1, c], "oracle", st_o[f][t - 1, c])
            sys.exit(0)
print("no difference")
