"""End to end through the Python mirror of the reference's `Sampler` loop: `sample(settings, logp)` for the headline workload —
engine creation, the 500-try init, num_tune + num_draws draws of 4096 chains x dim 1024 delivered to one host trace.  One JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402

dim, chains, tune, draws = 1024, 4096, 400, 100
s = N.DiagNutsSettings(num_chains=chains, seed=20260928, num_tune=tune, num_draws=draws)
t = time.time()
pos, st = N.sample(s, N.LogpSpec.iid_normal(dim, 3.0))
dt = time.time() - t
steps = int(st["n_steps"].sum())
t = time.time()
smp = N.Sampler(s, N.LogpSpec.iid_normal(dim, 3.0), chunk_draws=32)
res = smp.wait_timeout(600.0)
dt_c = time.time() - t
print(json.dumps({"workload": f"sample(): K2 dim {dim} x {chains} chains, {tune} + {draws} draws to one host trace of {pos.nbytes / 1e9:.1f} GB",
                  "sample_seconds": dt, "sample_steps_dims_per_s": steps * dim / dt, "trace_GBps": (pos.nbytes + st.nbytes) / dt / 1e9,
                  "controller_seconds": dt_c, "controller_kind": res.kind,
                  "controller_steps_dims_per_s": int(res.trace["stats"]["n_steps"].sum()) * dim / dt_c,
                  "post_warmup_mean": float(pos[tune:].mean()), "post_warmup_var": float(pos[tune:].var())}))
