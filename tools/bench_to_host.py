"""The PCIe-inclusive rate of the headline workload (K2: iid N(3,1) dim 1024 x 4096 chains): every draw and its statistics
copied to host memory (nm_engine_draw_ex_to_host, the engine's chunked staging path) against the same draws left in device
buffers (what bench.py's `value` measures).  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuts_rs_amd as N  # noqa: E402

dim, chains, draws = 1024, 4096, 100
s = N.DiagNutsSettings(num_chains=chains, seed=20260928, num_tune=400)
b = N.ChainBatch(s, N.LogpSpec.iid_normal(dim, 3.0), chains)
b.set_position(b.init_positions_uniform())
b.draw_device(400)
pos = torch.empty((draws, chains, dim), dtype=torch.float64, device="cuda")
st = torch.empty((draws, chains, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
b.reset_counters()
t = time.time(); b.draw_device(draws, pos.data_ptr(), st.data_ptr()); dt_dev = time.time() - t
steps_dev = b.counters()["total_leapfrogs"]
b.reset_counters()
t = time.time(); p, q = b.draw_many(draws); dt_fresh = time.time() - t           # fresh host arrays: page faults
steps_fresh = int(q["n_steps"].sum())
t = time.time(); b.draw_many(draws, out=(p, q)); dt_host = time.time() - t      # the same arrays again: the PCIe rate
steps_host = int(q["n_steps"].sum())
print(json.dumps({"workload": f"K2 dim {dim} x {chains} chains, {draws} post-warm-up draws",
                  "device_buffers_steps_dims_per_s": steps_dev * dim / dt_dev, "to_host_steps_dims_per_s": steps_host * dim / dt_host,
                  "to_host_over_device": (steps_host / dt_host) / (steps_dev / dt_dev),
                  "to_fresh_host_arrays_steps_dims_per_s": steps_fresh * dim / dt_fresh, "host_bytes": int(p.nbytes + q.nbytes),
                  "host_copy_GBps_incl_compute": (p.nbytes + q.nbytes) / dt_host / 1e9, "mean": float(np.mean(p[-1]))}))
b.close()
