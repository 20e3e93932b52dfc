import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nuts_rs_amd as N
C_, total = int(sys.argv[1]), int(sys.argv[2])
s = N.DiagNutsSettings(num_chains=C_, seed=20260928, num_tune=400, num_draws=50)
b = N.ChainBatch(s, N.LogpSpec.eight_schools(), C_)
b.init_with_retries()
st = torch.zeros((total, C_, N.STATS_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
import time
t0 = time.time(); b.draw_device(total, 0, st.data_ptr()); t1 = time.time()
print("draw_device returned after %.3f s" % (t1 - t0), "counters", b.counters())
if len(sys.argv) > 3:
    time.sleep(float(sys.argv[3])); torch.cuda.synchronize()
h = np.frombuffer(st.cpu().numpy().tobytes(), dtype=N.STATS_DTYPE).reshape(total, C_)
written = (h["n_steps"] > 0)
print("group launches", b.group_launches(), "rows written per draw (first 5, around 59, last):", written.sum(1)[:3], written.sum(1)[55:63], written.sum(1)[-3:])
print("chains fully written:", int(written.all(0).sum()), "of", C_, " statuses:", np.unique(h["chain_status"]), "draw idx ok:", bool((h["draw"][written] == np.nonzero(written)[0]).all()))
bad = np.nonzero(~written.all(0))[0]
print("first bad chains", bad[:10], "their written counts", written[:, bad[:10]].sum(0))
for c in (0, 9, 40000):
    w = np.nonzero(written[:, c])[0]
    print("chain", c, "written rows:", w[:5], "...", w[-5:], "draw field there:", h["draw"][w[:5], c], h["draw"][w[-5:], c], "chain field", h["chain"][w[:3], c])
    gaps = np.nonzero(np.diff(w) > 1)[0]
    print("   gaps after rows", w[gaps][:10], "count", len(gaps))
