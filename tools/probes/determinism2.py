"""development: two runs of the same engine, compared slot by slot after every draw"""
import sys, os, ctypes as C
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import nuts_rs_amd as N
n, dim, draws = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = N.DiagNutsSettings(num_chains=n, seed=20260928, num_tune=400)
logp = N.LogpSpec.iid_normal(dim, 3.0)
L = N.load_library()
names = "X GX Z GZ SIG ISIG MU E_DM E_DV E_GM E_GV B_DM B_DV B_GM B_GV V".split()
def slots(b):
    out = []
    for k in range(16):
        a = np.empty((n, dim)); assert L.nm_debug_read_slot(b._h, k, a.ctypes.data_as(C.c_void_p)) == 0; out.append(a)
    return out
shown = {}
bs = [N.ChainBatch(s, logp, n, chain_tiles=1, lane_groups=1) for _ in range(2)]
for b in bs:
    b.set_position(b.init_positions_uniform())
for t in range(draws + 1):
    sa, sb = slots(bs[0]), slots(bs[1])
    rep = []
    for k in range(16):
        d = (sa[k] != sb[k]) & ~(np.isnan(sa[k]) & np.isnan(sb[k]))
        if d.any():
            ch = np.nonzero(d.any(axis=1))[0]
            rel = np.abs(sa[k][d] - sb[k][d]) / np.maximum(np.abs(sa[k][d]), 1e-300)
            rep.append(f"{names[k]}: {len(ch)} chains (min {ch.min()}), {int(d.sum())} elems, rel diff max {rel.max():.2e}")
            if not shown.get(k):
                shown[k] = 1
                c0 = ch.min(); idx = np.nonzero(d[c0])[0]
                print("   ", names[k], "chain", c0, "elements", idx.tolist())
                for i in idx[:6]:
                    print("      %d: %016x %016x   %r %r" % (i, sa[k][c0, i].view(np.uint64), sb[k][c0, i].view(np.uint64), sa[k][c0, i], sb[k][c0, i]))
    print("after", t, "draws:", rep if rep else "identical", flush=True)
    if t < draws:
        for b in bs:
            b.draw_many(1)
