"""Round 5: the iid dim 129 LowRankNutsSettings mismatch of fuzz seed 551 (cases 54 / 109), reduced: first draws of engine and oracle side by side."""
import os, sys
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nuts_rs_amd as N
from oracle import oracle as O
from helpers import oracle_settings

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 129
lowrank = (sys.argv[2] if len(sys.argv) > 2 else "lr") == "lr"
dpl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n, draws = 3, 6
md = int(sys.argv[4]) if len(sys.argv) > 4 else 10
kw = dict(num_chains=n, seed=1234, num_tune=100, maxdepth=md)
s = N.LowRankNutsSettings(**kw) if lowrank else N.DiagNutsSettings(**kw)
if len(sys.argv) > 5 and sys.argv[5] == "nojit":
    s.adapt_options.step_size_settings.jitter = None
logp = N.LogpSpec.iid_normal(dim, 0.3)
x0 = O.init_positions_uniform(s.seed, 0, n, dim)
eng = dict(dims_per_lane=dpl, waves_per_chain=1) if dpl else {}
b = N.ChainBatch(s, logp, n, **eng)
st0 = b.set_position(x0, raise_on_error=False)
if lowrank:
    b.set_lowrank_estimator_place("device")
pos, st = b.draw_many(draws, raise_on_error=False)
tpc, k, order = b.threads_per_chain(), b.blocks_per_chain(), b.reduce_order()
b.close()
cfg = O.gpu_cfg(tpc, gpu_slice=0, lr_seq_dots=0)
est = {}
if lowrank:
    import ctypes as C
    from nuts_rs_amd import _lib
    est = dict(estimator=C.cast(_lib.load().nm_lowrank_block_twin, O.ESTIMATOR_FN))
pos_o, st_o, _, failed = O.run(oracle_settings(O, s), logp.kind, logp.dim, logp.params, cfg, n, x0, draws, n_threads=1, **est)
print("lib", os.environ.get("NUTS_AMD_LIB", "default"), "dim", dim, "lowrank", lowrank, "threads/chain", tpc, "maxdepth", md)
for t in range(draws):
    for c in range(n):
        d = np.abs(pos[t, c] - pos_o[t, c]).max()
        print(f"draw {t} chain {c}: max|dpos| {d:.3e}  depth {st['depth'][t, c]} / {st_o['depth'][t, c]}  n_steps {st['n_steps'][t, c]} / {st_o['n_steps'][t, c]}"
              f"  sym {st['mean_tree_accept_sym'][t, c]!r} / {st_o['mean_tree_accept_sym'][t, c]!r}  bar {st['step_size_bar'][t, c]!r} / {st_o['step_size_bar'][t, c]!r}  maxe {st['max_energy_error'][t, c]!r} / {st_o['max_energy_error'][t, c]!r}"
              f"  idx {st['index_in_trajectory'][t, c]} / {st_o['index_in_trajectory'][t, c]}  acc {st['mean_tree_accept'][t, c]!r} / {st_o['mean_tree_accept'][t, c]!r}  eerr {st['energy_error'][t, c]!r} / {st_o['energy_error'][t, c]!r}"
              f"  energy {st['energy'][t, c]!r} / {st_o['energy'][t, c]!r}  logp {st['logp'][t, c]!r} / {st_o['logp'][t, c]!r}  step {st['step_size'][t, c]!r} / {st_o['step_size'][t, c]!r}")
