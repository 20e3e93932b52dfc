"""Round 5: the cases of tests/test_density_module.py::test_user_density_with_low_rank_adaptation_trajectory_kinds_and_mclmc on the BUILT-IN diagonal
normal (the module's draws equal the built-in's there; both left the oracle on the r05x library): which family, which draw, which statistic."""
import ctypes as C, os, sys
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nuts_rs_amd as N
from nuts_rs_amd import _lib
from oracle import oracle as O
from helpers import oracle_settings, STAT_FIELDS_EXACT, STAT_FIELDS_FLOAT

DIM = int(sys.argv[1]) if len(sys.argv) > 1 else 40
prec = np.exp(np.random.default_rng(5).uniform(-3, 3, DIM))
n = 4
cases = [("low_rank", N.LowRankNutsSettings(num_chains=n, seed=81, num_tune=120), 150),
         ("exact_normal", N.DiagNutsSettings(num_chains=n, seed=82, num_tune=60, trajectory_kind=N.KineticEnergyKind.EXACT_NORMAL), 90),
         ("microcanonical", N.DiagNutsSettings(num_chains=n, seed=83, num_tune=60, trajectory_kind=N.KineticEnergyKind.MICROCANONICAL), 90),
         ("low_rank_microcanonical", N.LowRankNutsSettings(num_chains=n, seed=84, num_tune=100, trajectory_kind=N.KineticEnergyKind.MICROCANONICAL), 120),
         ("mclmc", N.DiagMclmcSettings(num_chains=n, seed=85, num_tune=60, step_size=0.5, momentum_decoherence_length=3.0), 90),
         ("plain_nuts", N.DiagNutsSettings(num_chains=n, seed=86, num_tune=60), 90)]
for name, s, draws in cases:
    x0 = O.init_positions_uniform(s.seed, 0, n, DIM)
    logp = N.LogpSpec.diag_normal(prec)
    b = N.ChainBatch(s, logp, n)
    b.set_position(x0)
    pos, st = b.draw_many(draws)
    tpc = b.threads_per_chain()
    b.close()
    est = dict(estimator=C.cast(_lib.load().nm_lowrank_block_twin, O.ESTIMATOR_FN)) if "low_rank" in name else {}
    pos_o, st_o, _, failed = O.run(oracle_settings(O, s), logp.kind, DIM, prec, O.gpu_cfg(tpc), n, x0, draws, **est)
    first = None
    for t in range(draws):
        for c in range(n):
            bad = [f for f in STAT_FIELDS_EXACT + STAT_FIELDS_FLOAT if not (st[f][t, c] == st_o[f][t, c] or (st[f].dtype.kind == "f" and np.isnan(st[f][t, c]) and np.isnan(st_o[f][t, c])))]
            if (pos[t, c] != pos_o[t, c]).any():
                bad.append("position")
            if bad and first is None:
                first = (t, c, bad)
    print(name, "dim", DIM, "threads/chain", tpc, "->", "ok" if first is None else f"first difference at draw {first[0]} chain {first[1]}: {first[2]}", flush=True)
    if first:
        t, c, _ = first
        for f in ("depth", "n_steps", "index_in_trajectory", "energy", "logp", "step_size", "mean_tree_accept", "mean_tree_accept_sym"):
            print("   ", f, st[f][t, c], st_o[f][t, c])
