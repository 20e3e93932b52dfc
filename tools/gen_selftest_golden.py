"""Generate nuts_rs_amd/selftest_golden.json — the known answers of nuts_rs_amd.selftest: the last positions'
bits and every draw's n_steps / depth of small fixed-seed runs, computed by the CPU oracle (oracle/, pinned to the reference's vectors:
tests/test_oracle_golden.py).  DATA, generated once and committed; regenerate only when the contract (stream, arithmetic order) changes.

  python tools/gen_selftest_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nuts_rs_amd as N  # noqa: E402
from oracle import oracle as O  # noqa: E402
from nuts_rs_amd.selftest import CASES  # noqa: E402


def main():
    out = {"generator": "tools/gen_selftest_golden.py (oracle/: the CPU restatement of the reference's algorithm)", "cases": {}}
    for name, (mk, chains, seed, draws, threads) in CASES.items():
        logp = mk()
        s = N.DiagNutsSettings(num_chains=chains, seed=seed, num_tune=20, num_draws=draws)
        so = O.Settings()
        c = s.to_c()
        for f, _ in O.Settings._fields_:
            setattr(so, f, getattr(c, f))
        x0 = O.init_positions_uniform(s.seed, 0, chains, logp.dim)
        pos, st, steps, failed = O.run(so, logp.kind, logp.dim, logp.params, O.gpu_cfg(threads), chains, x0, draws, n_threads=4)
        assert failed == 0
        out["cases"][name] = {"chains": chains, "dim": logp.dim, "seed": seed, "draws": draws, "threads_per_chain": threads,
                              "last_position_bits": [format(int(v), "016x") for v in pos[-1].reshape(-1).view(np.uint64)],
                              "n_steps": st["n_steps"].astype(int).tolist(), "depth": st["depth"].astype(int).tolist(),
                              "total_leapfrogs": int(steps)}
        print(name, "leapfrogs", steps)
    json.dump(out, open(os.path.join(ROOT, "nuts_rs_amd", "selftest_golden.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
