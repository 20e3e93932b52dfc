"""Generate the known answers of nuts_rs_amd.selftest — DATA computed by the CPU oracle (oracle/, pinned to the reference's vectors:
tests/test_oracle_golden.py), generated once and committed; regenerate only when the contract (stream, arithmetic order) changes.

  python tools/gen_selftest_golden.py            # both files
  python tools/gen_selftest_golden.py --only small | instantiations

* nuts_rs_amd/selftest_golden.json: last positions' bits + every draw's n_steps / depth of two small runs that the
  wave, group and lane kernels all serve.
* nuts_rs_amd/selftest_instantiations.json: for EVERY (density, settings family, tiling) of the one-chain-per-block
  kernels, at both ends of the tiling's range of dims (nuts_rs_amd/selftest_cases.py: 328 runs), the SHA-256 of every position and exact
  statistic of a short adaptive run, plus the leapfrog count.  Needs no GPU: the summation order of a wave kernel is fixed by its tiling
  (threads per chain = 64 x wavefronts), and the low-rank estimator the adapting cases use is the library's host twin."""
import ctypes as C
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nuts_rs_amd as N  # noqa: E402
from oracle import oracle as O  # noqa: E402
from nuts_rs_amd.selftest import CASES  # noqa: E402
from nuts_rs_amd import selftest_cases as SC  # noqa: E402


def oracle_settings(settings):
    c = settings.to_c()
    so = O.Settings()
    for f, _ in O.Settings._fields_:
        setattr(so, f, getattr(c, f))
    return so


def small():
    out = {"generator": "tools/gen_selftest_golden.py (oracle/: the CPU restatement of the reference's algorithm)", "cases": {}}
    for name, (mk, chains, seed, draws, threads) in CASES.items():
        logp = mk()
        s = N.DiagNutsSettings(num_chains=chains, seed=seed, num_tune=20, num_draws=draws)
        x0 = O.init_positions_uniform(s.seed, 0, chains, logp.dim)
        pos, st, steps, failed = O.run(oracle_settings(s), logp.kind, logp.dim, logp.params, O.gpu_cfg(threads), chains, x0, draws, n_threads=4)
        assert failed == 0
        out["cases"][name] = {"chains": chains, "dim": logp.dim, "seed": seed, "draws": draws, "threads_per_chain": threads,
                              "last_position_bits": [format(int(v), "016x") for v in pos[-1].reshape(-1).view(np.uint64)],
                              "n_steps": st["n_steps"].astype(int).tolist(), "depth": st["depth"].astype(int).tolist(),
                              "total_leapfrogs": int(steps)}
        print(name, "leapfrogs", steps)
    json.dump(out, open(os.path.join(ROOT, "nuts_rs_amd", "selftest_golden.json"), "w"), indent=0)


def oracle_answer(c):
    """The oracle's run of one case (what tests/test_gpu_every_instantiation.py compares the engine with)."""
    r = SC.make_run(N, c)
    s, logp, transform, draws, n = r["settings"], r["logp"], r["transform"], r["draws"], r["n_chains"]
    x0 = O.init_positions_uniform(s.seed, 0, n, logp.dim)
    adapt = transform == "adapt"
    est = {}
    if adapt:
        from nuts_rs_amd import _lib
        est = dict(estimator=C.cast(_lib.load().nm_lowrank_block_twin, O.ESTIMATOR_FN))
    tf = None if adapt else transform
    cfg = O.gpu_cfg(64 * c["w"], gpu_slice=0, lr_seq_dots=0)
    pos, st, steps, failed = O.run(oracle_settings(s), logp.kind, logp.dim, logp.params, cfg, n, x0, draws, n_threads=1, transform=tf, **est)
    return pos, st, int(steps), int(failed)


def instantiations():
    cs = SC.cases()
    out = {"generator": "tools/gen_selftest_golden.py instantiations (oracle/; cases: nuts_rs_amd/selftest_cases.py)", "cases": {}}

    def one(c):
        pos, st, steps, failed = oracle_answer(c)
        return SC.case_id(c), dict(failed=failed, leapfrogs=steps, sha256=None if failed else SC.digest(pos, st))

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for k, (cid, ans) in enumerate(ex.map(one, cs)):
            out["cases"][cid] = ans
            if k % 20 == 0:
                print(k, cid, ans, flush=True)
    json.dump(out, open(os.path.join(ROOT, "nuts_rs_amd", "selftest_instantiations.json"), "w"), indent=0, sort_keys=True)
    print(len(out["cases"]), "cases,", sum(1 for a in out["cases"].values() if a["failed"]), "with a failed chain (no answer)")


if __name__ == "__main__":
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""
    if only in ("", "small"):
        small()
    if only in ("", "instantiations"):
        instantiations()
