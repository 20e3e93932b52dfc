"""Known-answer self-test of the built library on the GPU.

Two code-generation-sensitive miscompiles were caught in round 4 only because the parity suite happened to run; a user
who builds a density module, or rebuilds the library with another compiler, does not run that suite.  `run()` draws a few small
fixed-seed chains through every kernel family the built-in densities reach — one chain per wavefront, several chains per wavefront,
one chain per lane — and compares the last positions' BITS and every draw's n_steps / depth with known answers computed by the CPU oracle
(nuts_rs_amd/selftest_golden.json: data, generated once by tools/gen_selftest_golden.py).  No oracle code runs here.

Called by nuts_rs_amd.build.build() when a GPU is present, by __graft_entry__.smoke(), and once per process before the first
NM_LOGP_MODULE engine is created (sampler.ChainBatch): a library or runtime that no longer reproduces the answers fails loudly."""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(_HERE, "selftest_golden.json")


def _iid():
    from . import LogpSpec
    return LogpSpec.iid_normal(10, 3.0)


def _schools():
    from . import LogpSpec
    return LogpSpec.eight_schools()


# name -> (density, chains, seed, draws, threads per chain the engine uses: the oracle's summation order)
CASES = {"iid_dim10": (_iid, 4, 42, 30, 64), "eight_schools": (_schools, 8, 43, 30, 64)}
# the kernel families a case is run on: engine keyword arguments
FAMILIES = {"wave": dict(lane_groups=1, lane_chains=1), "group": dict(lane_groups=2, lane_chains=1), "lane": dict(lane_chains=2)}
_done = False


class SelfTestError(RuntimeError):
    pass


def run(verbose=False, device=-1):
    """Raises SelfTestError on the first answer that is not reproduced; returns the number of (case, family) runs checked."""
    global _done
    from . import ChainBatch, DiagNutsSettings
    gold = json.load(open(GOLDEN))["cases"]
    n = 0
    for name, (mk, chains, seed, draws, _threads) in CASES.items():
        g = gold[name]
        want_pos = np.array([int(h, 16) for h in g["last_position_bits"]], dtype=np.uint64).reshape(chains, g["dim"])
        want_steps, want_depth = np.array(g["n_steps"]), np.array(g["depth"])
        for fam, kw in FAMILIES.items():
            s = DiagNutsSettings(num_chains=chains, seed=seed, num_tune=20, num_draws=draws)
            b = ChainBatch(s, mk(), chains, device=device, **kw)
            try:
                if (b.set_position(b.init_positions_uniform()) != 0).any():
                    raise SelfTestError(f"self-test {name} / {fam}: set_position failed")
                pos, st = b.draw_many(draws)
            finally:
                b.close()
            ok = (st["n_steps"] == want_steps).all() and (st["depth"] == want_depth).all() and (pos[-1].view(np.uint64) == want_pos).all()
            if not ok:
                bad = np.argwhere(st["n_steps"] != want_steps)
                where = f"n_steps first differs at (draw, chain) {bad[0].tolist()}" if bad.size else "positions differ"
                raise SelfTestError(f"self-test {name} on the {fam} kernels: the built library does not reproduce the known answers ({where}); "
                                    "this build (compiler, flags, a density module's tuning build) or this runtime is not the one the parity suite verified")
            n += 1
            if verbose:
                print(f"self-test {name} / {fam}: ok ({int(st['n_steps'].sum())} leapfrogs)")
    _done = True
    return n


def run_once(device=-1):
    """The self-test, once per process (before the first density-module engine)."""
    if not _done:
        run(device=device)
