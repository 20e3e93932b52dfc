"""Known-answer self-test of the built library on the GPU.

Two code-generation-sensitive miscompiles were caught in round 4 only because the parity suite happened to run; a user
who builds a density module, or rebuilds the library with another compiler, does not run that suite.  `run()` draws a few small
fixed-seed chains through every kernel family the built-in densities reach — one chain per wavefront, several chains per wavefront,
one chain per lane — and compares the last positions' BITS and every draw's n_steps / depth with known answers computed by the CPU oracle
(nuts_rs_amd/selftest_golden.json: data, generated once by tools/gen_selftest_golden.py).  No oracle code runs here.

Called by nuts_rs_amd.build.build() when a GPU is present, by __graft_entry__.smoke(), and once per process before the first
NM_LOGP_MODULE engine is created (sampler.ChainBatch): a library or runtime that no longer reproduces the answers fails loudly.

Round 6: EVERY instantiation of the one-chain-per-block kernels has its own known answer
(selftest_instantiations.json: 328 runs — density x settings family x tiling x both ends of the tiling's dims —, the oracle's SHA-256 of
every position and exact statistic; cases in selftest_cases.py).  `run_all()` checks all of them (build() on a GPU box; ~1 min);
`first_use(batch)` checks the runs of the ONE instantiation an engine is about to launch, the first time a process creates such an engine
(~0.1 - 0.3 s; sampler.ChainBatch calls it; NUTS_AMD_SELFTEST=0 turns it off).  The wrong-results incidents of rounds 4 - 5
were all single instantiations that a compiler defect had hit while their siblings were right: the check is per instantiation.
What this does NOT cover: the code of a user's density module (there is no oracle for a functor this repo has never seen) — a module is
protected by the assembly scans of its build (nuts_rs_amd.build: tools/check_exec_spill.py names and repairs the defect itself)."""
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


# ---------------------------------------------------------------------------------------------
# every instantiation (round 6)
# ---------------------------------------------------------------------------------------------
GOLDEN_INST = os.path.join(_HERE, "selftest_instantiations.json")
_inst_gold = None
_checked = set()           # (density, family, doubles per lane, wavefronts) verified in this process
_busy = False              # a self-test run creates engines itself


def _gold_inst():
    global _inst_gold
    if _inst_gold is None:
        _inst_gold = json.load(open(GOLDEN_INST))["cases"]
    return _inst_gold


def run_case(c, device=-1):
    """One known-answer run on the engine -> (positions, statistics) or None when an initial point is rejected."""
    import nuts_rs_amd as N
    from . import selftest_cases as SC
    global _busy
    r = SC.make_run(N, c)
    s, logp, transform, draws, n = r["settings"], r["logp"], r["transform"], r["draws"], r["n_chains"]
    was, _busy = _busy, True
    try:
        b = N.ChainBatch(s, logp, n, device=device, **r["engine"])
        try:
            x0 = b.init_positions_uniform()
            status = b.set_position(x0, raise_on_error=False)
            if not (status == 0).all():
                return None
            if transform == "adapt":
                b.set_lowrank_estimator_place("device")
            elif transform is not None:
                b.set_transform(*transform)
            cut = draws // 2
            pa, sa = b.draw_many(cut, raise_on_error=False)
            pb, sb = b.draw_many(draws - cut, raise_on_error=False)
            return np.concatenate([pa, pb]), np.concatenate([sa, sb])
        finally:
            b.close()
    finally:
        _busy = was


def check_case(c, device=-1):
    """-> None if the engine reproduces the case's known answer, else a message."""
    from . import selftest_cases as SC
    cid = SC.case_id(c)
    g = _gold_inst().get(cid)
    if g is None:
        return f"{cid}: no known answer on file (regenerate selftest_instantiations.json: tools/gen_selftest_golden.py)"
    out = run_case(c, device)
    if g["failed"]:
        return None if (out is None or not (out[1]["chain_status"] == 0).all()) else f"{cid}: the oracle's chains fail, the engine's do not"
    if out is None:
        return f"{cid}: the engine rejected an initial point the oracle accepts"
    pos, st = out
    if SC.digest(pos, st) != g["sha256"]:
        return f"{cid}: the draws differ from the known answer ({int(st['n_steps'].sum())} leapfrogs against {g['leapfrogs']})"
    return None


def run_all(verbose=False, device=-1, only=None):
    """Every instantiation against its known answer; raises SelfTestError naming every instantiation that fails.  Returns the number of runs."""
    from . import selftest_cases as SC
    bad = []
    n = 0
    for c in SC.cases():
        if only and only not in SC.case_id(c):
            continue
        msg = check_case(c, device)
        n += 1
        if msg:
            bad.append(msg)
        elif verbose:
            print("self-test", SC.case_id(c), "ok")
        _checked.add((c["dens"], c["fam"], c["dpl"], c["w"]))
    if bad:
        raise SelfTestError(f"{len(bad)} of {n} kernel instantiations of the built library do not reproduce their known answers:\n  " + "\n  ".join(bad[:40]) +
                            "\nthis build (compiler, flags, headers) is not the one the parity suite verified; see DESIGN §22")
    return n


def first_use(batch, device=-1):
    """Called by sampler.ChainBatch after nm_engine_create: the known answers of the instantiation this engine launches, once per process."""
    from . import selftest_cases as SC
    if _busy or os.environ.get("NUTS_AMD_SELFTEST", "1") == "0":
        return 0
    dens = SC.KIND_NAME.get(int(batch.logp.kind))
    if dens is None or batch.blocks_per_chain() != 1:
        return 0
    dpl, w = batch.dims_per_lane(), batch.threads_per_chain() // 64
    n = 0
    for fam in SC.family_of(batch.settings):
        key = (dens, fam, dpl, w)
        if key in _checked:
            continue
        _checked.add(key)
        for c in SC.cases():
            if (c["dens"], c["fam"], c["dpl"], c["w"]) == key:
                msg = check_case(c, device)
                n += 1
                if msg:
                    raise SelfTestError("the kernel instantiation this engine is about to launch does not reproduce its known answer: " + msg +
                                        "; the library was built by a compiler / from headers other than the verified ones (DESIGN §22), or the runtime differs")
    return n


if __name__ == "__main__":      # python -m nuts_rs_amd.selftest [--all] [-v]: for hosts that do not go through sampler.ChainBatch (C / C++ / Rust over the C ABI)
    import sys
    os.environ["NUTS_AMD_SELFTEST"] = "0"          # (run_all checks everything itself)
    n = run(verbose="-v" in sys.argv)
    print(f"{n} small runs on the wave / group / lane kernels reproduce their known answers")
    if "--all" in sys.argv:
        print(f"{run_all(verbose='-v' in sys.argv)} kernel instantiations reproduce their known answers")
