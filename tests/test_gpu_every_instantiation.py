"""Every kernel instantiation of the one-chain-per-block family is run against the oracle at least once.

Round 5 found a build in which ONE instantiation — `nuts_draw_kernel<4, 1, LrWrap<IidNormal>>` — computed wrong trajectories while
its 200 siblings were right, and no test held that (density, settings family, tiling) triple: the suite covered every density, every
family and every tiling, but not their product ("fourth incident").  A miscompiled translation unit is a per-instantiation
event, so the guard has to be per instantiation:

    density  x  {DiagNutsSettings, LowRankNutsSettings frozen / adapting, ExactNormal, Microcanonical, MCLMC, LowRank MCLMC}
             x  the eight tilings (doubles per lane, wavefronts per chain)

— a few chains, a short adaptive run each, positions and every statistic bit for bit (reference: src/chain.rs:150-243 is the draw every
one of these kernels restates; src/sampler.rs:199-245, :266-384 the settings families)."""
import ctypes as C

import numpy as np
import pytest

import nuts_rs_amd as N
from helpers import STAT_FIELDS_EXACT, oracle_settings

pytestmark = pytest.mark.gpu

from nuts_rs_amd import selftest_cases as SC

# (the cases live in the package — nuts_rs_amd/selftest_cases.py — because the library's self-test runs the same ones against the oracle's
# answers as data: both ends of every tiling's range of dims since round 6)


def _cases():
    return [pytest.param(c, id=SC.case_id(c)) for c in SC.cases()]


@pytest.mark.parametrize("case", _cases())
def test_instantiation_bit_exact(oracle, case):
    O = oracle
    dens, fam, dpl, wpc, dim = case["dens"], case["fam"], case["dpl"], case["w"], case["dim"]
    r = SC.make_run(N, case)
    s, logp, transform, draws, n = r["settings"], r["logp"], r["transform"], r["draws"], r["n_chains"]
    x0 = O.init_positions_uniform(s.seed, 0, n, logp.dim)
    b = N.ChainBatch(s, logp, n, **r["engine"])     # the one-chain-per-block kernels, nothing else
    assert (b.dims_per_lane(), b.threads_per_chain() // 64) == ((2, 1) if dens == "schools" else (dpl, wpc))
    status = b.set_position(x0, raise_on_error=False)
    adapt = transform == "adapt"
    if adapt:
        transform = None
        b.set_lowrank_estimator_place("device")
    if transform is not None and (status == 0).all():
        b.set_transform(*transform)
    if (status == 0).all():
        cut = draws // 2
        pa, sa = b.draw_many(cut, raise_on_error=False)
        pb, sb = b.draw_many(draws - cut, raise_on_error=False)
        pos, st = np.concatenate([pa, pb]), np.concatenate([sa, sb])
    tpc, k, order = b.threads_per_chain(), b.blocks_per_chain(), b.reduce_order()
    b.close()
    cfg = O.gpu_cfg(tpc, gpu_slice=0, lr_seq_dots=order if transform is not None else 0)
    est = {}
    if adapt:
        from nuts_rs_amd import _lib
        est = dict(estimator=C.cast(_lib.load().nm_lowrank_block_twin, O.ESTIMATOR_FN))
    pos_o, st_o, _, failed = O.run(oracle_settings(O, s), logp.kind, logp.dim, logp.params, cfg, n, x0, draws, n_threads=8, transform=transform, **est)
    if not (status == 0).all():
        assert failed, "the engine refused an initial point the oracle accepts"
        pytest.skip("initial point rejected by both")
    if failed:
        assert not (st["chain_status"] == 0).all(), "an oracle chain failed, the engine's did not"
        return
    bad = np.argwhere((pos.view(np.uint64) != pos_o.view(np.uint64)).any(axis=2))
    assert bad.size == 0, f"positions differ first at (draw, chain) = {bad[0]}"
    for f in list(STAT_FIELDS_EXACT) + ["step_size", "energy", "logp", "mean_tree_accept", "energy_error"]:
        a, bb = st[f], st_o[f]
        same = (a == bb) | (np.isnan(a.astype(float)) & np.isnan(bb.astype(float))) if a.dtype.kind == "f" else (a == bb)
        assert same.all(), f"stat {f} differs first at (draw, chain) = {np.argwhere(~same)[0]}"
