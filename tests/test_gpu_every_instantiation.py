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

TILINGS = [(2, 1), (4, 1), (8, 1), (16, 1), (8, 2), (16, 2), (4, 4), (16, 4)]
FAMILIES = ["nuts", "lr_frozen", "lr_adapt", "exact", "micro", "mclmc", "lr_mclmc"]


def _cases():
    out = []
    for dens in ("iid", "diag", "funnel", "mvn", "schools"):
        for (d, w) in TILINGS:
            cap = d * 64 * w
            if dens == "schools" and (d, w) != (2, 1):
                continue
            if dens == "mvn" and cap > 2048:
                continue
            for fam in FAMILIES:
                if fam in ("lr_adapt", "lr_mclmc") and cap > 256:   # the device estimator's range (the oracle runs its twin); beyond it: test_gpu_lowrank, test_gpu_mclmc
                    continue
                # full tiles on every second tiling for the densities whose kernels have a full-tile path, ragged tiles otherwise
                full = (TILINGS.index((d, w)) % 2 == 1) and dens in ("iid", "diag", "mvn")
                dim = 10 if dens == "schools" else (cap if full else cap - 3)
                if dens == "mvn":
                    dim = min(dim, 700)                      # a dense precision matrix: O(dim^2) per leapfrog in the oracle
                out.append(pytest.param(dens, fam, d, w, dim, id=f"{dens}-{fam}-{d}x{w}-dim{dim}"))
    return out


def _logp(dens, dim, seed):
    r = np.random.default_rng(seed)
    if dens == "iid":
        return N.LogpSpec.iid_normal(dim, float(r.normal()))
    if dens == "diag":
        return N.LogpSpec.diag_normal(np.exp(r.uniform(-2, 2, dim)))
    if dens == "funnel":
        return N.LogpSpec.funnel(dim)
    if dens == "schools":
        return N.LogpSpec.eight_schools()
    a = r.normal(size=(dim, 8))
    p = a @ a.T / 8 + np.eye(dim)
    return N.LogpSpec.mvn_precision((p + p.T) / 2)


@pytest.mark.parametrize("dens,fam,dpl,wpc,dim", _cases())
def test_instantiation_bit_exact(oracle, dens, fam, dpl, wpc, dim):
    O = oracle
    n, seed = 3, 1000 + 17 * dpl + wpc
    num_tune = 40 if dim > 600 else 70
    draws = num_tune + 10
    kw = dict(num_chains=n, seed=seed, num_tune=num_tune)
    transform = None
    if fam in ("mclmc", "lr_mclmc"):
        mk = N.DiagMclmcSettings if fam == "mclmc" else N.LowRankMclmcSettings
        if fam == "lr_mclmc":
            kw["num_tune"] = num_tune = 100
            draws = 110
            transform = "adapt"
        s = mk(step_size=0.4, momentum_decoherence_length=3.0, trajectory_kind=1, dynamic_step_size=True, subsample_frequency=0.5, **kw)
    elif fam in ("lr_frozen", "lr_adapt"):
        if fam == "lr_adapt":
            kw["num_tune"] = num_tune = 100
            draws = 110
        s = N.LowRankNutsSettings(freeze_transform=(fam == "lr_frozen"), maxdepth=6, **kw)
        if fam == "lr_adapt":
            s.adapt_options.mass_matrix_update_freq = 5
            transform = "adapt"
        else:
            r = np.random.default_rng(seed + 1)
            rank = min(dim, 5)
            vecs = np.linalg.qr(r.normal(size=(dim, rank)))[0].T[:rank]
            transform = (np.exp(r.normal(0, 0.3, dim)), r.normal(0, 1, dim), np.exp(r.uniform(-1, 2, rank)), np.ascontiguousarray(vecs), r.normal(0, 0.3, dim))
    else:
        s = N.DiagNutsSettings(maxdepth=6, trajectory_kind={"nuts": 0, "exact": 1, "micro": 2}[fam], **kw)
    logp = _logp(dens, dim, seed)
    x0 = O.init_positions_uniform(s.seed, 0, n, logp.dim)
    eng = {} if dens == "schools" else dict(dims_per_lane=dpl, waves_per_chain=wpc)
    b = N.ChainBatch(s, logp, n, lane_groups=1, lane_chains=1, chain_tiles=1, **eng)     # the one-chain-per-block kernels, nothing else
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
