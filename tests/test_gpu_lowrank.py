"""GPU parity of the low-rank transformation (reference src/transform/low_rank.rs,
src/transform/adapt/low_rank.rs, src/math/cpu_math.rs:332-425) against the CPU oracle, through the C ABI.

  * the three maps of LowRankMassMatrix on device vectors (nm_lowrank_transform_batch)            — bit-exact
  * whole chains with a FIXED transformation (nm_engine_set_transform; per chain and shared; rank << dim and rank = dim,
    BASELINE config 5's "dense" case at dim 256)                                                   — bit-exact
  * whole chains with LowRankNutsSettings' ADAPTATION, the estimator's dense linear algebra injected identically on both
    sides (the oracle's LAPACK restatement), so that everything else — windows, schedule, pause / resume protocol,
    re-whitening, step-size search after the first update, statistics — must agree                 — bit-exact
  * the built-in estimator: its device form (one block per paused chain, what a default run executes) bit for bit against its
    host twin, and a whole adaptive warm-up bit-exact against the oracle running that twin; the host form end to end
    (statistical, and window by window against the literal reference algorithm); the full-size K5 run (properties)
"""
import ctypes as C

import numpy as np
import pytest

import nuts_rs_amd as N
from nuts_rs_amd import _lib
from helpers import assert_bit_exact, oracle_settings

pytestmark = pytest.mark.gpu


def random_transform(rng, dim, rank, per_chain=0):
    shape = (per_chain,) if per_chain else ()
    stds = np.exp(rng.normal(0, 0.5, shape + (dim,)))
    mean = rng.normal(0, 1, shape + (dim,))
    mu = rng.normal(0, 0.3, shape + (dim,))
    vals = np.exp(rng.uniform(-2, 3, shape + (rank,)))
    if per_chain:
        vecs = np.stack([np.linalg.qr(rng.normal(size=(dim, rank)))[0].T for _ in range(per_chain)])
    else:
        vecs = np.linalg.qr(rng.normal(size=(dim, rank)))[0].T
    return stds, mean, vals, np.ascontiguousarray(vecs), mu


def correlated_precision(rng, dim, rank, scale=50.0):
    u = np.linalg.qr(rng.normal(size=(dim, rank)))[0]
    sigma = np.eye(dim) + u @ np.diag(rng.uniform(5.0, scale, rank)) @ u.T
    sc = np.exp(rng.normal(0, 0.5, dim))
    sigma = np.diag(sc) @ sigma @ np.diag(sc)
    prec = np.linalg.inv(sigma)
    return (prec + prec.T) / 2, sigma


@pytest.mark.parametrize("dim,rank", [(3, 1), (10, 4), (40, 40), (100, 7), (256, 16), (256, 256), (1000, 5)])
def test_lowrank_maps_bit_exact(oracle, dim, rank):
    import torch
    rng = np.random.default_rng(dim * 1000 + rank)
    n = 3
    stds, mean, vals, vecs, mu = random_transform(rng, dim, rank, per_chain=n)
    vin = rng.normal(size=(n, dim))
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    t = [dev(a) for a in (stds, mean, vals, vecs, mu, vin)]
    out = torch.empty((n, dim), dtype=torch.float64, device="cuda")
    cfg = oracle.gpu_cfg(64)
    for which in (0, 1, 2):
        _lib.check(_lib.load().nm_lowrank_transform_batch(which, n, dim, rank, 0, *[x.data_ptr() for x in t], out.data_ptr(), None))
        got = out.cpu().numpy()
        for i in range(n):
            want = oracle.lowrank_kat(cfg, np.ones(dim), stds[i], mean[i], vals[i], vecs[i], mu[i], vin[i], which=which)["z"]
            bad = np.argwhere(got[i].view(np.uint64) != want.view(np.uint64))
            assert bad.size == 0, f"map {which} chain {i}: first difference at element {bad[0]}: {got[i][bad[0][0]]!r} vs {want[bad[0][0]]!r}"


def lowrank_settings(**kw):
    return N.LowRankNutsSettings(**kw)


def run_fixed(oracle, logp, settings, n_chains, transform, n_draws, waves_per_chain=0, dims_per_lane=0, chain_tiles=0,
              expect_tiles=None, splits=()):
    x0 = oracle.init_positions_uniform(settings.seed, 0, n_chains, logp.dim)
    b = N.ChainBatch(settings, logp, n_chains, waves_per_chain=waves_per_chain, dims_per_lane=dims_per_lane, chain_tiles=chain_tiles)
    assert (b.set_position(x0) == 0).all()
    b.set_transform(*transform)
    cuts = [0] + [c for c in splits if 0 < c < n_draws] + [n_draws]
    parts = [b.draw_many(hi - lo) for lo, hi in zip(cuts[:-1], cuts[1:])]
    pos, st = np.concatenate([p for p, _ in parts]), np.concatenate([q for _, q in parts])
    tpc = b.threads_per_chain()
    tiles = b.tile_launches() > 0              # the matrix-core kernels sum U'v sequentially: the oracle's lr_seq_dots mode
    order = b.reduce_order()                   # (1: the tile kernel, 2: the lockstep kernel with its own order of the sums over dim)
    if expect_tiles is not None:
        assert tiles == expect_tiles
    assert (order > 0) == tiles and (order == 2) == (b.lockstep_launches() > 0)
    b.close()
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, settings), logp.kind, logp.dim, logp.params,
                                        oracle.gpu_cfg(tpc, lr_seq_dots=order), n_chains, x0, n_draws, n_threads=8, transform=transform)
    assert failed == 0
    return pos, st, pos_o, st_o


@pytest.mark.parametrize("case", ["dim10_rank3_shared", "dim100_rank6_per_chain", "dim40_full_rank", "dim300_rank12_two_waves",
                                  "dim256_rank256_mvn", "dim256_rank256_mvn_one_chain_per_block"])
def test_fixed_transform_chains_bit_exact(oracle, case):
    rng = np.random.default_rng(abs(hash(case)) % 2**31)
    kw = {}
    if case == "dim10_rank3_shared":
        dim, n, tr, logp, draws = 10, 5, random_transform(rng, 10, 3), None, 120
    elif case == "dim100_rank6_per_chain":
        dim, n, tr, logp, draws = 100, 4, random_transform(rng, 100, 6, per_chain=4), None, 90
    elif case == "dim40_full_rank":
        dim, n, tr, logp, draws = 40, 3, random_transform(rng, 40, 40), None, 90
    elif case == "dim300_rank12_two_waves":
        dim, n, tr, logp, draws = 300, 3, random_transform(rng, 300, 12), None, 60
        kw = dict(waves_per_chain=2, dims_per_lane=8)
    else:   # BASELINE config 5: N(0, Sigma) with full Sigma at dim 256, the exact "dense" preconditioner as a rank-256 transform
        dim, n, draws = 256, 3, 40
        prec, sigma = correlated_precision(rng, dim, 6)
        w, u = np.linalg.eigh(sigma)
        tr = (np.ones(dim), np.zeros(dim), w, np.ascontiguousarray(u.T), np.zeros(dim))
        logp = N.LogpSpec.mvn_precision(prec)
    if logp is None:
        logp = N.LogpSpec.diag_normal(np.exp(rng.uniform(-2, 2, dim)))
    s = lowrank_settings(num_chains=n, seed=31, num_tune=draws - 20, freeze_transform=True)
    if case == "dim256_rank256_mvn":
        kw = dict(expect_tiles=True)                     # shared + frozen + full-precision normal: the matrix-core kernel
    elif case.endswith("one_chain_per_block"):
        kw = dict(chain_tiles=1, expect_tiles=False)
    pos, st, pos_o, st_o = run_fixed(oracle, logp, s, n, tr, draws, **kw)
    assert_bit_exact(pos, st, pos_o, st_o)
    assert (st["transformation_update_id"][0] == 1).all() and (st["num_eigenvalues"][0] == len(tr[2].reshape(-1, tr[2].shape[-1])[0])).all()
    if case.startswith("dim256_rank256_mvn"):   # the exact preconditioner whitens the target: shallow trees at a large step size
        assert st["depth"][-10:].mean() <= 4.5 and st["step_size"][-1].min() > 0.25


@pytest.mark.parametrize("chain_tiles", [0, 2], ids=["lockstep", "tile"])
@pytest.mark.parametrize("dim,rank,n_chains", [(64, 16, 21), (128, 64, 40), (256, 32, 35), (200, 200, 16)])
def test_matrix_core_kernel_bit_exact(oracle, dim, rank, n_chains, chain_tiles):
    """nuts_tile.hpp: 16 chains per block, U'z / U s / P x as MFMA products over the block's column tile.  Ragged chain
    counts (empty columns in the last tile), several tiles per block, launches cut in the middle, dims that are not 256."""
    rng = np.random.default_rng(dim + rank)
    prec, sigma = correlated_precision(rng, dim, 4)
    w, u = np.linalg.eigh(sigma)
    keep = np.argsort(np.abs(np.log(w)))[::-1][:rank]
    tr = (np.exp(rng.normal(0, 0.2, dim)), rng.normal(0, 0.5, dim), w[keep], np.ascontiguousarray(u[:, keep].T), rng.normal(0, 0.1, dim))
    s = lowrank_settings(num_chains=n_chains, seed=17, num_tune=40, freeze_transform=True)
    pos, st, pos_o, st_o = run_fixed(oracle, N.LogpSpec.mvn_precision(prec), s, n_chains, tr, 60, expect_tiles=True, splits=(1, 33), chain_tiles=chain_tiles)
    assert_bit_exact(pos, st, pos_o, st_o)
    assert len(np.unique(st["depth"])) >= 2                       # trees of different sizes inside a tile


def estimator_pair(oracle):
    """the oracle's LAPACK estimator, as a callback for the oracle and as a callback for the engine (same function)"""
    from oracle import lowrank as LR
    rec = []
    cb_o = LR.estimator_callback(rec)
    cb_e = C.cast(cb_o, _lib.LOWRANK_ESTIMATOR_FN)
    return cb_o, cb_e, rec


@pytest.mark.parametrize("case", ["diag_normal_dim12", "mvn_dim20_correlated", "funnel_dim11", "mvn_dim64_update_freq5"])
def test_lowrank_adaptation_bit_exact(oracle, case):
    rng = np.random.default_rng(abs(hash(case)) % 2**31)
    n, tune, draws, freq = 4, 120, 150, 20
    if case == "diag_normal_dim12":
        logp = N.LogpSpec.diag_normal(np.exp(rng.uniform(-2, 2, 12)))
    elif case == "mvn_dim20_correlated":
        logp = N.LogpSpec.mvn_precision(correlated_precision(rng, 20, 2)[0])
    elif case == "funnel_dim11":
        logp = N.LogpSpec.funnel(11)
    else:
        logp, n, tune, draws, freq = N.LogpSpec.mvn_precision(correlated_precision(rng, 64, 3)[0]), 3, 100, 120, 5
    s = lowrank_settings(num_chains=n, seed=5, num_tune=tune, store_mass_matrix=True)
    s.adapt_options.mass_matrix_update_freq = freq
    cb_o, cb_e, rec = estimator_pair(oracle)
    x0 = oracle.init_positions_uniform(s.seed, 0, n, logp.dim)
    b = N.ChainBatch(s, logp, n)
    assert (b.set_position(x0) == 0).all()
    b.set_lowrank_estimator(cb_e, n_threads=1)
    pos, st, vec = b.expanded_draw_many(draws, vectors=["mass_matrix_inv", "mass_matrix_eigvals", "gradient"])
    tpc = b.threads_per_chain()
    n_eng = len(rec)
    b.close()
    vo = {}
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, logp.dim, logp.params, oracle.gpu_cfg(tpc), n, x0,
                                        draws, estimator=cb_o, vectors=vo)
    assert failed == 0 and len(rec) == 2 * n_eng and n_eng >= n * 3          # both sides asked the estimator equally often
    assert_bit_exact(pos, st, pos_o, st_o)
    assert (st["num_eigenvalues"] == st_o["num_eigenvalues"]).all()
    for k in vec:
        both_nan = np.isnan(vec[k]) & np.isnan(vo[k])
        assert ((vec[k].view(np.uint64) == vo[k].view(np.uint64)) | both_nan).all(), k
    assert (st["transformation_update_id"] >= 0).sum() >= n * 3 and st["num_eigenvalues"].max() >= (1 if "mvn" in case else 0)


@pytest.mark.parametrize("kind", [N.KineticEnergyKind.EXACT_NORMAL, N.KineticEnergyKind.MICROCANONICAL], ids=["exact_normal", "microcanonical"])
def test_lowrank_with_trajectory_kinds_bit_exact(oracle, kind):
    """`LowRankNutsSettings::trajectory_kind` (src/sampler.rs:224-232 on NutsSettings<EuclideanAdaptOptions<LowRankSettings>>): the
    geodesic and the ESH leapfrog around the low-rank transformation — a given per-chain transformation, and the whole adaptation
    with the injected estimator."""
    rng = np.random.default_rng(1000 + kind)
    # (a) fixed per-chain transformations, iid normal
    dim, n = 40, 4
    s = lowrank_settings(num_chains=n, seed=31 + kind, num_tune=60, freeze_transform=True, trajectory_kind=kind)
    pos, st, pos_o, st_o = run_fixed(oracle, N.LogpSpec.iid_normal(dim, 1.0), s, n, random_transform(rng, dim, 6, per_chain=n), 90, splits=(33,))
    assert_bit_exact(pos, st, pos_o, st_o)
    # (b) adaptation on a correlated normal
    n, tune, draws = 4, 100, 130
    logp = N.LogpSpec.mvn_precision(correlated_precision(rng, 20, 2)[0])
    s = lowrank_settings(num_chains=n, seed=5, num_tune=tune, store_mass_matrix=True, trajectory_kind=kind)
    cb_o, cb_e, rec = estimator_pair(oracle)
    x0 = oracle.init_positions_uniform(s.seed, 0, n, logp.dim)
    b = N.ChainBatch(s, logp, n)
    assert (b.set_position(x0) == 0).all()
    b.set_lowrank_estimator(cb_e, n_threads=1)
    pos, st = b.draw_many(draws)
    tpc = b.threads_per_chain()
    b.close()
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, logp.dim, logp.params, oracle.gpu_cfg(tpc), n, x0,
                                        draws, estimator=cb_o)
    assert failed == 0
    assert_bit_exact(pos, st, pos_o, st_o)
    assert (st["num_eigenvalues"] == st_o["num_eigenvalues"]).all() and (st["transformation_update_id"] >= 0).sum() >= n * 3


def test_builtin_estimator_matches_lapack_restatement(oracle):
    """nm_lowrank_compute_update (csrc/lowrank_host.cpp) against oracle/lowrank.py on well-conditioned windows."""
    from oracle import lowrank as LR
    L = _lib.load()
    rng = np.random.default_rng(8)
    for dim, n, rank in ((12, 60, 2), (64, 200, 5), (30, 90, 3)):
        prec, sigma = correlated_precision(rng, dim, rank)
        x = np.linalg.cholesky(sigma) @ rng.normal(size=(dim, n)) + rng.normal(size=(dim, 1))
        g = -prec @ (x - 1.0)
        d, gg = np.ascontiguousarray(x.T), np.ascontiguousarray(g.T)
        m = min(dim, 2 * n)
        stds, mean, vals, vecs, mu = np.empty(dim), np.empty(dim), np.empty(m), np.empty((m, dim)), np.empty(dim)
        ne = C.c_uint64()
        rc = L.nm_lowrank_compute_update(None, dim, n, d.ctypes.data, gg.ctypes.data, 1e-5, 2.0, stds.ctypes.data, mean.ctypes.data,
                                         C.byref(ne), vals.ctypes.data, vecs.ctypes.data, mu.ctypes.data)
        ref = LR.compute_update(x, g, 1e-5, 2.0, rank_revealing=True)
        assert rc == 0 and ref is not None and ne.value == len(ref[2])
        k = ne.value
        op = lambda v, u: u.T @ np.diag(np.sqrt(v) - 1) @ u
        assert np.allclose(stds, ref[0], rtol=1e-12) and np.allclose(mean, ref[1], atol=1e-10)
        assert np.allclose(np.sort(vals[:k]), np.sort(ref[2]), rtol=1e-8)
        assert np.abs(op(vals[:k], vecs[:k]) - op(ref[2], ref[3].T)).max() < 1e-8 * max(1.0, np.abs(op(ref[2], ref[3].T)).max())
        assert np.allclose(mu, ref[4], atol=1e-8)


def twin_update(L, d, g, gamma, cutoff):
    """nm_lowrank_block_twin on one window given as [n][dim] rows"""
    n, dim = d.shape
    m = min(dim, 2 * n)
    stds, mean, vals, vecs, mu = np.zeros(dim), np.zeros(dim), np.zeros(m), np.zeros((m, dim)), np.zeros(dim)
    ne = C.c_uint64()
    rc = L.nm_lowrank_block_twin(None, dim, n, d.ctypes.data, g.ctypes.data, gamma, cutoff, stds.ctypes.data, mean.ctypes.data, C.byref(ne),
                                 vals.ctypes.data, vecs.ctypes.data, mu.ctypes.data)
    return rc, int(ne.value), stds, mean, vals, vecs, mu


@pytest.mark.parametrize("dim,n", [(1, 5), (5, 3), (8, 40), (20, 12), (37, 50), (64, 30), (64, 100), (128, 100), (130, 70), (256, 60),
                                   (300, 40), (512, 24)])           # (dim > 256: the kernel built for the 512-column vectors, round 5)
def test_estimator_kernel_is_its_twin_bit_for_bit(dim, n):
    """The estimator KERNEL (csrc/lowrank_device.hip: one 256-thread block per window) against its host twin (the same template
    code walked by one thread with the kernel's reduction trees): sigma, mean, mu, the eigenvalues, every eigenvector, the rank,
    the log-determinant and the None cases, bit for bit.  The twin is what the CPU tests pin to the reference's KATs and to the
    literal algorithm (tests/test_lowrank_estimator_builtin.py, impl "block")."""
    import test_lowrank_estimator_builtin as T
    L = _lib.load()
    rng = np.random.default_rng(dim * 977 + n)
    nw, gamma, cutoff = 7, 1e-5, 2.0
    D, G = np.zeros((nw, n, dim)), np.zeros((nw, n, dim))
    for w in range(nw):
        if dim >= 4:
            x, g = T.correlated_window(rng, dim, n, min(3, dim - 1))
        else:
            x = rng.normal(size=(dim, n)) * 3.0 + 1.0
            g = -(x - 1.0) / 9.0
        D[w], G[w] = x.T, g.T
    G[3, :, dim // 2] = 2.0                               # grad variance 0 -> sigma not finite -> None
    D[5] = D[5, :1]                                       # every draw the same point: zero variance -> None
    m = min(dim, 2 * n)
    stds, mean, mu = np.zeros((nw, dim)), np.zeros((nw, dim)), np.zeros((nw, dim))
    vals, vecs = np.zeros((nw, m)), np.zeros((nw, m, dim))
    n_eig, status, ld = np.zeros(nw, dtype=np.uint64), np.zeros(nw, dtype=np.uint64), np.zeros(nw, dtype=np.uint64)
    er = L.nm_lowrank_test_block_device(dim, n, nw, D.ctypes.data, G.ctypes.data, gamma, cutoff, stds.ctypes.data, mean.ctypes.data, n_eig.ctypes.data,
                                        vals.ctypes.data, vecs.ctypes.data, mu.ctypes.data, status.ctypes.data, ld.ctypes.data)
    assert er == 0
    some = 0
    for w in range(nw):
        rc, k, st, mn, vl, vc, m_ = twin_update(L, np.ascontiguousarray(D[w]), np.ascontiguousarray(G[w]), gamma, cutoff)
        assert int(status[w]) == rc, (w, int(status[w]), rc)
        if rc:
            continue
        some += 1
        assert int(n_eig[w]) == k
        for name, a, b in (("sigma", stds[w], st), ("mean", mean[w], mn), ("mu", mu[w], m_), ("vals", vals[w, :k], vl[:k]), ("vecs", vecs[w, :k], vc[:k])):
            assert (a.view(np.uint64) == b.view(np.uint64)).all(), (w, name, np.abs(a - b).max())
        want_ld = -0.0
        for v in vl[:k]:
            want_ld += -0.5 * np.log(v)
        assert abs(ld[w:w + 1].view(np.float64)[0] - want_ld) <= 1e-12 * (1.0 + abs(want_ld))
    assert some >= nw - 2 and status[3] == 1 and status[5] == 1


@pytest.mark.parametrize("case", ["mvn_dim20_correlated", "funnel_dim11", "mvn_dim64_update_freq5", "mvn_dim128"])
def test_lowrank_adaptation_device_estimator_bit_exact(oracle, case):
    """A whole LowRankNutsSettings warm-up as a DEFAULT run executes it — the built-in estimator on the device, one block per
    paused chain (NM_LR_PLACE_AUTO) — against the oracle with the device form's host twin as its estimator: bit-exact, every
    draw, every statistic, the mass-matrix vectors.  (The same run with the estimator on the host threads,
    NM_LR_PLACE_HOST, is a different rounding of the same algorithm: compared through the tolerances of
    test_builtin_estimator_whole_run_vs_literal_reference_algorithm.)"""
    L = _lib.load()
    rng = np.random.default_rng(abs(hash(case)) % 2**31)
    n, tune, draws, freq = 4, 120, 150, 20
    if case == "mvn_dim20_correlated":
        logp = N.LogpSpec.mvn_precision(correlated_precision(rng, 20, 2)[0])
    elif case == "funnel_dim11":
        logp = N.LogpSpec.funnel(11)
    elif case == "mvn_dim64_update_freq5":
        logp, n, tune, draws, freq = N.LogpSpec.mvn_precision(correlated_precision(rng, 64, 3)[0]), 3, 100, 120, 5
    else:
        logp, n, tune, draws, freq = N.LogpSpec.mvn_precision(correlated_precision(rng, 128, 5)[0]), 3, 160, 180, 20
    s = lowrank_settings(num_chains=n, seed=5, num_tune=tune, store_mass_matrix=True)
    s.adapt_options.mass_matrix_update_freq = freq
    x0 = oracle.init_positions_uniform(s.seed, 0, n, logp.dim)
    b = N.ChainBatch(s, logp, n)
    assert (b.set_position(x0) == 0).all()
    b.set_lowrank_estimator_place("device")
    pos, st, vec = b.expanded_draw_many(draws, vectors=["mass_matrix_inv", "mass_matrix_eigvals", "gradient"])
    tpc = b.threads_per_chain()
    n_dev = b.lowrank_device_updates()
    b.close()
    assert n_dev >= n * 3
    cb_o = C.cast(L.nm_lowrank_block_twin, oracle.ESTIMATOR_FN)
    vo = {}
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, logp.dim, logp.params, oracle.gpu_cfg(tpc), n, x0,
                                        draws, estimator=cb_o, vectors=vo)
    assert failed == 0
    assert_bit_exact(pos, st, pos_o, st_o)
    assert (st["num_eigenvalues"] == st_o["num_eigenvalues"]).all()
    for k in vec:
        both_nan = np.isnan(vec[k]) & np.isnan(vo[k])
        assert ((vec[k].view(np.uint64) == vo[k].view(np.uint64)) | both_nan).all(), k
    assert (st["transformation_update_id"] >= 0).sum() >= n * 3 and st["num_eigenvalues"].max() >= (1 if "mvn" in case else 0)


def test_estimator_place_host_and_device_agree_statistically_and_unsupported_shapes(oracle):
    """NM_LR_PLACE_HOST keeps the estimator on the host threads (no device updates); NM_LR_PLACE_DEVICE refuses a shape the block
    algorithm does not take (dim > 512; 256 until round 5) at the first window, NM_LR_PLACE_AUTO runs it on the host — and takes the device
    at dim 300, which rounds 3 - 4 left to the host threads."""
    rng = np.random.default_rng(8)
    logp = N.LogpSpec.mvn_precision(correlated_precision(rng, 24, 2)[0])
    s = lowrank_settings(num_chains=8, seed=3, num_tune=100)
    eps = {}
    for place in ("host", "device"):
        b = N.ChainBatch(s, logp, 8)
        b.set_position(b.init_positions_uniform())
        b.set_lowrank_estimator_place(place)
        _, st = b.draw_many(140)
        assert (b.lowrank_device_updates() > 0) == (place == "device")
        eps[place] = st["step_size"][100:].mean()
        b.close()
    assert abs(np.log(eps["host"] / eps["device"])) < np.log(1.3)
    s = lowrank_settings(num_chains=2, seed=3, num_tune=40)
    mid = N.LogpSpec.iid_normal(300, 1.0)
    b = N.ChainBatch(s, mid, 2)
    b.set_position(b.init_positions_uniform())
    b.draw_many(45)
    assert b.lowrank_device_updates() > 0                 # auto: the device (dim <= 512)
    b.close()
    big = N.LogpSpec.iid_normal(600, 1.0)
    b = N.ChainBatch(s, big, 2)
    b.set_position(b.init_positions_uniform())
    b.draw_many(45)
    assert b.lowrank_device_updates() == 0                # auto: host threads
    b.close()
    b = N.ChainBatch(s, big, 2)
    b.set_position(b.init_positions_uniform())
    b.set_lowrank_estimator_place("device")
    with pytest.raises(Exception, match="NM_LR_PLACE_DEVICE"):
        b.draw_many(45)
    b.close()


@pytest.mark.parametrize("dim,tune", [(64, 300), (128, 220)])
def test_builtin_estimator_whole_run_vs_literal_reference_algorithm(oracle, dim, tune):
    """A whole LowRankNutsSettings warm-up with the engine's BUILT-IN estimator (what a default run executes) against the oracle
    with the LITERAL reference estimator (oracle/lowrank.py rank_revealing=False = adapt/low_rank.rs:73-290 on LAPACK).
    The early windows have fewer draws than dims (10, 10, ..., 30, 50 draws at dim 64 / 128: rank deficient).  Stated tolerances
    (tests/test_lowrank_estimator_builtin.py):
      * up to and including the draw of the first update both sides see identical inputs: draws bit-exact;
      * EVERY window the engine hands to its estimator: the built-in's answer against the literal algorithm on the same
        window: sigma / mean 1e-13, the applied operator 1e-6 on full-rank windows (1e-3 where n_draws - 1 < 1.05 dim: measured 2.4e-6) and 0.25 on rank-deficient ones, signal eigenvalues (> 2 x cutoff) equal in number and within 5 %;
      * after the first update the two runs are different chaotic trajectories of the same sampler: the adapted step size,
        tree sizes and the quality of the final transformation agree statistically."""
    from oracle import lowrank as LR
    import test_lowrank_estimator_builtin as T
    L = _lib.load()
    rng = np.random.default_rng(dim)
    n = 4
    prec, sigma = correlated_precision(rng, dim, max(3, dim // 20))
    logp = N.LogpSpec.mvn_precision(prec)
    s = lowrank_settings(num_chains=n, seed=13, num_tune=tune)
    windows = []

    def recording_builtin(ctx, ndim, ndraws, draws, grads, gamma, cutoff, stds, mean, n_eig, vals, vecs, mu):
        d = np.ctypeslib.as_array(draws, shape=(ndraws, ndim)).copy()
        g = np.ctypeslib.as_array(grads, shape=(ndraws, ndim)).copy()
        rc = L.nm_lowrank_compute_update(None, ndim, ndraws, C.cast(draws, C.c_void_p), C.cast(grads, C.c_void_p), gamma, cutoff,
                                         C.cast(stds, C.c_void_p), C.cast(mean, C.c_void_p), n_eig, C.cast(vals, C.c_void_p),
                                         C.cast(vecs, C.c_void_p), C.cast(mu, C.c_void_p))
        k = int(n_eig[0]) if rc == 0 else 0
        res = None
        if rc == 0:
            res = (np.ctypeslib.as_array(stds, shape=(ndim,)).copy(), np.ctypeslib.as_array(mean, shape=(ndim,)).copy(),
                   np.ctypeslib.as_array(vals, shape=(max(k, 1),))[:k].copy(), np.ctypeslib.as_array(vecs, shape=(max(k, 1), ndim))[:k].T.copy(),
                   np.ctypeslib.as_array(mu, shape=(ndim,)).copy())
        windows.append((d.T, g.T, gamma, cutoff, res))
        return rc

    cb_e = _lib.LOWRANK_ESTIMATOR_FN(recording_builtin)
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    b = N.ChainBatch(s, logp, n)
    assert (b.set_position(x0) == 0).all()
    b.set_lowrank_estimator(cb_e, n_threads=1)
    pos, st = b.draw_many(tune + 60)
    tpc = b.threads_per_chain()
    stds_e, _ = b.mass_matrix()
    n_eig_e, vals_sqrt_e, vecs_e, _ = b.lowrank()
    b.close()
    cb_o = LR.estimator_callback(None, rank_revealing=False)
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, dim, logp.params, oracle.gpu_cfg(tpc), n, x0, tune + 60, estimator=cb_o)
    assert failed == 0
    # (1) identical until the first update has been applied
    upd = (st["transformation_update_id"][1:] >= 0).any(axis=1)     # (row 0 always reports the initial version: compared with -1)
    first = 1 + int(np.argmax(upd))                                  # the first draw whose adapt() replaced the transformation
    assert upd.any() and first >= 5
    assert_bit_exact(pos[:first], st[:first], pos_o[:first], st_o[:first])
    assert (pos[first].view(np.uint64) == pos_o[first].view(np.uint64)).all()      # (that draw itself still ran under the old one)
    # (2) every window of the run
    assert len(windows) >= n * 6
    n_def = n_full = 0
    worst = dict(full=0.0, deficient=0.0)
    for d, g, gamma, cutoff, bi in windows:
        lit = LR.compute_update(d, g, gamma, cutoff, rank_revealing=False)
        assert (bi is None) == (lit is None)
        if bi is None:
            continue
        deficient = d.shape[1] - 1 < dim
        assert np.max(np.abs(bi[0] - lit[0]) / lit[0]) <= T.TOL_DIAG and np.max(np.abs(bi[1] - lit[1])) <= T.TOL_DIAG * (1.0 + np.abs(lit[1]).max())
        a, bb = T.op_of(bi[2], bi[3]), T.op_of(lit[2], lit[3])
        d_bi = np.linalg.norm(a - bb, 2) / np.linalg.norm(bb, 2)
        if deficient:
            n_def += 1
            rr = LR.compute_update(d, g, gamma, cutoff, rank_revealing=True)
            d_rr = np.linalg.norm(T.op_of(rr[2], rr[3]) - bb, 2) / np.linalg.norm(bb, 2)
            assert d_bi <= T.TOL_RANK_DEFICIENT, (d.shape, d_bi, d_rr)      # (d_rr: how far the two LAPACK forms are from each other on this window)
            sig_bi, sig_lit = np.sort(bi[2][bi[2] > 4.0]), np.sort(lit[2][lit[2] > 4.0])
            if not (np.abs(np.concatenate([bi[2], lit[2]]) - 4.0) < 4.0 * T.TOL_SIGNAL_EIG).any():
                assert len(sig_bi) == len(sig_lit) and np.allclose(sig_bi, sig_lit, rtol=T.TOL_SIGNAL_EIG)
            worst["deficient"] = max(worst["deficient"], d_bi)
        else:
            n_full += 1
            marginal = d.shape[1] - 1 < 1.05 * dim           # just enough draws: the window's smallest singular value is ~0
            assert d_bi <= (T.TOL_MARGINAL_RANK if marginal else T.TOL_FULL_RANK), (d.shape, d_bi)
            worst["full"] = max(worst["full"], d_bi)
    assert n_def >= n * 3 and (n_full >= n or dim > 64)
    print(f"dim {dim}: {n_def} rank-deficient windows (worst operator departure {worst['deficient']:.3g}), {n_full} full-rank ({worst['full']:.3g})")
    # (3) statistically the same sampler afterwards
    tail, tail_o = st[tune:], st_o[tune:]
    assert abs(np.log(tail["step_size"].mean() / tail_o["step_size"].mean())) < np.log(1.35)
    assert abs(np.log(tail["n_steps"].mean() / tail_o["n_steps"].mean())) < np.log(1.5)
    assert tail["diverging"].mean() < 0.02 and (st["chain_status"] == 0).all()
    cs = []
    for c in range(n):                       # the adapted space's conditioning of the target's covariance
        k = int(n_eig_e[c])
        a = np.diag(stds_e[c]) @ (np.eye(dim) + vecs_e[c, :k].T @ np.diag(vals_sqrt_e[c, :k] - 1.0) @ vecs_e[c, :k])
        ai = np.linalg.inv(a)
        cs.append(np.linalg.cond(ai @ sigma @ ai.T))
    d0 = np.sqrt(np.diag(sigma))
    assert np.median(cs) < 0.3 * np.linalg.cond(sigma / np.outer(d0, d0))


def test_builtin_estimator_end_to_end(oracle):
    """LowRankNutsSettings with the engine's own estimator on a correlated normal: the adapted transformation shortens the
    trajectories compared with DiagNutsSettings, and the draws have the target's covariance."""
    rng = np.random.default_rng(21)
    dim, n = 48, 64
    prec, sigma = correlated_precision(rng, dim, 3, scale=200.0)
    logp = N.LogpSpec.mvn_precision(prec)
    res, cond = {}, {}
    for name, s in (("diag", N.DiagNutsSettings(num_chains=n, seed=9, num_tune=300)),
                    ("low_rank", lowrank_settings(num_chains=n, seed=9, num_tune=300))):
        b = N.ChainBatch(s, logp, n)
        b.set_position(b.init_positions_uniform())
        pos, st = b.draw_many(500)
        res[name] = (pos[300:], st[300:])
        stds, _ = b.mass_matrix()
        if name == "low_rank":
            n_eig, vals_sqrt, vecs, _ = b.lowrank()
            assert (n_eig >= 3).all() and (n_eig <= dim).all()
        cs = []
        for c in range(n):                       # condition number of the target's covariance in the adapted space
            a = np.diag(stds[c])
            if name == "low_rank":
                k = int(n_eig[c])
                a = a @ (np.eye(dim) + vecs[c, :k].T @ np.diag(vals_sqrt[c, :k] - 1.0) @ vecs[c, :k])
            ai = np.linalg.inv(a)
            cs.append(np.linalg.cond(ai @ sigma @ ai.T))
        cond[name] = float(np.median(cs))
        b.close()
    # (NUTS trees end on the U-turn of the many short directions, so tree size does not separate the two; the geometry does)
    assert cond["low_rank"] < 0.2 * cond["diag"] and cond["low_rank"] < 20, cond
    cov = np.cov(res["low_rank"][0].reshape(-1, dim).T)
    assert np.abs(cov - sigma).max() < 0.15 * np.abs(sigma).max()
    assert res["low_rank"][1]["diverging"].mean() < 0.01


def test_k5_full_size_properties(oracle):
    """BASELINE config 5 at full size: N(0, Sigma), full Sigma, dim 256 x 4096 chains, the exact dense preconditioner as a
    rank-256 transformation shared by all chains (frozen), step size adapted.  Size-independent properties: moments of the
    draws, chains 0-1 bit-exact against the oracle, results independent of how the batch is cut into launches."""
    rng = np.random.default_rng(55)
    dim, n, tune, draws = 256, 4096, 60, 80
    prec, sigma = correlated_precision(rng, dim, 8, scale=100.0)
    w, u = np.linalg.eigh(sigma)
    tr = (np.ones(dim), np.zeros(dim), w, np.ascontiguousarray(u.T), np.zeros(dim))
    logp = N.LogpSpec.mvn_precision(prec)
    s = lowrank_settings(num_chains=n, seed=77, num_tune=tune, freeze_transform=True)
    x0 = oracle.init_positions_uniform(s.seed, 0, n, dim)
    b = N.ChainBatch(s, logp, n)
    assert (b.set_position(x0) == 0).all()
    b.set_transform(*tr)
    pos1, st1 = b.draw_many(50)
    pos2, st2 = b.draw_many(draws - 50)
    pos, st = np.concatenate([pos1, pos2]), np.concatenate([st1, st2])
    tpc = b.threads_per_chain()
    assert b.tile_launches() >= 2                                   # the launches (draw_many cuts big ones into chunks) ran on the matrix cores
    order = b.reduce_order()
    assert order == 2 and b.lockstep_launches() == b.tile_launches()   # ... in their lockstep form
    b.close()
    assert (st["chain_status"] == 0).all() and st["diverging"].mean() < 1e-3
    sample = pos[tune:].reshape(-1, dim)
    z = sample @ (u / np.sqrt(w))                                   # whitened draws ~ N(0, I)
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.02
    assert np.abs(np.cov(z.T) - np.eye(dim)).max() < 0.06
    pos_o, st_o, _, failed = oracle.run(oracle_settings(oracle, s), logp.kind, dim, logp.params, oracle.gpu_cfg(tpc, lr_seq_dots=order), 2,
                                        x0[:2], draws, n_threads=2, transform=tr)
    assert failed == 0
    assert_bit_exact(pos[:, :2], st[:, :2], pos_o, st_o)
