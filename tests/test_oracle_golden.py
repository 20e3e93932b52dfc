"""CPU tests that pin the oracle: every golden vector / known-answer test the reference's own tests hold for
the hot path, the published ChaCha keystream vectors, the reference's behavioural envelopes,
and the bounds between the oracle's two arithmetic modes."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
    KATS = json.load(f)
with open(os.path.join(HERE, "golden", "oracle_k1_pins.json")) as f:
    K1 = json.load(f)


def ulps(a, b):
    a, b = np.float64(a), np.float64(b)
    if np.isnan(a) and (np.isnan(b) or np.isinf(b)):
        return 0
    if np.isnan(b) and (np.isnan(a) or np.isinf(a)):
        return 0
    if a == b:
        return 0
    ia, ib = np.int64(a.view(np.int64)), np.int64(b.view(np.int64))
    if ia < 0:
        ia = np.int64(-(2 ** 63)) - ia
    if ib < 0:
        ib = np.int64(-(2 ** 63)) - ib
    return abs(int(ia) - int(ib))


# ---- RNG -------------------------------------------------------------------------------------------
def test_chacha_published_keystream(oracle):
    k = KATS["chacha_zero_key_block0"]
    key = (C.c_uint32 * 8)()
    out = (C.c_uint32 * 16)()
    for rounds, name in ((20, "rounds20"), (8, "rounds8")):
        oracle.lib().nmo_chacha_block(key, 0, 0, rounds, out)
        assert bytes(out).hex() == k[name]


def test_rng_word_stream_and_distributions(oracle):
    L = oracle.lib()
    key = bytes(range(32))
    kb = (C.c_uint8 * 32).from_buffer_copy(key)
    words = (C.c_uint32 * 40)()
    L.nmo_rng_words(kb, 0, 40, words)
    # block b = words 16b..16b+15 with counter b
    k32 = (C.c_uint32 * 8).from_buffer_copy(key)
    blk = (C.c_uint32 * 16)()
    for b in range(2):
        L.nmo_chacha_block(k32, b, 0, 8, blk)
        assert list(blk) == list(words[16 * b:16 * b + 16])
    w = list(words)
    out = np.empty(8)
    L.nmo_rng_samples(kb, 0, 0.0, 0.0, 8, out)       # bool: sign bit of one u32
    assert [int(v) for v in out] == [1 if x >> 31 else 0 for x in w[:8]]
    L.nmo_rng_samples(kb, 1, 0.0, 0.0, 8, out)       # f64: 53 high bits of lo|hi<<32
    exp = [((w[2 * i] | (w[2 * i + 1] << 32)) >> 11) * 2.0 ** -53 for i in range(8)]
    assert list(out) == exp
    L.nmo_rng_samples(kb, 2, 0.25, 0.0, 8, out)      # Bernoulli(0.25): u64 < p * 2^64
    assert [int(v) for v in out] == [1 if (w[2 * i] | (w[2 * i + 1] << 32)) < 2 ** 62 else 0 for i in range(8)]
    L.nmo_rng_samples(kb, 3, 0.9, 1.1, 8, out)       # Uniform(0.9, 1.1)
    assert ((out >= 0.9) & (out < 1.1)).all()


def test_standard_normal_ziggurat(oracle):
    L = oracle.lib()
    x, f = np.empty(257), np.empty(257)
    L.nmo_zig_tables(x, f)
    assert x[1] == 3.654152885361008796 and x[256] == 0.0
    assert x[0] == 3.910757959537090045 and x[2] == 3.449278298560964462   # rand_distr's ZIG_NORM_X[0], [2] as printed there
    assert (np.diff(x) < 0).all() and f[256] == 1.0
    # the tables are fixed constants ('%.18f' round trip of the generator's doubles), the same in the engine's header
    for v in np.concatenate([x, f]):
        assert float("%.18f" % v) == v
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nuts_rs_amd", "csrc", "zig_tables.hpp")).read()
    vals = [float.fromhex(t) for t in re.findall(r"-?0x[0-9a-f.]+p[+-]\d+", hdr)]
    assert vals[:257] == list(x) and vals[257:514] == list(f)
    key = (C.c_uint8 * 32).from_buffer_copy(bytes(range(32)))
    n = 200000
    out = np.empty(n)
    for cfg in (oracle.ref_cfg(), oracle.gpu_cfg()):
        words = L.nmo_standard_normal_stream(C.byref(cfg), key, n, out)
        assert 2 * n < words < 2.1 * n                       # ~1.2 % of samples take extra words
        assert abs(out.mean()) < 0.01 and abs(out.std() - 1.0) < 0.01
        assert abs((np.abs(out) > 3.0).mean() - 0.0027) < 0.0006
        from scipy import stats
        assert stats.kstest(out, "norm").pvalue > 1e-3


def test_seed_from_u64_and_chain_keys(oracle):
    k0, k1 = oracle.chain_key(0, 0), oracle.chain_key(0, 1)
    assert k0 != k1 and len(k0) == 32
    assert oracle.chain_key(0, 0) == k0 and oracle.chain_key(1, 0) != k0
    x = oracle.init_positions_uniform(7, 3, 2, 5)
    assert x.shape == (2, 5) and ((x >= -1) & (x < 1)).all()
    # invariant to how chains are sharded over GPUs: position of chain 4 does not depend on offset
    assert (oracle.init_positions_uniform(7, 4, 1, 5)[0] == x[1]).all()


# ---- scalar math -----------------------------------------------------------------------------------
def test_logaddexp_reference_kats(oracle):
    k = KATS["logaddexp"]
    for cfg in (oracle.ref_cfg(), oracle.gpu_cfg()):
        la = lambda a, b: oracle.lib().nmo_logaddexp(C.byref(cfg), a, b)
        for a, b, e in k["neginf_cases"]:
            assert la(a, b) == e
        assert la(-math.inf, -math.inf) == -math.inf
        rng = np.random.default_rng(0)
        pts = list(map(tuple, k["regression_xy"])) + [tuple(p) for p in rng.uniform(-10, 10, size=(2000, 2))]
        for x, y in pts:
            b = la(x, y)
            assert abs(math.log(math.exp(x) + math.exp(y)) - b) < k["abs_tol_vs_naive"]
            assert b == la(y, x)
            assert la(x, -math.inf) == x
            assert math.isnan(la(math.nan, x))


def test_detmath_within_ulps_of_libm(oracle):
    """The restated exp/ln (what the GPU runs) against this box's libm (what the reference would call)."""
    L = oracle.lib()
    det, ref = oracle.gpu_cfg(), oracle.ref_cfg()
    rng = np.random.default_rng(1)
    worst = {0: 0, 1: 0, 2: 0}
    for x in np.concatenate([rng.uniform(-700, 700, 20000), rng.uniform(-2, 2, 20000), rng.normal(0, 1e-5, 2000)]):
        worst[0] = max(worst[0], ulps(L.nmo_scalar_fn(C.byref(det), 0, x, 0), L.nmo_scalar_fn(C.byref(ref), 0, x, 0)))
    for x in np.concatenate([np.exp(rng.uniform(-700, 700, 20000)), rng.uniform(0.5, 2, 20000), [1.0, 2.0, 0.5]]):
        worst[1] = max(worst[1], ulps(L.nmo_scalar_fn(C.byref(det), 1, x, 0), L.nmo_scalar_fn(C.byref(ref), 1, x, 0)))
    for x in np.concatenate([np.exp(rng.uniform(-40, 0, 20000)), [1.0, 0.0]]):
        worst[2] = max(worst[2], ulps(L.nmo_scalar_fn(C.byref(det), 2, x, 0), L.nmo_scalar_fn(C.byref(ref), 2, x, 0)))
    assert worst[0] <= 1 and worst[1] <= 1 and worst[2] <= 4, worst
    for x, e in ((0.0, 1.0), (-math.inf, 0.0), (math.inf, math.inf)):
        assert L.nmo_scalar_fn(C.byref(det), 0, x, 0) == e
    assert L.nmo_scalar_fn(C.byref(det), 1, 0.0, 0) == -math.inf and math.isnan(L.nmo_scalar_fn(C.byref(det), 1, -1.0, 0))
    assert L.nmo_scalar_fn(C.byref(det), 1, 1.0, 0) == 0.0
    # count^-k of dual averaging (dual_avg.rs:60) for the counts that occur
    for c in range(1, 2000):
        a, b = L.nmo_scalar_fn(C.byref(det), 6, float(c), -0.75), L.nmo_scalar_fn(C.byref(ref), 6, float(c), -0.75)
        assert abs(a - b) <= 4e-15 * b   # exp(-k ln c): a few ulp of libm powf, far inside 1e-9


# ---- vector primitives (reference src/math/util.rs:893-961, max_ulps = 32) ---------------------------
def test_primitive_formulas_and_regressions(oracle):
    L = oracle.lib()
    R = KATS["primitive_regressions"]
    rng = np.random.default_rng(2)
    cases = [(np.array(c["x"]), np.array(c["y"]), c["a"]) for c in R["axpy"]]
    for n in (0, 1, 3, 4, 7, 16, 17, 33):
        cases.append((rng.normal(size=n), rng.normal(size=n), float(rng.normal())))
    for x, y, a in cases:
        y2 = y.copy()
        L.nmo_axpy(np.ascontiguousarray(x), y2, a, len(x))
        out = np.empty(len(x))
        L.nmo_axpy_out(np.ascontiguousarray(x), np.ascontiguousarray(y), a, out, len(x))
        for i in range(len(x)):
            e = math.fma(a, x[i], y[i]) if hasattr(math, "fma") else float(np.float64(a) * x[i] + y[i])
            assert ulps(y2[i], e) <= R["max_ulps"] and ulps(out[i], e) <= R["max_ulps"]
    for c in R["axpy_out"]:
        x, y = np.array(c["x"]), np.array(c["y"])
        out = np.empty(len(x))
        L.nmo_axpy_out(x, y, c["a"], out, len(x))
        with np.errstate(all="ignore"):
            exp = y + c["a"] * x
        for i in range(len(x)):
            assert ulps(out[i], exp[i]) <= R["max_ulps"]
    for lanes in (1, 2, 4, 8):
        for cfg in (oracle.ref_cfg(lanes), oracle.gpu_cfg()):
            for c in R["vector_dot"]:
                with np.errstate(all="ignore"):
                    got = L.nmo_vector_dot(C.byref(cfg), np.array(c["x"]), np.array(c["y"]), len(c["x"]))
                    exp = float(np.sum(np.array(c["x"]) * np.array(c["y"])))
                assert ulps(got, exp) <= R["max_ulps"]
            for c in R["scalar_prods3"]:
                o = np.empty(2)
                a = [np.array(c[k]) for k in ("x1", "x2", "x3", "y1", "y2")]
                L.nmo_scalar_prods3(C.byref(cfg), *a, len(a[0]), o)
                with np.errstate(all="ignore"):
                    s = a[0] - a[1] + a[2]
                    assert ulps(o[0], float(s @ a[3])) <= R["max_ulps"] and ulps(o[1], float(s @ a[4])) <= R["max_ulps"]
            for n in (0, 1, 5, 16, 17, 100, 1024):
                x, y = rng.normal(size=n), rng.normal(size=n)
                got = L.nmo_vector_dot(C.byref(cfg), x, y, n)
                assert abs(got - math.fsum(x * y)) <= 1e-12 * max(1.0, float(np.abs(x * y).sum()))
                p1, n1, p2 = rng.normal(size=n), rng.normal(size=n), rng.normal(size=n)
                o = np.empty(2)
                L.nmo_scalar_prods3(C.byref(cfg), p1, n1, p2, x, y, n, o)
                s = p1 - n1 + p2
                assert abs(o[0] - math.fsum(s * x)) <= 1e-12 * max(1.0, float(np.abs(s * x).sum()))
                assert abs(o[1] - math.fsum(s * y)) <= 1e-12 * max(1.0, float(np.abs(s * y).sum()))


# ---- DiagMassMatrix known answers (reference src/transform/mod.rs:175-377) ---------------------------
def _diag_kat(oracle, cfg, case):
    s2 = np.array(case["sigma2"])
    n = len(s2)
    z, gz, xrt, sd, isd, mean = (np.empty(n) for _ in range(6))
    logp, logdet, logp_rt, logdet_rt = (C.c_double() for _ in range(4))
    rc = oracle.lib().nmo_diag_kat(C.byref(cfg), n, 1.0 / s2, np.array(case["draw_mean"]), np.array(case["grad_mean"]),
                                   s2, 1.0 / s2, np.array(case["x"]), z, gz, C.byref(logp), C.byref(logdet), xrt,
                                   C.byref(logp_rt), C.byref(logdet_rt), sd, isd, mean)
    assert rc == 0
    return dict(z=z, gz=gz, logp=logp.value, logdet=logdet.value, x_rt=xrt, logp_rt=logp_rt.value,
                logdet_rt=logdet_rt.value, stds=sd, s2=s2)


@pytest.mark.parametrize("mode", ["ref", "gpu"])
def test_diag_mass_matrix_reference_kats(oracle, mode):
    cfg = oracle.ref_cfg() if mode == "ref" else oracle.gpu_cfg()
    c = KATS["diag_transform_position_and_gradient"]
    r = _diag_kat(oracle, cfg, c)
    assert np.abs(r["z"] - c["expect_z"]).max() <= c["tol"] and np.abs(r["gz"] - c["expect_gz"]).max() <= c["tol"]
    assert abs(r["logdet"] - sum(-(0.5 * math.log(s)) for s in c["sigma2"])) < c["tol"]
    std_normal_logp = -0.5 * (len(r["z"]) * math.log(math.tau) + float((r["z"] ** 2).sum()))
    assert abs((r["logp"] - r["logdet"]) - std_normal_logp) < c["tol"]
    c = KATS["diag_round_trip"]
    r = _diag_kat(oracle, cfg, c)
    assert np.abs(r["x_rt"] - c["x"]).max() <= c["tol"]
    assert abs(r["logp"] - r["logp_rt"]) < c["tol"] and abs(r["logdet"] - r["logdet_rt"]) < c["tol"]
    c = KATS["diag_nonzero_mean"]
    r = _diag_kat(oracle, cfg, c)
    assert np.abs(r["z"] - c["expect_z"]).max() <= c["tol"]


def test_default_settings_match_reference(oracle):
    s = oracle.default_settings()
    for k, v in KATS["default_settings"].items():
        if k != "source":
            assert getattr(s, k) == v, k


# ---- chain level -------------------------------------------------------------------------------------
def test_k1_oracle_pins(oracle):
    """Seeded K1 draws are stable (regression pin of the restatement; see make_golden.py for what this is not)."""
    s = oracle.default_settings(seed=0, num_chains=4)
    pos, st, steps, failed = oracle.run(s, oracle.LOGP_IID_NORMAL, 10, [3.0], oracle.gpu_cfg(64), 4,
                                        np.zeros((4, 10)), 1400)
    assert failed == 0 and steps == K1["total_steps"]
    for i, t in enumerate(K1["draws"]):
        for c in range(4):
            assert [float(v).hex() for v in pos[t, c]] == K1["positions_hex"][i][c]
            assert float(st["step_size"][t, c]).hex() == K1["step_size_hex"][i][c]
            assert int(st["depth"][t, c]) == K1["depth"][i][c] and int(st["n_steps"][t, c]) == K1["n_steps"][i][c]


def test_reference_behavioural_envelopes(oracle):
    """The reference's own statistical tests, on the oracle in reference arithmetic (libm + SIMD-order sums):
    src/adapt_strategy.rs:367-435 (N(30,1) x 10 from x0 = 1.5, num_tune 100: draw 201 within 5 of 30 and not diverging),
    src/nuts.rs:399-419 (10 draws, not diverging), tests/sample_normal.rs:359-364 (6 chains x dim 100 defaults)."""
    cfg = oracle.ref_cfg()
    s = oracle.default_settings(num_tune=100, num_draws=100, seed=42)
    ch = oracle.Chain(s, oracle.LOGP_IID_NORMAL, 10, [30.0], cfg, 0)
    assert ch.set_position(np.full(10, 1.5)) == 0
    for t in range(201):
        pos, st, rc = ch.draw()
        assert rc == 0
    assert (np.abs(pos - 30.0) < 5).all() and not st["diverging"]      # exactly what the reference asserts
    s = oracle.default_settings(seed=1)
    pos, st, steps, failed = oracle.run(s, oracle.LOGP_IID_NORMAL, 100, [3.0], cfg, 6, np.zeros((6, 100)), 600)
    assert failed == 0 and st["diverging"][400:].sum() == 0 and st["diverging"][:20].sum() == st["diverging"].sum()
    post = pos[400:]
    assert abs(post.mean() - 3.0) < 0.05 and abs(post.std() - 1.0) < 0.05
    assert abs(st["mean_tree_accept"][400:].mean() - 0.8) < 0.08
    assert (st["tuning"][:400] == 1).all() and (st["tuning"][400:] == 0).all()
    assert (st["draw"][:, 0] == np.arange(600)).all()


def test_oracle_modes_agree_within_north_star_tolerance(oracle):
    """libm/SIMD-order (reference arithmetic) vs restated-exp/lane-order (GPU arithmetic): same trees, same draws
    to 1e-9 relative on a seeded run (north_star's tolerance)."""
    s = oracle.default_settings(seed=3)
    x0 = oracle.init_positions_uniform(3, 0, 4, 50)
    a = oracle.run(s, oracle.LOGP_IID_NORMAL, 50, [3.0], oracle.ref_cfg(), 4, x0, 500)
    b = oracle.run(s, oracle.LOGP_IID_NORMAL, 50, [3.0], oracle.gpu_cfg(), 4, x0, 500)
    assert (a[1]["depth"] == b[1]["depth"]).all() and (a[1]["n_steps"] == b[1]["n_steps"]).all()
    assert np.abs(a[0] - b[0]).max() <= 1e-9 * np.abs(a[0]).max()
    assert np.abs(a[1]["step_size"] - b[1]["step_size"]).max() <= 1e-9


def test_bad_initial_gradient_is_rejected(oracle):
    """init_state rejects a zero whitened gradient: starting iid normal at its mean."""
    s = oracle.default_settings()
    ch = oracle.Chain(s, oracle.LOGP_IID_NORMAL, 5, [3.0], oracle.gpu_cfg(), 0)
    assert ch.set_position(np.full(5, 3.0)) == 1
    ch2 = oracle.Chain(s, oracle.LOGP_IID_NORMAL, 5, [3.0], oracle.gpu_cfg(), 0)
    assert ch2.set_position(np.array([0.0, np.nan, 0, 0, 0])) == 1


def test_other_densities_gradients(oracle):
    """finite-difference check of the densities this repo defines (funnel, 8 schools)."""
    cfg = oracle.ref_cfg()
    rng = np.random.default_rng(5)
    y = [28., 8., -3., 7., -1., 1., 18., 12.]
    sig = [15., 10., 16., 11., 9., 11., 10., 18.]
    for kind, dim, params in ((oracle.LOGP_FUNNEL, 11, [0.0]), (oracle.LOGP_EIGHT_SCHOOLS, 10, y + sig),
                              (oracle.LOGP_DIAG_NORMAL, 4, [1.0, 0.25, 4.0, 2.0]),
                              (oracle.LOGP_MVN_PREC, 3, [2.0, 0.5, -0.3, 0.5, 1.0, 0.2, -0.3, 0.2, 3.0])):
        p = np.array(params)
        x = rng.normal(size=dim) * 0.5
        g = np.empty(dim)
        lp = C.c_double()
        assert oracle.lib().nmo_logp(C.byref(cfg), kind, dim, p, len(p), x, g, C.byref(lp)) == 0
        for i in range(dim):
            h = 1e-6
            xp, xm = x.copy(), x.copy()
            xp[i] += h
            xm[i] -= h
            lpp, lpm, gg = C.c_double(), C.c_double(), np.empty(dim)
            oracle.lib().nmo_logp(C.byref(cfg), kind, dim, p, len(p), xp, gg, C.byref(lpp))
            oracle.lib().nmo_logp(C.byref(cfg), kind, dim, p, len(p), xm, gg, C.byref(lpm))
            assert abs((lpp.value - lpm.value) / (2 * h) - g[i]) < 1e-5 * max(1.0, abs(g[i]))


def test_expanded_draw_vector_statistics(oracle):
    """The vector-valued statistics of `expanded_draw` (reference src/chain.rs:190-204 and the three extract_stats
    it flattens): what each row must contain follows from the reference's definitions."""
    O = oracle
    dim, n_chains, n_draws = 7, 6, 140
    s = O.default_settings(num_tune=80, seed=3)
    x0 = O.init_positions_uniform(3, 0, n_chains, dim)
    vec = {}
    cfg, par = O.ref_cfg(), np.zeros(1)
    pos, st, _, failed = O.run(s, O.LOGP_FUNNEL, dim, par, cfg, n_chains, x0, n_draws, vectors=vec)
    assert failed == 0

    def grad(x):
        g, lp = np.empty(dim), C.c_double()
        assert O.lib().nmo_logp(C.byref(cfg), O.LOGP_FUNNEL, dim, par, len(par), np.ascontiguousarray(x), g, C.byref(lp)) == 0
        return g
    # PointStats: gradient is the density's gradient at the draw; z, g_z are the whitened point under the
    # transformation the point was produced with (transformation_index)
    for t, c in [(5, 0), (60, 3), (139, 5)]:
        assert (vec["gradient"][t, c] == grad(pos[t, c])).all()
    # DiagMassMatrixStats: an event exactly when the version moved; first draw compares with -1
    upd = st["transformation_update_id"]
    assert (upd[0] >= 0).all() and (upd[-1] == -1).all()
    for c in range(n_chains):
        ids = upd[:, c][upd[:, c] >= 0]
        assert (np.diff(ids) > 0).all()                                        # strictly increasing versions
        rows = upd[:, c] >= 0
        assert (np.isnan(vec["mass_matrix_inv"][:, c]).all(axis=1) == ~rows).all()
        assert (np.isnan(vec["transformation_mu"][:, c]).all(axis=1) == ~rows).all()
        # after warm-up: z = (x - mu) / sigma with the last stored event
        last = np.flatnonzero(rows)[-1]
        sig, mu = vec["mass_matrix_inv"][last, c], vec["transformation_mu"][last, c]
        z = (pos[-1, c] - mu) / sig
        assert np.allclose(vec["transformed_position"][-1, c], z, rtol=1e-12, atol=1e-12)
        assert np.allclose(vec["transformed_gradient"][-1, c], vec["gradient"][-1, c] * sig, rtol=1e-14)
    # DivergenceStats: rows only on diverging draws; start/end are one leapfrog apart
    div = st["diverging"] != 0
    assert div.sum() > 0
    assert (np.isnan(vec["divergence_start"]).all(axis=2) == ~div).all()
    assert (np.isnan(vec["divergence_end"]).all(axis=2) == ~div).all()
    t, c = np.argwhere(div)[0]
    assert (vec["divergence_start_gradient"][t, c] == grad(vec["divergence_start"][t, c])).all()
    assert not (vec["divergence_start"][t, c] == vec["divergence_end"][t, c]).all()


def test_adam_step_size_adaptation(oracle):
    """Adam adaptor (reference src/stepsize/adam.rs:42-112): the update formula on a hand-computed sequence, and the
    sampler converging with it (the reference has no dedicated test; the envelope is adapt_strategy.rs:367-435's)."""
    O = oracle
    # closed form of the first step: m_hat = g, v_hat = g^2  =>  log_step += lr * g / (|g| + eps)
    b1, b2, eps, lr, target = 0.9, 0.999, 1e-8, 0.05, 0.8
    log_step, m, v = np.log(0.1), 0.0, 0.0
    expect = []
    for t, acc in enumerate([0.95, 0.6, 0.81, 0.2], start=1):
        g = acc - target
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        log_step += lr * (m / (1 - b1 ** t)) / (np.sqrt(v / (1 - b2 ** t)) + eps)
        expect.append(log_step)
    got = O.adam_sequence(0.1, [0.95, 0.6, 0.81, 0.2], target, b1, b2, eps, lr)
    assert np.allclose(got, expect, rtol=1e-14)
    assert abs(got[0] - (np.log(0.1) + lr * 0.15 / (0.15 + eps))) < 1e-15
    s = O.default_settings(num_tune=300, seed=5, step_size_method=1)
    x0 = O.init_positions_uniform(5, 0, 4, 10)
    pos, st, _, failed = O.run(s, O.LOGP_IID_NORMAL, 10, [3.0], O.ref_cfg(), 4, x0, 500)
    assert failed == 0 and st["diverging"][300:].sum() == 0
    assert abs(pos[300:].mean() - 3.0) < 0.15
    acc = st["mean_tree_accept"][200:300].mean()
    assert 0.6 < acc < 0.95                                   # moved the step size toward target_accept = 0.8
    assert (st["step_size_bar"][-1] > 0.2).all()              # from initial_step 0.1 up to the O(1) scale of N(3, 1)


def test_reference_crate_draws(oracle):
    """Closes the RNG pin where somebody could run the real crate: tools/ref_golden (Rust, needs cargo — absent from
    the build image) dumps seeded K1 draws of nuts-rs itself into tests/golden/reference_k1_draws.json; the oracle in
    its reference mode (libm, SIMD-order sums) must then reproduce them: same tree sizes, positions within the
    north-star tolerance.  Skipped while the fixture does not exist (parity unpinned)."""
    path = os.path.join(HERE, "golden", "reference_k1_draws.json")
    if not os.path.exists(path):
        pytest.skip("no reference_k1_draws.json: tools/ref_golden has not been run (no Rust toolchain in this image)")
    ref = json.load(open(path))
    n = ref["num_tune"] + ref["num_draws"]
    s = oracle.default_settings(seed=ref["seed"], num_chains=4, num_tune=ref["num_tune"], num_draws=ref["num_draws"])
    pos, st, _, failed = oracle.run(s, oracle.LOGP_IID_NORMAL, 10, [3.0], oracle.ref_cfg(), 4, np.zeros((4, 10)), n)
    assert failed == 0
    for ch in ref["chains"]:
        c = ch["chain"]
        assert (st["n_steps"][:, c] == np.array(ch["num_steps"])).all(), "tree sizes differ: the random stream is not the crate's"
        assert np.allclose(pos[:, c], np.array(ch["draws"]), rtol=1e-9, atol=1e-12)


# ---- low-rank transformation (reference src/transform/low_rank.rs, adapt/low_rank.rs) ----------------------------------
def _std_normal_logp(z):
    return -0.5 * (len(z) * math.log(2 * math.pi) + float(np.sum(np.asarray(z) ** 2)))


@pytest.mark.parametrize("mode", ["ref", "gpu"])
def test_lowrank_mass_matrix_reference_kats(oracle, mode):
    """The reference's own LowRankMassMatrix tests (src/transform/mod.rs:391-674), tolerance 1e-12 as there."""
    cfg = oracle.ref_cfg() if mode == "ref" else oracle.gpu_cfg()
    for name in ("lowrank_transform_position_and_gradient", "lowrank_round_trip", "lowrank_with_rank1_correction",
                 "lowrank_nonzero_mean"):
        k = KATS[name]
        tol = k["tol"]
        if "sigma2" in k:
            sigma2 = np.array(k["sigma2"])
            prec, stds = 1.0 / sigma2, np.sqrt(sigma2)
        else:
            prec, stds = np.array(k["precision_diag"]), np.array(k["stds"])
        mean = np.array(k["mean"], dtype=float)
        x = mean + stds if k.get("x_is_mean_plus_sigma") else np.array(k["x"])
        r = oracle.lowrank_kat(cfg, prec, stds, mean, k["vals"], k["vecs"], k["mu_lr"], x)
        assert r["rc"] == 0
        if "expect_z" in k:
            assert np.abs(r["z"] - k["expect_z"]).max() <= tol, name
        if "expect_gz" in k:
            assert np.abs(r["gz"] - k["expect_gz"]).max() <= tol, name
        assert np.abs(r["x_rt"] - x).max() <= tol, name                       # init_from_transformed_position recovers x
        assert abs(r["logp"] - r["logp_rt"]) < tol and abs(r["logdet"] - r["logdet_rt"]) < tol
        expected_logdet = float(np.sum(-np.log(stds))) + float(np.sum(-0.5 * np.log(np.array(k["vals"], dtype=float))))
        assert abs(r["logdet"] - expected_logdet) < tol, name
        if k.get("adapted_logp_is_standard_normal"):
            assert abs((r["logp"] - r["logdet"]) - _std_normal_logp(r["z"])) < tol, name


@pytest.mark.parametrize("mode", ["ref", "gpu"])
def test_lowrank_module_round_trips(oracle, mode):
    """src/transform/low_rank.rs:415-533: compute_transformed_position and compute_untransformed_position are inverses."""
    cfg = oracle.ref_cfg() if mode == "ref" else oracle.gpu_cfg()
    K = KATS["lowrank_module_round_trips"]
    for c in K["cases"]:
        a = (np.ones(3), c["stds"], c["mean"], c["vals"], c["vecs"], c["mu_lr"])
        first, second = (0, 1) if c["start"] == "x" else (1, 0)
        mid = oracle.lowrank_kat(cfg, *a, c["point"], which=first)["z"]
        back = oracle.lowrank_kat(cfg, *a, mid, which=second)["z"]
        assert np.abs(back - c["point"]).max() <= K["tol"], c["name"]


def test_lowrank_estimator_reference_kats(oracle):
    """src/transform/adapt/low_rank.rs:354-407 (test_spd_mean, test_estimate_mass_matrix) on the oracle's estimator."""
    from oracle import lowrank as LR
    K = KATS["lowrank_estimator"]
    k = K["spd_mean"]
    out = LR.spd_mean(np.diag(k["x_diag"]), np.diag(k["y_diag"]))
    assert np.allclose(out, np.diag(k["expected_diag"]), rtol=k["rel_tol"], atol=k["abs_tol"])
    k = K["estimate_mass_matrix"]
    draws = np.random.default_rng(1).normal(size=tuple(k["shape"]))
    vals, vecs = LR.estimate_mass_matrix(draws, -draws, k["gamma"])
    assert (vals > 0).all() and np.isfinite(vecs).all()
    assert np.allclose(vals, 1.0, rtol=k["rel_tol"], atol=k["abs_tol"])
    # the whole update on an exactly low-rank correlated Gaussian: x = L e, score = -Sigma^-1 x  =>  F whitens it
    rng = np.random.default_rng(5)
    dim, n = 12, 60
    u = np.linalg.qr(rng.normal(size=(dim, 2)))[0]
    sigma = np.eye(dim) + u @ np.diag([30.0, 8.0]) @ u.T
    x = np.linalg.cholesky(sigma) @ rng.normal(size=(dim, n))
    g = -np.linalg.solve(sigma, x)
    stds, mean, vals, vecs, mu = LR.compute_update(x, g, 1e-5, 2.0)
    assert len(vals) >= 2 and vals.max() > 4 and np.isfinite(vecs).all()
    a = np.eye(dim) + vecs @ np.diag(np.sqrt(vals) - 1.0) @ vecs.T            # F's linear part without sigma
    cov_adapted = np.linalg.inv(np.diag(stds) @ a) @ sigma @ np.linalg.inv(np.diag(stds) @ a).T
    assert np.linalg.cond(cov_adapted) < 0.2 * np.linalg.cond(np.diag(1 / stds) @ sigma @ np.diag(1 / stds))


def test_default_settings_low_rank(oracle):
    k = KATS["default_settings_low_rank"]
    s = oracle.default_settings(low_rank=True)
    for name, v in k.items():
        if name != "source":
            assert getattr(s, name) == v, name
    assert s.adaptation == oracle.ADAPT_LOW_RANK


def test_lowrank_adaptation_behaviour(oracle):
    """LowRankNutsSettings on a correlated normal: the adapted transformation must shorten the trajectories compared with
    the diagonal adaptation (the reason the reference has it), and a fixed exact transformation must whiten the target."""
    from oracle import lowrank as LR
    dim = 16
    rng = np.random.default_rng(11)
    u = np.linalg.qr(rng.normal(size=(dim, 2)))[0]
    sigma = np.eye(dim) + u @ np.diag([200.0, 50.0]) @ u.T
    prec = np.linalg.inv(sigma)
    prec = (prec + prec.T) / 2
    x0 = oracle.init_positions_uniform(3, 0, 2, dim)
    steps = {}
    for low_rank in (False, True):
        s = oracle.default_settings(low_rank=low_rank, seed=3, num_tune=300, num_chains=2)
        est = LR.estimator_callback() if low_rank else None
        pos, st, _, failed = oracle.run(s, oracle.LOGP_MVN_PREC, dim, prec.reshape(-1), oracle.ref_cfg(), 2, x0, 400,
                                        estimator=est)
        assert failed == 0
        steps[low_rank] = st["n_steps"][300:].mean()
        if low_rank:
            assert (st["num_eigenvalues"][st["transformation_update_id"] >= 0][-1:] >= 1).all()
            assert np.abs(np.cov(pos[300:].reshape(-1, dim).T) - sigma).max() < 0.6 * np.abs(sigma).max()
    assert steps[True] < 0.8 * steps[False], steps


def test_arithmetic_bridge_is_no_wider_than_the_references_own_simd_spread(oracle):
    """north_star: draws "within 1e-9 relative" of the CpuMath path.  The reference's sums depend on the SIMD width pulp
    picks at run time (src/math/util.rs:357-395), and NUTS dynamics amplify a last-bit difference by a constant factor per
    draw, so two runs of the REFERENCE on different CPUs leave the 1e-9 band after a few dozen draws.  The engine's
    arithmetic (restated exp / ln, lane-order sums) must do no worse than that: same tree sizes and 1e-9 agreement for as
    long as the reference agrees with itself (tools/arithmetic_bridge.py writes the full report to profiles/)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("arithmetic_bridge", os.path.join(os.path.dirname(HERE), "tools", "arithmetic_bridge.py"))
    ab = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ab)
    cfgs = ab.configs()
    for name in ("iid_normal_50", "eight_schools_10", "funnel_101"):
        kind, dim, params, tpc = cfgs[name]
        firsts = {"gpu": [], "simd": []}
        for seed in (1, 2, 3):
            s = oracle.default_settings(seed=seed, num_chains=4)
            x0 = oracle.init_positions_uniform(seed, 0, 4, dim)
            run = lambda cfg: oracle.run(s, kind, dim, params, cfg, 4, x0, 120, n_threads=4)
            ref4, gpu, ref8 = run(oracle.ref_cfg(4)), run(oracle.gpu_cfg(tpc)), run(oracle.ref_cfg(8))
            f_gpu, w_gpu = ab.compare(ref4, gpu)
            f_simd, _ = ab.compare(ref4, ref8)
            assert (w_gpu <= 1e-9).all()
            firsts["gpu"] += list(f_gpu)
            firsts["simd"] += list(f_simd)
        assert min(firsts["gpu"]) >= 10, (name, firsts)                                   # nothing departs in the first draws
        assert np.median(firsts["gpu"]) >= 0.6 * np.median(firsts["simd"]), (name, firsts)


# ---- the non-Euclidean trajectory kinds (NutsSettings::trajectory_kind) ----------------------------------
def test_trajectory_kind_primitives_reference_formulas(oracle):
    """The reference's own tests of std_norm_flow / std_norm_grad_flow (src/math/util.rs:808-868) compare the SIMD kernels
    with the scalar formulas; the same formulas pin the oracle here — exactly, head (fused) and scalar tail (unfused)
    separately — plus esh_momentum_update / array_normalize (src/math/cpu_math.rs:496-551) against their published algebra."""
    rng = np.random.default_rng(12)
    ref = oracle.ref_cfg()
    fma = math.fma if hasattr(math, "fma") else None
    for n in (3, 4, 32, 35):
        for eps in (-7.3, -0.4, 0.0, 0.25, 3.9):
            p, v, g = rng.normal(size=n), rng.normal(size=n), rng.normal(size=n)
            es, ec = math.sin(eps), math.cos(eps)
            po, vo = oracle.traj_kat(ref, "flow", p, v, eps=eps)
            gv = oracle.traj_kat(ref, "grad_flow", p, g, v, eps=eps)
            head = n - n % 4
            for i in range(n):
                if i < head and fma:
                    assert po[i] == fma(p[i], ec, v[i] * es) and vo[i] == fma(p[i], -es, v[i] * ec)
                    assert gv[i] == fma(eps, p[i] + g[i], v[i])
                elif i >= head:       # the scalar tail loops are written without mul_add (util.rs:561-565, :644-646)
                    assert po[i] == p[i] * ec + v[i] * es and vo[i] == p[i] * (-es) + v[i] * ec
                    assert gv[i] == v[i] + eps * (p[i] + g[i])
                assert abs(po[i] - (p[i] * ec + v[i] * es)) <= 4e-16 * (abs(p[i]) + abs(v[i]))
            # the flow is a rotation of every (q_i, v_i) plane
            assert np.allclose(po ** 2 + vo ** 2, p ** 2 + v ** 2, rtol=1e-14)
    for n in (2, 7, 64):
        g, u = rng.normal(size=n) * 3, rng.normal(size=n)
        un = oracle.traj_kat(ref, "normalize", u)
        assert un.tolist() == (u * (1.0 / math.sqrt(sum(x * x for x in u)))).tolist()
        for step in (0.3, -0.3, 2.0):
            m, dke = oracle.traj_kat(ref, "esh", g, un, eps=step)
            gn = np.linalg.norm(g)
            e = g / gn
            ue = float(un @ e)
            delta = step * gn / (n - 1)
            zeta = math.exp(-delta)
            raw = e * (1 - zeta) * (1 + zeta + ue * (1 - zeta)) + 2 * zeta * un
            assert np.allclose(m, raw / np.linalg.norm(raw), rtol=1e-12, atol=1e-15)
            assert abs(np.linalg.norm(m) - 1) < 1e-15
            assert abs(dke - (delta - math.log(2) + math.log1p(ue + (1 - ue) * zeta * zeta)) * (n - 1)) < 1e-12 * max(1, abs(dke))
            # the engine's arithmetic differs from it by reduction order only
            m2, dke2 = oracle.traj_kat(oracle.gpu_cfg(64), "esh", g, un, eps=step)
            assert np.abs(m - m2).max() < 1e-14 and abs(dke - dke2) < 1e-12 * max(1, abs(dke))


def test_det_sincos_within_an_ulp_of_libm(oracle):
    gpu, ref = oracle.gpu_cfg(64), oracle.ref_cfg()
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-10, 10, 4000), rng.uniform(-3000, 3000, 500),
                         [0.0, -0.0, 1e-300, 5e-9, math.pi / 4, math.pi / 2, math.pi, 1.5707963267948966, 3.0, 100.0]])
    for x in xs:
        s_, c_ = oracle.traj_kat(gpu, "sincos", eps=float(x))
        assert ulps(s_, math.sin(x)) <= 1 and ulps(c_, math.cos(x)) <= 1, x
        sm, cm = oracle.traj_kat(gpu, "sincos", eps=float(-x))
        assert sm == -s_ and cm == c_                     # exactly odd / even, like libm's
        sr, cr = oracle.traj_kat(ref, "sincos", eps=float(x))      # the platform libm (this interpreter may carry another build of it)
        assert ulps(sr, math.sin(x)) <= 1 and ulps(cr, math.cos(x)) <= 1
    for bad in (float("nan"), float("inf"), -float("inf")):
        s_, c_ = oracle.traj_kat(gpu, "sincos", eps=bad)
        assert math.isnan(s_) and math.isnan(c_)


def test_trajectory_kinds_sample_the_right_posterior(oracle):
    """Both kinds inside the NUTS tree reproduce the moments of an iid normal (the reference's envelope for the Euclidean
    kind, src/adapt_strategy.rs:367-435); the exact-normal geodesic accepts (almost) everything; and the two arithmetic
    modes of the oracle agree on the first draws."""
    dim, n = 20, 4
    x0 = oracle.init_positions_uniform(5, 0, n, dim)
    for kind in (oracle.TRAJ_EXACT_NORMAL, oracle.TRAJ_MICROCANONICAL):
        s = oracle.default_settings()
        s.num_tune, s.num_draws, s.seed, s.trajectory_kind = 300, 300, 5, kind
        pos, st, steps, failed = oracle.run(s, 0, dim, np.array([3.0]), oracle.ref_cfg(), n, x0, 600)
        post = pos[300:]
        assert failed == 0 and abs(post.mean() - 3.0) < 0.1 and abs(post.var() - 1.0) < 0.15
        if kind == oracle.TRAJ_EXACT_NORMAL:
            assert st["mean_tree_accept"][300:].mean() > 0.99
        else:
            assert st["diverging"].sum() <= 5
        # (only the first draws: at step sizes near pi — the dual-averaging cap — two geodesic steps return to the start and the
        # U-turn products are pure rounding noise, so trajectories of the two arithmetics part early for this kind)
        pos2, st2, _, _ = oracle.run(s, 0, dim, np.array([3.0]), oracle.gpu_cfg(64), n, x0, 3)
        assert np.allclose(pos2[:3], pos[:3], rtol=1e-9, atol=1e-12)


# ---- MclmcChain (src/mclmc.rs; experimental upstream) ----------------------------------------------------
def test_mclmc_reference_tests(oracle):
    """The reference's own MCLMC tests (src/mclmc.rs:566-680): DiagMclmcSettings on a 10-dim N(3, 1) from x0 = 0, the three
    trajectory kinds, 500 draws without a divergence and the last position near the mean; plus the default settings
    (src/sampler.rs:342-374) and what the draw statistics must satisfy by construction."""
    d = oracle.default_settings(mclmc=True)
    assert (d.sampler, d.mclmc_step_size, d.momentum_decoherence_length, d.num_tune, d.num_draws, d.num_chains) == (1, 0.5, 3.0, 400, 1000, 6)
    assert (d.subsample_frequency, d.dynamic_step_size, d.mclmc_trajectory_kind, d.trajectory_switch_fraction) == (1.0, 1, 2, 0.3)
    assert d.max_energy_error == 1000.0 and d.step_size_method == 2 and d.fixed_step_size == 0.5
    dim = 10
    for kind, step in ((oracle.MCLMC_MICROCANONICAL, 0.5), (oracle.MCLMC_EUCLIDEAN, 0.3), (oracle.MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL, 0.5)):
        s = oracle.default_settings(mclmc=True, num_tune=200, num_draws=500, mclmc_step_size=step, mclmc_trajectory_kind=kind, seed=kind)
        pos, st, steps, failed = oracle.run(s, 0, dim, np.array([3.0]), oracle.ref_cfg(), 4, np.zeros((4, dim)), 500)
        assert failed == 0 and st["diverging"].sum() == 0
        assert (np.abs(pos[-1].mean(axis=1) - 3.0) < 3.0).all()
        assert abs(pos[200:].mean() - 3.0) < 0.2
        # num_steps = round(L / eps) leapfrogs per draw at factor 1: average_step_size is the (jittered) step size of the draw
        assert (st["depth"] == st["n_steps"]).all() and (st["index_in_trajectory"] == st["depth"]).all()
        assert np.allclose(st["average_step_size"][1:], st["step_size"][:-1], rtol=1e-12)
        assert (st["energy_change"] == st["energy_error"]).all()


def test_mclmc_ladder_and_divergences(oracle):
    """dynamic_step_size: a divergent step halves the factor and asks for two steps (at most 10 halvings); without it the
    draw is a divergence, the chain stays where it was (src/mclmc.rs:236-388)."""
    dim, n = 11, 6
    x0 = oracle.init_positions_uniform(8, 0, n, dim)
    common = dict(num_tune=100, seed=8, mclmc_step_size=0.8, fixed_step_size=0.8, max_energy_error=2.0)
    s = oracle.default_settings(mclmc=True, **common)
    pos, st, _, failed = oracle.run(s, 2, dim, np.zeros(0), oracle.ref_cfg(), n, x0, 150)
    assert failed == 0
    laddered = st["average_step_size"] < st["step_size"].max() * 0.7
    assert laddered.sum() > 0 and (st["depth"][laddered] > 4).all()
    s2 = oracle.default_settings(mclmc=True, dynamic_step_size=0, **common)
    pos2, st2, _, _ = oracle.run(s2, 2, dim, np.zeros(0), oracle.ref_cfg(), n, x0, 150)
    div = st2["diverging"] != 0
    assert div.sum() > 0
    t, c = np.argwhere(div & (np.arange(150)[:, None] > 0))[0]
    assert (pos2[t, c] == pos2[t - 1, c]).all()                 # the draw repeats the previous position
    assert st2["index_in_trajectory"][t, c] == 0 and st2["energy_error"][t, c] == 0.0


def test_det_expm1_within_ulps_of_libm(oracle):
    """exp_m1 of the isokinetic refresh (transformed_hamiltonian.rs:800-801) in the engine's arithmetic."""
    import ctypes as C
    L = oracle.lib()
    if not hasattr(L, "nmo_scalar_fn"):
        pytest.skip("no scalar entry point")
    cfg = oracle.gpu_cfg(64)
    rng = np.random.default_rng(4)
    L.nmo_scalar_fn.restype = C.c_double
    for x in np.concatenate([rng.uniform(-0.35, 0.35, 3000), rng.uniform(-5, 5, 500), [0.0, 1e-300, -1e-20, 0.35, -0.35, 0.3500001]]):
        got = L.nmo_scalar_fn(C.byref(cfg), 7, C.c_double(float(x)), C.c_double(0.0))
        assert ulps(got, math.expm1(x)) <= 3, x
