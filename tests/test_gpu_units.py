"""GPU unit parity of the kernels' building blocks (the batched per-vector `Math` seam of include/nuts_amd.h)
against the CPU oracle, through the C ABI, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import nuts_rs_amd as N

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ptr(t):
    return C.c_void_p(t.data_ptr())


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_scalar_math_bit_exact(oracle):
    """exp / ln / ln_1p / logaddexp restatements and IEEE sqrt, division: device == oracle, every bit."""
    L = N.load_library()
    rng = np.random.default_rng(0)
    n = 200000
    cfg = oracle.gpu_cfg()
    cases = {
        0: (np.concatenate([rng.uniform(-750, 720, n), rng.uniform(-1, 1, n), rng.normal(0, 1e-6, 1000),
                            [0.0, -0.0, np.inf, -np.inf, np.nan, 709.9, -745.2, 1e-300]]), None),
        1: (np.concatenate([np.exp(rng.uniform(-745, 709, n)), rng.uniform(0.9, 1.1, n),
                            [0.0, -1.0, 1.0, np.inf, np.nan, 5e-324, 2.2250738585072014e-308]]), None),
        2: (np.concatenate([np.exp(rng.uniform(-60, 0, n)), [0.0, 1.0, 1e-17]]), None),
        3: (rng.uniform(-50, 50, n), rng.uniform(-50, 50, n)),
        4: (np.concatenate([np.exp(rng.uniform(-700, 700, n)), [0.0, 1.0, 2.0]]), None),
        5: (rng.normal(size=n) * np.exp(rng.uniform(-300, 300, n)), rng.normal(size=n) * np.exp(rng.uniform(-300, 300, n))),
        # exp_m1 (the isokinetic momentum refresh) and sin / cos (the exact-normal geodesic step)
        7: (np.concatenate([rng.uniform(-0.36, 0.36, n), rng.uniform(-40, 40, 20000), [0.0, -0.0, 0.35, -0.35, np.inf, -np.inf, np.nan, 1e-300]]), None),
        8: (np.concatenate([rng.uniform(-10, 10, n), rng.uniform(-1e4, 1e4, 20000), [0.0, -0.0, np.pi / 2, np.inf, np.nan, 1e-300]]), None),
        9: (np.concatenate([rng.uniform(-10, 10, n), rng.uniform(-1e4, 1e4, 20000), [0.0, -0.0, np.pi / 2, np.inf, np.nan, 1e-300]]), None),
    }
    for op, (a, b) in cases.items():
        da, db = dev(a), dev(b) if b is not None else None
        out = torch.empty(len(a), dtype=torch.float64, device="cuda")
        assert L.nm_scalar_math_batch(op, len(a), ptr(da), ptr(db) if db is not None else None, ptr(out), None) == 0
        got = out.cpu().numpy()
        bb = b if b is not None else np.zeros_like(a)
        with np.errstate(all="ignore"):
            exp = np.array([oracle.lib().nmo_scalar_fn(C.byref(cfg), op, float(x), float(y)) for x, y in zip(a, bb)])
        nan_both = np.isnan(got) & np.isnan(exp)
        bad = np.argwhere((bits(got) != bits(exp)) & ~nan_both)
        assert bad.size == 0, f"op {op}: {len(bad)} mismatches, first a={a[bad[0][0]]!r} got={got[bad[0][0]]!r} exp={exp[bad[0][0]]!r}"


def test_standard_normal_stream_bit_exact(oracle):
    """ChaCha8 -> ziggurat on the device (lane-parallel speculative fill) == the sequential oracle stream."""
    L = N.load_library()
    n, count = 48, 3000
    keys = b"".join(oracle.chain_key(11, c) for c in range(n))
    kb = (C.c_uint8 * (32 * n)).from_buffer_copy(keys)
    out = torch.empty((n, count), dtype=torch.float64, device="cuda")
    words = (C.c_uint64 * n)()
    assert L.nm_standard_normal_batch(n, count, kb, ptr(out), words, None) == 0
    got = out.cpu().numpy()
    cfg = oracle.gpu_cfg()
    slow = 0
    for i in range(n):
        exp = np.empty(count)
        k = (C.c_uint8 * 32).from_buffer_copy(keys[32 * i:32 * i + 32])
        w = oracle.lib().nmo_standard_normal_stream(C.byref(cfg), k, count, exp)
        assert (bits(got[i]) == bits(exp)).all()
        assert words[i] == w
        slow += w - 2 * count
    assert slow > 0          # the slow path (rejections / tail) was exercised


@pytest.mark.parametrize("dim,dpl", [(1, 2), (10, 2), (64, 2), (127, 2), (128, 2), (129, 4), (200, 4), (300, 8),
                                      (512, 8), (513, 16), (1000, 16), (1024, 16), (100, 16)])
@pytest.mark.parametrize("kind", ["iid", "diag"])
def test_leapfrog_batch_bit_exact(oracle, dim, dpl, kind):
    """fused leapfrog + logp/grad (transformed_hamiltonian.rs:524-615) on ragged dims incl. padding edges."""
    L = N.load_library()
    rng = np.random.default_rng(dim * 7 + dpl)
    n = 9
    z, v, gz = rng.normal(size=(n, dim)), rng.normal(size=(n, dim)), rng.normal(size=(n, dim))
    sigma, mu = np.exp(rng.normal(size=(n, dim))), rng.normal(size=(n, dim)) * 3
    eps = rng.uniform(0.01, 0.5, n) * rng.choice([-1.0, 1.0], n)
    logdet, e0 = rng.normal(size=n), rng.normal(size=n) * 10
    logp = N.LogpSpec.iid_normal(dim, 3.0) if kind == "iid" else N.LogpSpec.diag_normal(np.exp(rng.normal(size=dim)))
    spec = logp.to_c()
    d = [dev(a) for a in (z, v, gz, sigma, mu, eps, logdet, e0)]
    outs = [torch.empty((n, dim), dtype=torch.float64, device="cuda") for _ in range(5)]
    souts = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    rc = L.nm_leapfrog_batch(C.byref(spec), n, dpl, *[ptr(t) for t in d], *[ptr(t) for t in outs],
                             *[ptr(t) for t in souts], None)
    assert rc == 0, L.nm_last_error()
    cfg = oracle.gpu_cfg()
    params = np.ascontiguousarray(logp.params)
    for i in range(n):
        eo = [np.empty(dim) for _ in range(5)]
        so = [C.c_double() for _ in range(3)]
        rc = oracle.lib().nmo_leapfrog(C.byref(cfg), logp.kind, dim, params, len(params), z[i].copy(), v[i].copy(),
                                       gz[i].copy(), sigma[i].copy(), mu[i].copy(), eps[i], logdet[i], e0[i],
                                       *eo, *[C.byref(x) for x in so])
        assert rc == 0
        for name, g, e in zip(("z", "v", "gz", "x", "gx"), outs, eo):
            assert (bits(g[i].cpu().numpy()) == bits(e)).all(), (name, i)
        for name, g, e in zip(("logp", "ke", "energy_error"), souts, so):
            assert bits(g[i].item()) == bits(e.value), (name, i, g[i].item(), e.value)


@pytest.mark.parametrize("dim,dpl", [(3, 2), (128, 2), (130, 4), (1024, 16)])
def test_turning_batch_bit_exact(oracle, dim, dpl):
    """U-turn criterion sums (scalar_prods3 semantics) in the engine's reduction order."""
    L = N.load_library()
    rng = np.random.default_rng(dim)
    n = 7
    zs, vs, ze, ve = (rng.normal(size=(n, dim)) for _ in range(4))
    out = torch.empty(2 * n, dtype=torch.float64, device="cuda")
    keep = [dev(a) for a in (zs, vs, ze, ve)]
    assert L.nm_turning_batch(n, dim, dpl, *[ptr(t) for t in keep], ptr(out), None) == 0
    got = out.cpu().numpy()
    cfg = oracle.gpu_cfg()
    zeros = np.zeros(dim)
    for i in range(n):
        o = np.empty(2)
        oracle.lib().nmo_scalar_prods3(C.byref(cfg), ze[i].copy(), zs[i].copy(), zeros, vs[i].copy(), ve[i].copy(), dim, o)
        assert (bits(got[2 * i:2 * i + 2]) == bits(o)).all()
        # and within the reference's own 32-ulp envelope of the SIMD-order sum (src/math/util.rs:916-926)
        o2 = np.empty(2)
        oracle.lib().nmo_scalar_prods3(C.byref(oracle.ref_cfg()), ze[i].copy(), zs[i].copy(), zeros, vs[i].copy(), ve[i].copy(), dim, o2)
        assert np.abs(o - o2).max() <= 1e-12 * max(1.0, np.abs(o2).max()) * dim ** 0.5


@pytest.mark.parametrize("kind,dim,dpl", [("funnel", 2, 2), ("funnel", 101, 2), ("funnel", 700, 16), ("schools", 10, 2)])
def test_leapfrog_batch_other_densities(oracle, kind, dim, dpl):
    """The densities this repo defines for BASELINE configs K3/K4 (funnel, non-centered 8 schools): fused leapfrog
    == oracle, bit for bit (same operation order on both sides)."""
    L = N.load_library()
    rng = np.random.default_rng(dim)
    n = 11
    z, v, gz = rng.normal(size=(n, dim)) * 0.7, rng.normal(size=(n, dim)), rng.normal(size=(n, dim))
    sigma, mu = np.exp(rng.normal(size=(n, dim)) * 0.3), rng.normal(size=(n, dim)) * 0.5
    eps = rng.uniform(0.01, 0.3, n) * rng.choice([-1.0, 1.0], n)
    logdet, e0 = rng.normal(size=n), rng.normal(size=n) * 10
    logp = N.LogpSpec.funnel(dim) if kind == "funnel" else N.LogpSpec.eight_schools()
    spec = logp.to_c()
    d = [dev(a) for a in (z, v, gz, sigma, mu, eps, logdet, e0)]
    outs = [torch.empty((n, dim), dtype=torch.float64, device="cuda") for _ in range(5)]
    souts = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    rc = L.nm_leapfrog_batch(C.byref(spec), n, dpl, *[ptr(t) for t in d], *[ptr(t) for t in outs],
                             *[ptr(t) for t in souts], None)
    assert rc == 0, L.nm_last_error()
    cfg = oracle.gpu_cfg()
    params = np.ascontiguousarray(logp.params if len(logp.params) else np.zeros(1))
    for i in range(n):
        eo = [np.empty(dim) for _ in range(5)]
        so = [C.c_double() for _ in range(3)]
        rc = oracle.lib().nmo_leapfrog(C.byref(cfg), logp.kind, dim, params, len(logp.params), z[i].copy(), v[i].copy(),
                                       gz[i].copy(), sigma[i].copy(), mu[i].copy(), eps[i], logdet[i], e0[i],
                                       *eo, *[C.byref(x) for x in so])
        assert rc == 0
        for name, g, e in zip(("z", "v", "gz", "x", "gx"), outs, eo):
            assert (bits(g[i].cpu().numpy()) == bits(e)).all(), (name, i)
        for name, g, e in zip(("logp", "ke", "energy_error"), souts, so):
            assert bits(g[i].item()) == bits(e.value), (name, i, g[i].item(), e.value)


def test_issue_rate_probe_reports_a_plausible_figure():
    """nm_probe_issue (ABI v16): nanoseconds per dependent v_fma_f64 of a lone wavefront, one wavefront per SIMD — the box calibration bench.py
    prints beside every config.  A 64-lane f64 instruction occupies a SIMD for 4 cycles: between 4 and ~10 cycles at 1.4 - 2.5 GHz."""
    import ctypes as C
    from nuts_rs_amd import _lib
    L = _lib.load()
    ns = C.c_double()
    _lib.check(L.nm_probe_issue(0, 1 << 18, C.byref(ns)))
    assert 1.5 < ns.value < 8.0, ns.value
    one = C.c_double()
    _lib.check(L.nm_probe_issue(64, 1 << 18, C.byref(one)))      # 64 wavefronts on the whole chip: the same chain, no neighbours
    assert 1.5 < one.value < 8.0          # (no ordering against the full-chip figure: a short launch on an idle device may run at another clock)
    assert L.nm_probe_issue(0, 8, C.byref(ns)) != 0                # a chain shorter than one loop trip is refused
