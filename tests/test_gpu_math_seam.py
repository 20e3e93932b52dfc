"""The per-vector `Math` seam (reference src/math/math.rs:15-314) on device vectors: every hot-path method
M1-M15 of SURVEY §8(a) against the oracle's primitive (element-wise: the reference's formula, bit for bit; reductions: the
oracle's engine-order sums, bit for bit)."""
import ctypes as C
import math

import numpy as np
import pytest

import nuts_rs_amd as N
from nuts_rs_amd import _lib

pytestmark = pytest.mark.gpu


class DevMath:
    def __init__(self, logp):
        self.L = _lib.load()
        self._spec = logp.to_c()
        self.h = C.c_void_p()
        assert self.L.nm_math_create(C.byref(self._spec), C.byref(self.h)) == 0, self.L.nm_math_last_error()
        self.dim = int(self.L.nm_math_dim(self.h))
        self.threads = int(self.L.nm_math_threads(self.h))
        self.vecs = []

    def vec(self, data=None):
        v = C.c_void_p()
        assert self.L.nm_vec_new(self.h, C.byref(v)) == 0
        self.vecs.append(v)
        if data is not None:
            a = np.ascontiguousarray(data, dtype=np.float64)
            assert self.L.nm_vec_read_from_slice(self.h, v, a.ctypes.data) == 0
        return v

    def get(self, v):
        out = np.empty(self.dim)
        assert self.L.nm_vec_write_to_slice(self.h, v, out.ctypes.data) == 0
        return out

    def close(self):
        for v in self.vecs:
            self.L.nm_vec_free(v)
        self.L.nm_math_destroy(self.h)


def same(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return ((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))).all()


@pytest.mark.parametrize("dim", [1, 10, 101, 128, 300, 1024, 1500, 4000])
def test_math_methods_bit_exact(oracle, dim):
    rng = np.random.default_rng(dim)
    M = DevMath(N.LogpSpec.iid_normal(dim, 3.0))
    L, h, O = M.L, M.h, oracle.lib()
    cfg = oracle.gpu_cfg(M.threads)
    x, y, z, p, q = (rng.normal(size=dim) * np.exp(rng.normal(0, 2, dim)) for _ in range(5))
    vx, vy, vz, vp_, vq, out = M.vec(x), M.vec(y), M.vec(z), M.vec(p), M.vec(q), M.vec()
    a = float(rng.normal())
    fma = np.vectorize(math.fma) if hasattr(math, "fma") else None
    # M1 / M2 / M3
    assert L.nm_vec_axpy_out(h, vx, vy, a, out) == 0
    want = np.empty(dim); O.nmo_axpy_out(x, y, a, want, dim)
    assert same(M.get(out), want)
    y2 = y.copy(); O.nmo_axpy(x, y2, a, dim)
    vy2 = M.vec(y)
    assert L.nm_vec_axpy(h, vx, vy2, a) == 0 and same(M.get(vy2), y2)
    assert L.nm_vec_array_mult(h, vx, vy, out) == 0 and same(M.get(out), x * y)
    vin = M.vec(x)
    assert L.nm_vec_array_mult(h, vin, vy, vin) == 0 and same(M.get(vin), x * y)           # array_mult_inplace
    # M4 / M5 / M11 / M15: reductions in the engine's order
    r = C.c_double()
    assert L.nm_vec_array_vector_dot(h, vx, vy, C.byref(r)) == 0
    assert same([r.value], [O.nmo_vector_dot(C.byref(cfg), x, y, dim)])
    r2 = (C.c_double * 2)()
    assert L.nm_vec_scalar_prods3(h, vx, vy, vz, vp_, vq, r2) == 0
    w2 = np.empty(2); O.nmo_scalar_prods3(C.byref(cfg), x, y, z, p, q, dim, w2)
    assert same(list(r2), w2)
    pos_ = np.abs(x) + 1e-3
    vpos = M.vec(pos_)
    assert L.nm_vec_array_sum_ln(h, vpos, C.byref(r)) == 0
    c1 = oracle.Settings  # noqa: F841
    want_ln = oracle_sum(oracle, cfg, [oracle.lib().nmo_scalar_fn(C.byref(cfg), 1, float(v), 0.0) for v in pos_])
    assert same([r.value], [want_ln])
    assert L.nm_vec_sq_norm_sum(h, vx, vy, C.byref(r)) == 0
    assert same([r.value], [oracle_sum(oracle, cfg, (x + y) * (x + y))])
    # M7
    mean, var = rng.normal(size=dim), np.abs(rng.normal(size=dim))
    vm, vv = M.vec(mean), M.vec(var)
    assert L.nm_vec_array_update_variance(h, vm, vv, vx, 0.125) == 0
    d = x - mean
    assert same(M.get(vm), mean + d * 0.125) and same(M.get(vv), var + d * d)
    # M8 / M9 / M10 with invalid entries
    dv, gv = np.abs(rng.normal(size=dim)) + 1e-3, np.abs(rng.normal(size=dim)) + 1e-3
    if dim > 3:
        dv[1], gv[2], dv[3] = 0.0, 0.0, np.inf
    old_inv, old_std = rng.uniform(1, 2, dim), rng.uniform(1, 2, dim)
    for has_fill in (0, 1):
        vi, vs = M.vec(old_inv), M.vec(old_std)
        assert L.nm_vec_array_update_var_inv_std_draw_grad(h, vi, vs, M.vec(dv), M.vec(gv), has_fill, 1.0, 1e-20, 1e20) == 0
        with np.errstate(all="ignore"):
            val = np.sqrt(dv / gv)
        bad = ~np.isfinite(val) | (val == 0)
        cl = np.clip(val, 1e-20, 1e20)
        ws = np.where(bad, np.where(has_fill, 1.0, old_std), np.sqrt(cl))
        wi = np.where(bad, np.where(has_fill, 1.0, old_inv), np.sqrt(1.0 / cl))
        assert same(M.get(vs), ws) and same(M.get(vi), wi)
        vi, vs = M.vec(old_inv), M.vec(old_std)
        assert L.nm_vec_array_update_var_inv_std_draw(h, vi, vs, M.vec(dv), 0.25, has_fill, 1.0, 1e-20, 1e20) == 0
        val = dv * 0.25
        bad = ~np.isfinite(val) | (val == 0)
        cl = np.clip(val, 1e-20, 1e20)
        assert same(M.get(vs), np.where(bad, np.where(has_fill, 1.0, old_std), np.sqrt(cl)))
        assert same(M.get(vi), np.where(bad, np.where(has_fill, 1.0, old_inv), np.sqrt(1.0 / cl)))
    g = x.copy()
    if dim > 2:
        g[0], g[1] = 0.0, np.inf
    vi, vs = M.vec(), M.vec()
    assert L.nm_vec_array_update_var_inv_std_grad(h, vi, vs, M.vec(g), 1.0, 1e-20, 1e20) == 0
    val = 1.0 / np.clip(np.abs(g), 1e-20, 1e20)
    val = np.where(np.isfinite(val), val, 1.0)
    assert same(M.get(vs), np.sqrt(val)) and same(M.get(vi), np.sqrt(1.0 / val))
    # M12 / M13
    flag = C.c_uint64()
    assert L.nm_vec_array_all_finite(h, vx, 0, C.byref(flag)) == 0 and flag.value == 1
    assert L.nm_vec_array_all_finite(h, M.vec(g), 0, C.byref(flag)) == 0 and flag.value == (0 if dim > 2 else 1)
    zero_in = x.copy(); zero_in[dim // 2] = 0.0
    assert L.nm_vec_array_all_finite(h, M.vec(zero_in), 1, C.byref(flag)) == 0 and flag.value == 0
    assert L.nm_vec_fill_array(h, out, 2.5) == 0 and same(M.get(out), np.full(dim, 2.5))
    assert L.nm_vec_array_recip(h, vpos, out) == 0 and same(M.get(out), 1.0 / pos_)
    assert L.nm_vec_copy_into(h, vq, out) == 0 and same(M.get(out), q)
    # M14: logp_array of the iid normal (benches/sample.rs:49-62)
    grad, lp, st = M.vec(), C.c_double(), C.c_uint64(7)
    assert L.nm_vec_logp_array(h, vx, grad, C.byref(lp), C.byref(st)) == 0 and st.value == 0
    og, olp = np.empty(dim), C.c_double()
    assert O.nmo_logp(C.byref(cfg), oracle.LOGP_IID_NORMAL, dim, np.array([3.0]), 1, x, og, C.byref(olp)) == 0
    assert same(M.get(grad), og) and same([lp.value], [olp.value])
    # M6: array_gaussian = stds * the chain generator's next `dim` standard normals, stream position advanced
    if dim <= 1024:
        key = bytes(range(7, 39))
        kb = (C.c_uint8 * 32).from_buffer_copy(key)
        pos0 = C.c_uint64(0)
        stds = np.exp(rng.normal(size=dim))
        assert L.nm_vec_array_gaussian(h, kb, C.byref(pos0), out, M.vec(stds)) == 0
        normals = np.empty(dim)
        words = O.nmo_standard_normal_stream(C.byref(cfg), kb, dim, normals)
        assert same(M.get(out), stds * normals) and pos0.value == words
    M.close()


def oracle_sum(oracle, cfg, terms):
    """sum of `terms` in the engine's reduction order: the oracle's sum_terms through nmo_vector_dot with ones would fuse;
    use the diag-normal log-det path instead: sum_terms is what IidNormal's logp uses, reachable as sum t_i = -2 logp of
    x_i = mu + sqrt(t_i)... simpler: the oracle exports the ordered sum through nmo_logp of a funnel-free density? No —
    use vector_dot with exact products: t_i * 1.0 accumulates as fma(t_i, 1.0, acc) = acc + t_i exactly."""
    t = np.ascontiguousarray(terms, dtype=np.float64)
    return oracle.lib().nmo_vector_dot(C.byref(cfg), t, np.ones(len(t)), len(t))


@pytest.mark.parametrize("dim", [2, 10, 101, 300, 1024, 4000])
def test_trajectory_kind_methods_bit_exact(oracle, dim):
    """std_norm_flow / std_norm_grad_flow(_inplace) / esh_momentum_update / array_normalize (src/math/math.rs:155-200) against
    the oracle's primitives in the engine's arithmetic."""
    rng = np.random.default_rng(100 + dim)
    M = DevMath(N.LogpSpec.iid_normal(dim, 0.0))
    L, h = M.L, M.h
    cfg = oracle.gpu_cfg(M.threads)
    p, v, g = rng.normal(size=dim), rng.normal(size=dim), rng.normal(size=dim) * 2
    for eps in (0.37, -1.9, 3.0):
        vp_, vv, vo = M.vec(p), M.vec(v), M.vec()
        assert L.nm_vec_std_norm_flow(h, vp_, vo, vv, eps) == 0
        po, vn = oracle.traj_kat(cfg, "flow", p, v, eps=eps)
        assert same(M.get(vo), po) and same(M.get(vv), vn)
        vg, vv2, vout = M.vec(g), M.vec(v), M.vec()
        assert L.nm_vec_std_norm_grad_flow(h, vp_, vg, vv2, vout, eps) == 0
        want = oracle.traj_kat(cfg, "grad_flow", p, g, v, eps=eps)
        assert same(M.get(vout), want)
        assert L.nm_vec_std_norm_grad_flow(h, vp_, vg, vv2, vv2, eps) == 0 and same(M.get(vv2), want)     # _inplace
    vu = M.vec(v)
    assert L.nm_vec_array_normalize(h, vu) == 0
    un = oracle.traj_kat(cfg, "normalize", v)
    assert same(M.get(vu), un)
    for step in (0.2, -0.45, 1.7):
        vm, dke = M.vec(un), C.c_double()
        assert L.nm_vec_esh_momentum_update(h, M.vec(g), vm, step, C.byref(dke)) == 0
        m_o, dke_o = oracle.traj_kat(cfg, "esh", g, un, eps=step)
        assert same(M.get(vm), m_o) and same([dke.value], [dke_o])
    M.close()
