"""ctypes binding of the CPU oracle (oracle/libnuts_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (nuts_rs_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnuts_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("nuts_oracle.cpp", "nmo_math.hpp", "nmo_rng.hpp", "nmo_zig_tables.hpp", "nmo_nuts.hpp")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class Settings(C.Structure):
    """Field-for-field the reference's DiagNutsSettings (see include/nuts_amd.h nm_settings)."""
    _fields_ = [
        ("num_tune", C.c_uint64), ("num_draws", C.c_uint64), ("maxdepth", C.c_uint64), ("mindepth", C.c_uint64),
        ("max_energy_error", C.c_double), ("check_turning", C.c_uint64), ("extra_doublings", C.c_uint64),
        ("seed", C.c_uint64), ("num_chains", C.c_uint64),
        ("store_gradient", C.c_uint64), ("store_unconstrained", C.c_uint64), ("store_transformed", C.c_uint64),
        ("store_divergences", C.c_uint64),
        ("has_target_integration_time", C.c_uint64), ("target_integration_time", C.c_double),
        ("early_window", C.c_double), ("step_size_window", C.c_double),
        ("mass_matrix_switch_freq", C.c_uint64), ("early_mass_matrix_switch_freq", C.c_uint64),
        ("mass_matrix_update_freq", C.c_uint64), ("mass_matrix_window_growth", C.c_double),
        ("store_mass_matrix", C.c_uint64), ("use_grad_based_estimate", C.c_uint64),
        ("target_accept", C.c_double), ("initial_step", C.c_double), ("has_jitter", C.c_uint64),
        ("jitter", C.c_double), ("step_size_method", C.c_uint64), ("fixed_step_size", C.c_double),
        ("da_k", C.c_double), ("da_t0", C.c_double), ("da_gamma", C.c_double), ("da_max_step_size", C.c_double),
        ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_epsilon", C.c_double),
        ("adam_learning_rate", C.c_double),
        ("adaptation", C.c_uint64), ("lr_gamma", C.c_double), ("lr_eigval_cutoff", C.c_double),
        ("freeze_transform", C.c_uint64), ("trajectory_kind", C.c_uint64),
        ("sampler", C.c_uint64), ("mclmc_step_size", C.c_double), ("momentum_decoherence_length", C.c_double),
        ("subsample_frequency", C.c_double), ("dynamic_step_size", C.c_uint64), ("mclmc_trajectory_kind", C.c_uint64),
        ("trajectory_switch_fraction", C.c_double),
    ]


ADAPT_DIAG, ADAPT_LOW_RANK = 0, 1
TRAJ_EUCLIDEAN, TRAJ_EXACT_NORMAL, TRAJ_MICROCANONICAL = 0, 1, 2
SAMPLER_NUTS, SAMPLER_MCLMC = 0, 1
MCLMC_MICROCANONICAL, MCLMC_EUCLIDEAN, MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL = 0, 1, 2

STATS_DTYPE = np.dtype([
    ("draw", "<u8"), ("chain", "<u8"), ("depth", "<u8"), ("maxdepth_reached", "<u8"), ("diverging", "<u8"),
    ("tuning", "<u8"), ("n_steps", "<u8"), ("index_in_trajectory", "<i8"), ("transformation_index", "<i8"),
    ("step_size", "<f8"), ("step_size_bar", "<f8"), ("mean_tree_accept", "<f8"), ("mean_tree_accept_sym", "<f8"),
    ("max_energy_error", "<f8"), ("logp", "<f8"), ("energy", "<f8"), ("energy_error", "<f8"),
    ("fisher_distance", "<f8"), ("divergence_energy_error", "<f8"), ("chain_status", "<u8"),
    ("transformation_update_id", "<i8"), ("num_eigenvalues", "<u8"), ("energy_change", "<f8"), ("average_step_size", "<f8"),
])

VECTOR_STATS = ("gradient", "transformed_position", "transformed_gradient", "mass_matrix_inv", "transformation_mu",
                "divergence_start", "divergence_start_gradient", "divergence_end", "mass_matrix_eigvals")


class DrawVectors(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in VECTOR_STATS]


class MathCfg(C.Structure):
    _fields_ = [("detmath", C.c_int64), ("reduce_mode", C.c_int64), ("simd_lanes", C.c_int64),
                ("gpu_threads", C.c_int64), ("gpu_slice", C.c_int64), ("lr_seq_dots", C.c_int64), ("tile_order", C.c_int64)]


# LowRankMassMatrixStrategy's dense linear algebra as a callback (oracle/lowrank.py implements it with LAPACK)
ESTIMATOR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                           C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                           C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


# CpuLogpFunc::logp shape of the oracle's LOGP_HOST_CALLBACK: (ctx, dim, x*, grad*, logp*) -> 0 ok, 1 recoverable, 2 fatal
HOST_LOGP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


class RunExtras(C.Structure):
    _fields_ = [("stds", C.c_void_p), ("mean", C.c_void_p), ("vals", C.c_void_p), ("vecs", C.c_void_p),
                ("mu_lr", C.c_void_p), ("n_eig", C.c_uint64), ("per_chain", C.c_uint64),
                ("estimator", ESTIMATOR_FN), ("estimator_ctx", C.c_void_p)]


REDUCE_REF_SIMD, REDUCE_GPU = 0, 1
LOGP_IID_NORMAL, LOGP_DIAG_NORMAL, LOGP_FUNNEL, LOGP_EIGHT_SCHOOLS, LOGP_MVN_PREC = 0, 1, 2, 3, 4


def ref_cfg(simd_lanes=4):
    """What the reference does on this box: libm transcendentals, pulp-style 4-accumulator SIMD sums."""
    return MathCfg(0, REDUCE_REF_SIMD, simd_lanes, 64, 0, 0, 0)


def gpu_cfg(gpu_threads=64, lr_seq_dots=0, gpu_slice=0):
    """The arithmetic contract of the HIP engine: restated exp/ln, fixed lane-tiled reduction order.
    lr_seq_dots=1: the matrix-core kernel for shared matrices (its U'v products are sequential fma chains).
    lr_seq_dots=2: the lockstep matrix-core kernel (nuts_lockstep.hpp): the same products, and every reduction over dim in that
    kernel's stripe order from the moment the shared transformation is set (set_position runs on the wave kernels before).
    gpu_slice=4096: a chain wider than one block (dim > 4096) — slice totals added in slice order."""
    return MathCfg(1, REDUCE_GPU, 4, gpu_threads, gpu_slice, lr_seq_dots, 0)


_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.nmo_settings_default.argtypes = [C.POINTER(Settings)]
    L.nmo_settings_size.restype = C.c_uint64
    L.nmo_draw_stats_size.restype = C.c_uint64
    L.nmo_chain_key.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
    L.nmo_init_position_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _dp]
    L.nmo_chain_create.restype = C.c_void_p
    L.nmo_chain_create.argtypes = [C.POINTER(Settings), C.c_int64, C.c_uint64, _dp, C.c_uint64,
                                   C.POINTER(MathCfg), C.c_uint64, C.c_void_p]
    L.nmo_chain_create_callback.restype = C.c_void_p
    L.nmo_chain_create_callback.argtypes = [C.POINTER(Settings), C.c_uint64, HOST_LOGP_FN, C.c_void_p, C.POINTER(MathCfg),
                                            C.c_uint64, C.c_void_p]
    L.nmo_chain_destroy.argtypes = [C.c_void_p]
    L.nmo_chain_set_position.argtypes = [C.c_void_p, _dp]
    L.nmo_chain_set_position.restype = C.c_int
    L.nmo_chain_draw.argtypes = [C.c_void_p, _dp, C.c_void_p]
    L.nmo_chain_draw.restype = C.c_int
    L.nmo_chain_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.nmo_run.restype = C.c_int
    L.nmo_run.argtypes = [C.POINTER(Settings), C.c_int64, C.c_uint64, _dp, C.c_uint64, C.POINTER(MathCfg),
                          C.c_uint64, C.c_uint64, _dp, C.c_uint64, C.c_void_p, C.c_void_p,
                          C.POINTER(C.c_uint64), C.c_uint64, C.c_void_p]
    L.nmo_run_ex.restype = C.c_int
    L.nmo_run_ex.argtypes = L.nmo_run.argtypes + [C.POINTER(RunExtras)]
    L.nmo_settings_default_low_rank.argtypes = [C.POINTER(Settings)]
    L.nmo_settings_default_mclmc.argtypes = [C.POINTER(Settings)]
    L.nmo_chain_set_estimator.argtypes = [C.c_void_p, ESTIMATOR_FN, C.c_void_p]
    L.nmo_chain_set_transform.restype = C.c_int
    L.nmo_chain_set_transform.argtypes = [C.c_void_p, _dp, _dp, C.c_uint64, _dp, _dp, _dp]
    L.nmo_lowrank_kat.restype = C.c_int
    L.nmo_lowrank_kat.argtypes = [C.POINTER(MathCfg), C.c_uint64, _dp, _dp, _dp, C.c_uint64, _dp, _dp, _dp, C.c_int64,
                                  _dp, _dp, _dp, C.POINTER(C.c_double), C.POINTER(C.c_double), _dp,
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.nmo_adam_sequence.restype = None
    L.nmo_adam_sequence.argtypes = [C.POINTER(MathCfg), C.c_double, C.POINTER(C.c_double), C.c_uint64, C.c_double,
                                    C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
    L.nmo_run_timed.restype = C.c_int
    L.nmo_run_timed.argtypes = [C.POINTER(Settings), C.c_int64, C.c_uint64, _dp, C.c_uint64, C.POINTER(MathCfg),
                                C.c_uint64, C.c_uint64, _dp, C.c_uint64, C.c_uint64, C.c_uint64,
                                C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                C.POINTER(C.c_uint64)]
    L.nmo_run_wall.restype = C.c_int
    L.nmo_run_wall.argtypes = L.nmo_run_timed.argtypes
    L.nmo_logaddexp.restype = C.c_double
    L.nmo_logaddexp.argtypes = [C.POINTER(MathCfg), C.c_double, C.c_double]
    L.nmo_scalar_fn.restype = C.c_double
    L.nmo_scalar_fn.argtypes = [C.POINTER(MathCfg), C.c_int64, C.c_double, C.c_double]
    L.nmo_chacha_block.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p]
    L.nmo_rng_words.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.nmo_seed_from_u64.argtypes = [C.c_uint64, C.c_void_p]
    L.nmo_standard_normal_stream.restype = C.c_uint64
    L.nmo_standard_normal_stream.argtypes = [C.POINTER(MathCfg), C.c_void_p, C.c_uint64, _dp]
    L.nmo_zig_tables.argtypes = [_dp, _dp]
    L.nmo_rng_samples.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_uint64, _dp]
    L.nmo_vector_dot.restype = C.c_double
    L.nmo_vector_dot.argtypes = [C.POINTER(MathCfg), _dp, _dp, C.c_uint64]
    L.nmo_scalar_prods3.argtypes = [C.POINTER(MathCfg), _dp, _dp, _dp, _dp, _dp, C.c_uint64, _dp]
    L.nmo_axpy.argtypes = [_dp, _dp, C.c_double, C.c_uint64]
    L.nmo_axpy_out.argtypes = [_dp, _dp, C.c_double, _dp, C.c_uint64]
    L.nmo_multiply.argtypes = [_dp, _dp, _dp, C.c_uint64]
    L.nmo_logp.restype = C.c_int
    L.nmo_logp.argtypes = [C.POINTER(MathCfg), C.c_int64, C.c_uint64, _dp, C.c_uint64, _dp, _dp,
                           C.POINTER(C.c_double)]
    L.nmo_diag_kat.restype = C.c_int
    L.nmo_diag_kat.argtypes = [C.POINTER(MathCfg), C.c_uint64] + [_dp] * 6 + [_dp, _dp, C.POINTER(C.c_double),
                               C.POINTER(C.c_double), _dp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                               _dp, _dp, _dp]
    L.nmo_leapfrog.restype = C.c_int
    L.nmo_leapfrog.argtypes = [C.POINTER(MathCfg), C.c_int64, C.c_uint64, _dp, C.c_uint64, _dp, _dp, _dp, _dp, _dp,
                               C.c_double, C.c_double, C.c_double, _dp, _dp, _dp, _dp, _dp,
                               C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    assert L.nmo_settings_size() == C.sizeof(Settings)
    assert L.nmo_draw_stats_size() == STATS_DTYPE.itemsize
    _lib = L
    return L


def default_settings(low_rank=False, mclmc=False, **overrides):
    """DiagNutsSettings::default(), with low_rank=True LowRankNutsSettings::default() (src/sampler.rs:630-642), with mclmc=True
    DiagMclmcSettings::default() (src/sampler.rs:368-374)."""
    s = Settings()
    L = lib()
    (L.nmo_settings_default_mclmc if mclmc else L.nmo_settings_default_low_rank if low_rank else L.nmo_settings_default)(C.byref(s))
    for k, v in overrides.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


def chain_key(seed, chain_id):
    key = (C.c_uint8 * 32)()
    lib().nmo_chain_key(seed, chain_id, key)
    return bytes(key)


def init_positions_uniform(seed, chain_offset, n_chains, dim):
    out = np.empty((n_chains, dim))
    row = np.empty(dim)
    for c in range(n_chains):
        lib().nmo_init_position_uniform(seed, chain_offset + c, dim, row)
        out[c] = row
    return out


class Chain:
    """One oracle chain: the reference's `settings.new_chain(chain, math, rng)` + set_position + draw."""

    def __init__(self, settings, kind, dim, params, cfg, chain_id=0, key=None, callback=None):
        """callback: an oracle.HOST_LOGP_FN — the density is then LOGP_HOST_CALLBACK (`kind` / `params` unused)."""
        self.dim = dim
        self.cfg = cfg
        self.params = np.ascontiguousarray(params, dtype=np.float64)
        if key is None:
            key = chain_key(settings.seed, chain_id)
        self._key = (C.c_uint8 * 32).from_buffer_copy(key)
        self._cb = callback
        if callback is not None:
            self._h = lib().nmo_chain_create_callback(C.byref(settings), dim, callback, None, C.byref(cfg), chain_id, self._key)
        else:
            self._h = lib().nmo_chain_create(C.byref(settings), kind, dim, self.params, len(self.params),
                                             C.byref(cfg), chain_id, self._key)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().nmo_chain_destroy(self._h)
            self._h = None

    def set_position(self, x0):
        return lib().nmo_chain_set_position(self._h, np.ascontiguousarray(x0, dtype=np.float64))

    def draw(self):
        pos = np.empty(self.dim)
        st = np.zeros(1, dtype=STATS_DTYPE)
        rc = lib().nmo_chain_draw(self._h, pos, st.ctypes.data)
        return pos, st[0], rc

    def state(self):
        x, gx, sd, mu = (np.empty(self.dim) for _ in range(4))
        eps = C.c_double()
        pos = C.c_uint64()
        lib().nmo_chain_get_state(self._h, x.ctypes.data, gx.ctypes.data, sd.ctypes.data, mu.ctypes.data,
                                  C.byref(eps), C.byref(pos))
        return dict(x=x, gx=gx, stds=sd, mean=mu, step_size=eps.value, rng_pos=pos.value)


def lowrank_kat(cfg, precision_diag, stds, mean, vals, vecs, mu_lr, x, which=-1):
    """LowRankMassMatrix::update(...) then init_from_untransformed_position(x) + round trip (which = -1), or one of
    compute_transformed_position / compute_untransformed_position / compute_transformed_gradient (which = 0 / 1 / 2)."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    dim = len(stds)
    vals = f(vals)
    vecs = f(vecs).reshape(len(vals), dim) if len(vals) else np.zeros((0, dim))
    z, gz, x_rt = np.empty(dim), np.empty(dim), np.empty(dim)
    logp, logdet, logp_rt, logdet_rt = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    rc = lib().nmo_lowrank_kat(C.byref(cfg), dim, f(precision_diag), f(stds), f(mean), len(vals),
                               vals if len(vals) else np.zeros(1), vecs if len(vals) else np.zeros(1), f(mu_lr), which, f(x),
                               z, gz, C.byref(logp), C.byref(logdet), x_rt, C.byref(logp_rt), C.byref(logdet_rt))
    return dict(rc=rc, z=z, gz=gz, x_rt=x_rt, logp=logp.value, logdet=logdet.value, logp_rt=logp_rt.value,
                logdet_rt=logdet_rt.value)


def traj_kat(cfg, op, a=None, b=None, c=None, eps=0.0):
    """The non-Euclidean trajectory kinds' vector primitives (nmo_traj_kat): op "flow" (pos, vel) -> (pos_out, vel),
    "grad_flow" (pos, grad, vel) -> vel_out, "esh" (gradient, momentum) -> (momentum, dKE), "normalize" (v) -> v,
    "sincos" -> (sin eps, cos eps)."""
    f = lambda x: np.ascontiguousarray(x if x is not None else np.zeros(1), dtype=np.float64)
    opn = {"flow": 0, "grad_flow": 1, "esh": 2, "normalize": 3, "sincos": 4}[op]
    n = len(a) if a is not None else 2
    o1, o2, sc = np.empty(max(n, 2)), np.empty(max(n, 2)), C.c_double()
    P = C.POINTER(C.c_double)
    rc = lib().nmo_traj_kat(C.byref(cfg), C.c_int64(opn), C.c_uint64(n if a is not None else 0), f(a).ctypes.data_as(P),
                            f(b).ctypes.data_as(P), f(c).ctypes.data_as(P), C.c_double(eps), o1.ctypes.data_as(P),
                            o2.ctypes.data_as(P), C.byref(sc))
    assert rc == 0
    if op == "flow":
        return o1[:n], o2[:n]
    if op == "esh":
        return o1[:n], sc.value
    if op == "sincos":
        return o1[0], o1[1]
    return o1[:n]


def adam_sequence(initial_step, accept, target, beta1, beta2, epsilon, learning_rate, cfg=None):
    acc = np.ascontiguousarray(accept, dtype=np.float64)
    out = np.empty(len(acc))
    cfg = cfg or ref_cfg()
    lib().nmo_adam_sequence(C.byref(cfg), C.c_double(initial_step), acc.ctypes.data_as(C.POINTER(C.c_double)), len(acc),
                            C.c_double(target), C.c_double(beta1), C.c_double(beta2), C.c_double(epsilon),
                            C.c_double(learning_rate), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def _transform_arrays(transform, n_chains, dim):
    """(stds, mean, vals, vecs, mu_lr) -> contiguous arrays + (n_eig, per_chain).  vecs is [n_eig][dim] (one eigenvector
    per row), or [n_chains][n_eig][dim] with every other entry carrying a leading chain axis too."""
    stds, mean, vals, vecs, mu = (np.ascontiguousarray(a, dtype=np.float64) for a in transform)
    per_chain = int(stds.ndim == 2)
    n_eig = vals.shape[-1] if vals.size else 0
    want = ((n_chains, dim) if per_chain else (dim,))
    assert stds.shape == want and mean.shape == want and mu.shape == want
    assert vecs.size == (n_chains if per_chain else 1) * n_eig * dim
    return (stds, mean, vals, vecs, mu), n_eig, per_chain


def run(settings, kind, dim, params, cfg, n_chains, x0, n_draws, chain_offset=0, n_threads=1,
        want_positions=True, want_stats=True, vectors=None, transform=None, estimator=None):
    """Many chains on host threads (reference Sampler structure).  Returns positions [draws][chains][dim], stats, steps.

    vectors: optional dict that receives the vector-valued statistics, [draws][chains][dim] each (NaN-filled; event
    rows are written only on the draws where the event happens)."""
    params = np.ascontiguousarray(params, dtype=np.float64)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    pos = np.empty((n_draws, n_chains, dim)) if want_positions else None
    st = np.zeros((n_draws, n_chains), dtype=STATS_DTYPE) if want_stats else None
    steps = C.c_uint64()
    dv = None
    if vectors is not None:
        dv = DrawVectors()
        for k in VECTOR_STATS:
            vectors[k] = np.full((n_draws, n_chains, dim), np.nan)
            setattr(dv, k, vectors[k].ctypes.data)
    ex = RunExtras()
    keep = None
    if transform is not None:
        keep, ex.n_eig, ex.per_chain = _transform_arrays(transform, n_chains, dim)
        ex.stds, ex.mean, ex.vals, ex.vecs, ex.mu_lr = (a.ctypes.data for a in keep)
    if estimator is not None:
        ex.estimator = estimator
        n_threads = 1               # a Python callback: one thread
    failed = lib().nmo_run_ex(C.byref(settings), kind, dim, params, len(params), C.byref(cfg), n_chains, chain_offset,
                              x0, n_draws, pos.ctypes.data if pos is not None else None,
                              st.ctypes.data if st is not None else None, C.byref(steps), n_threads,
                              C.byref(dv) if dv is not None else None, C.byref(ex))
    return pos, st, steps.value, failed


def run_timed(settings, kind, dim, params, cfg, n_chains, x0, n_warm, n_draws, chain_offset=0, n_threads=1):
    """CPU-baseline leg: per-chain tasks on n_threads host threads; returns post-warm-up (cpu_seconds_sum, steps)."""
    params = np.ascontiguousarray(params, dtype=np.float64)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    ws, secs = C.c_double(), C.c_double()
    wsteps, steps = C.c_uint64(), C.c_uint64()
    failed = lib().nmo_run_timed(C.byref(settings), kind, dim, params, len(params), C.byref(cfg), n_chains,
                                 chain_offset, x0, n_warm, n_draws, n_threads, C.byref(ws), C.byref(wsteps),
                                 C.byref(secs), C.byref(steps))
    return dict(failed=failed, warm_cpu_seconds=ws.value, warm_steps=wsteps.value, cpu_seconds=secs.value,
                steps=steps.value)


def run_wall(settings, kind, dim, params, cfg, n_chains, x0, n_warm, n_draws, chain_offset=0, n_threads=1):
    """CPU-baseline leg, wall clock: phase 1 (set_position + n_warm draws of every chain) and phase 2 (n_draws draws of
    every chain), each with one chain per task on n_threads host threads; returns the wall seconds and steps of both."""
    params = np.ascontiguousarray(params, dtype=np.float64)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    ws, secs = C.c_double(), C.c_double()
    wsteps, steps = C.c_uint64(), C.c_uint64()
    failed = lib().nmo_run_wall(C.byref(settings), kind, dim, params, len(params), C.byref(cfg), n_chains,
                                chain_offset, x0, n_warm, n_draws, n_threads, C.byref(ws), C.byref(wsteps),
                                C.byref(secs), C.byref(steps))
    return dict(failed=failed, warm_wall_seconds=ws.value, warm_steps=wsteps.value, wall_seconds=secs.value,
                steps=steps.value)
