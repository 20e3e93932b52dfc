"""Host-side mirror of the reference's per-chain driver seam, batched over many chains.

Reference interface being mirrored (pymc-devs/nuts-rs 0.18.3):
  * `DiagNutsSettings` and its nested option structs           src/sampler.rs:199-239, :507-531, :630-634
  * `Settings::new_chain(chain, math, rng) -> impl Chain`       src/sampler.rs:53-63, :745-772
  * `Chain::{set_position, draw, dim}`                          src/chain.rs:24-42, :137-188
  * `Progress`                                                  src/sampler.rs:165-174
Names, argument meaning and error behaviour follow the reference; the only structural change is that one
`ChainBatch` stands for `n_chains` independent `NutsChain`s that advance together on one MI355X.
All arithmetic happens in libnuts_amd.so (HIP); this file only marshals arguments.
"""
import ctypes as C
import os
from dataclasses import dataclass, field, fields
from typing import Optional

import numpy as np

from . import _lib
from ._lib import (NmDrawOutputs, NmEngineConfig, NmLogpSpec, NmSettings, NutsAmdError, STATS_DTYPE, VECTOR_STATS,
                   check)

LOGP_IID_NORMAL, LOGP_DIAG_NORMAL, LOGP_FUNNEL, LOGP_EIGHT_SCHOOLS, LOGP_MVN_PREC, LOGP_MODULE, LOGP_HOST_CALLBACK = 0, 1, 2, 3, 4, 5, 6
STEP_DUAL_AVERAGE, STEP_ADAM, STEP_FIXED = 0, 1, 2
ADAPT_DIAG, ADAPT_LOW_RANK = 0, 1
SAMPLER_NUTS, SAMPLER_MCLMC = 0, 1


@dataclass
class DualAverageOptions:          # src/stepsize/dual_avg.rs:12-31
    k: float = 0.75
    t0: float = 10.0
    gamma: float = 0.05
    max_step_size: float = float(np.pi)


@dataclass
class AdamOptions:                 # src/stepsize/adam.rs:12-34
    beta1: float = 0.9
    beta2: float = 0.999
    epsilon: float = 1e-8
    learning_rate: float = 0.05


@dataclass
class StepSizeSettings:            # src/stepsize/adapt.rs:308-329
    target_accept: float = 0.8
    initial_step: float = 0.1
    jitter: Optional[float] = 0.1
    method: int = STEP_DUAL_AVERAGE          # StepSizeAdaptMethod::{DualAverage, Adam, Fixed(f64)}
    fixed_step_size: float = 0.0
    dual_average: DualAverageOptions = field(default_factory=DualAverageOptions)
    adam: AdamOptions = field(default_factory=AdamOptions)


@dataclass
class DiagAdaptExpSettings:        # src/transform/adapt/diagonal.rs:92-106
    store_mass_matrix: bool = False
    use_grad_based_estimate: bool = True


class KineticEnergyKind:           # src/dynamics/transformed_hamiltonian.rs:27-50
    EUCLIDEAN, EXACT_NORMAL, MICROCANONICAL = 0, 1, 2


@dataclass
class LowRankSettings:             # src/transform/low_rank.rs:188-203
    store_mass_matrix: bool = False
    gamma: float = 1e-5
    eigval_cutoff: float = 2.0


@dataclass
class EuclideanAdaptOptions:       # src/adapt_strategy.rs:41-69
    step_size_settings: StepSizeSettings = field(default_factory=StepSizeSettings)
    mass_matrix_options: object = field(default_factory=DiagAdaptExpSettings)   # DiagAdaptExpSettings | LowRankSettings
    early_window: float = 0.3
    step_size_window: float = 0.15
    mass_matrix_switch_freq: int = 80
    early_mass_matrix_switch_freq: int = 10
    mass_matrix_update_freq: int = 1
    mass_matrix_window_growth: float = 1.5


@dataclass
class DiagNutsSettings:            # src/sampler.rs:199-239; Default: :630-634
    num_tune: int = 400
    num_draws: int = 1000
    maxdepth: int = 10
    mindepth: int = 0
    store_gradient: bool = False
    store_unconstrained: bool = False
    store_transformed: bool = False
    max_energy_error: float = 1000.0
    store_divergences: bool = False
    adapt_options: EuclideanAdaptOptions = field(default_factory=EuclideanAdaptOptions)
    check_turning: bool = True
    target_integration_time: Optional[float] = None
    num_chains: int = 6
    seed: int = 0
    extra_doublings: int = 0
    freeze_transform: bool = False   # engine knob (not a reference setting): see nm_settings.freeze_transform
    trajectory_kind: int = 0         # KineticEnergyKind (src/sampler.rs:224-232): KineticEnergyKind.EUCLIDEAN / EXACT_NORMAL / MICROCANONICAL

    def to_c(self) -> NmSettings:
        s = NmSettings()
        _lib.load().nm_settings_default(C.byref(s))
        a, st = self.adapt_options, self.adapt_options.step_size_settings
        s.num_tune, s.num_draws, s.maxdepth, s.mindepth = self.num_tune, self.num_draws, self.maxdepth, self.mindepth
        s.max_energy_error = self.max_energy_error
        s.check_turning = int(self.check_turning)
        s.extra_doublings, s.seed, s.num_chains = self.extra_doublings, self.seed, self.num_chains
        s.store_gradient, s.store_unconstrained = int(self.store_gradient), int(self.store_unconstrained)
        s.store_transformed, s.store_divergences = int(self.store_transformed), int(self.store_divergences)
        s.has_target_integration_time = int(self.target_integration_time is not None)
        s.target_integration_time = self.target_integration_time or 0.0
        s.early_window, s.step_size_window = a.early_window, a.step_size_window
        s.mass_matrix_switch_freq = a.mass_matrix_switch_freq
        s.early_mass_matrix_switch_freq = a.early_mass_matrix_switch_freq
        s.mass_matrix_update_freq = a.mass_matrix_update_freq
        s.mass_matrix_window_growth = a.mass_matrix_window_growth
        mo = a.mass_matrix_options
        s.store_mass_matrix = int(mo.store_mass_matrix)
        if isinstance(mo, LowRankSettings):          # LowRankNutsSettings = NutsSettings<EuclideanAdaptOptions<LowRankSettings>>
            s.adaptation = ADAPT_LOW_RANK
            s.lr_gamma, s.lr_eigval_cutoff = mo.gamma, mo.eigval_cutoff
        else:
            s.adaptation = ADAPT_DIAG
            s.use_grad_based_estimate = int(mo.use_grad_based_estimate)
        s.freeze_transform = int(self.freeze_transform)
        s.trajectory_kind = int(self.trajectory_kind)
        s.target_accept, s.initial_step = st.target_accept, st.initial_step
        s.has_jitter, s.jitter = int(st.jitter is not None), st.jitter or 0.0
        s.step_size_method, s.fixed_step_size = st.method, st.fixed_step_size
        s.da_k, s.da_t0, s.da_gamma = st.dual_average.k, st.dual_average.t0, st.dual_average.gamma
        s.da_max_step_size = st.dual_average.max_step_size
        s.adam_beta1, s.adam_beta2 = st.adam.beta1, st.adam.beta2
        s.adam_epsilon, s.adam_learning_rate = st.adam.epsilon, st.adam.learning_rate
        return s


class MclmcTrajectoryKind:         # src/mclmc.rs:44-70
    MICROCANONICAL, EUCLIDEAN, EUCLIDEAN_EARLY_THEN_MICROCANONICAL = 0, 1, 2


def _fixed_step_adapt_options():
    a = EuclideanAdaptOptions()
    a.step_size_settings.method, a.step_size_settings.fixed_step_size = STEP_FIXED, 0.5   # sampler.rs:371
    return a


@dataclass
class DiagMclmcSettings:           # MclmcSettings<EuclideanAdaptOptions<DiagAdaptExpSettings>> (src/sampler.rs:266-374; experimental upstream)
    step_size: float = 0.5
    momentum_decoherence_length: float = 3.0
    num_tune: int = 400
    num_draws: int = 1000
    num_chains: int = 6
    seed: int = 0
    max_energy_error: float = 1000.0
    store_unconstrained: bool = False
    store_gradient: bool = False
    store_transformed: bool = False
    store_divergences: bool = False
    adapt_options: EuclideanAdaptOptions = field(default_factory=_fixed_step_adapt_options)
    subsample_frequency: float = 1.0
    dynamic_step_size: bool = True
    trajectory_kind: int = MclmcTrajectoryKind.EUCLIDEAN_EARLY_THEN_MICROCANONICAL
    trajectory_switch_fraction: float = 0.3

    def to_c(self) -> NmSettings:
        n = DiagNutsSettings(num_tune=self.num_tune, num_draws=self.num_draws, num_chains=self.num_chains, seed=self.seed,
                             max_energy_error=self.max_energy_error, store_unconstrained=self.store_unconstrained,
                             store_gradient=self.store_gradient, store_transformed=self.store_transformed,
                             store_divergences=self.store_divergences, adapt_options=self.adapt_options)
        s = n.to_c()
        s.sampler = SAMPLER_MCLMC
        s.mclmc_step_size, s.momentum_decoherence_length = self.step_size, self.momentum_decoherence_length
        s.subsample_frequency, s.dynamic_step_size = self.subsample_frequency, int(self.dynamic_step_size)
        s.mclmc_trajectory_kind, s.trajectory_switch_fraction = int(self.trajectory_kind), self.trajectory_switch_fraction
        s.step_size_method, s.fixed_step_size = STEP_FIXED, self.step_size          # new_chain: Fixed(self.step_size) (sampler.rs:421-423)
        return s


def LowRankMclmcSettings(**kw):    # MclmcSettings<EuclideanAdaptOptions<LowRankSettings>> (src/sampler.rs:325-328; Default: :376-384)
    """`LowRankMclmcSettings`: the MCLMC settings with `LowRankSettings` as mass_matrix_options (num_tune 800,
    early_mass_matrix_switch_freq 20)."""
    a = _fixed_step_adapt_options()
    a.mass_matrix_options = LowRankSettings()
    a.early_mass_matrix_switch_freq = 20
    kw.setdefault("num_tune", 800)
    kw.setdefault("adapt_options", a)
    return DiagMclmcSettings(**kw)


def LowRankNutsSettings(**kw):     # src/sampler.rs:245; Default: :636-642 (num_tune 800, mass_matrix_update_freq 20)
    """`LowRankNutsSettings`: the same settings struct with `LowRankSettings` as mass_matrix_options."""
    lr = {k: kw.pop(k) for k in ("gamma", "eigval_cutoff", "store_mass_matrix") if k in kw}
    ao = kw.pop("adapt_options", None) or EuclideanAdaptOptions(mass_matrix_update_freq=20)
    ao.mass_matrix_options = LowRankSettings(**lr)
    kw.setdefault("num_tune", 800)
    return DiagNutsSettings(adapt_options=ao, **kw)


@dataclass
class Progress:                    # src/sampler.rs:165-174, one per chain
    draw: int
    chain: int
    diverging: bool
    tuning: bool
    step_size: float
    num_steps: int


class RecoverableLogpError(Exception):
    """`LogpError::is_recoverable() == true` (src/math/math.rs:9-13): the leapfrog becomes a divergence, the chain goes on."""


@dataclass
class LogpSpec:
    """A registered device density (the device-side stand-in for a `CpuLogpFunc`, src/math/cpu_math.rs:885-891)."""
    kind: int
    dim: int
    params: np.ndarray
    module_path: Optional[str] = None
    host_fn: object = None
    host_threads: int = 0

    @staticmethod
    def iid_normal(dim, mu=3.0):
        return LogpSpec(LOGP_IID_NORMAL, dim, np.array([mu], dtype=np.float64))

    @staticmethod
    def diag_normal(precision_diag):
        p = np.ascontiguousarray(precision_diag, dtype=np.float64)
        return LogpSpec(LOGP_DIAG_NORMAL, len(p), p)

    @staticmethod
    def funnel(dim=101):
        """Neal's funnel: x[0] = v ~ N(0, 9), x[1:] | v ~ N(0, e^v) (BASELINE config K3, defined by this repo)."""
        return LogpSpec(LOGP_FUNNEL, dim, np.zeros(0))

    @staticmethod
    def eight_schools(y=(28., 8., -3., 7., -1., 1., 18., 12.), sigma=(15., 10., 16., 11., 9., 11., 10., 18.)):
        """Non-centered 8 schools (mu, log tau, theta~[8]) (BASELINE config K4, defined by this repo)."""
        return LogpSpec(LOGP_EIGHT_SCHOOLS, 10, np.array(list(y) + list(sigma), dtype=np.float64))

    @staticmethod
    def mvn_precision(precision):
        """N(0, P^-1) with a full symmetric precision matrix: logp = -x'Px/2 (BASELINE config K5, defined by this repo)."""
        p = np.ascontiguousarray(precision, dtype=np.float64)
        if p.ndim != 2 or p.shape[0] != p.shape[1] or not (p == p.T).all():
            raise ValueError("precision must be a symmetric square matrix")
        return LogpSpec(LOGP_MVN_PREC, p.shape[0], p.reshape(-1))

    @staticmethod
    def module(dim, module_path, params=()):
        """A user density compiled into its own module (include/nuts_amd.h "User densities";
        nuts_rs_amd.build.build_density_module builds one from a header that defines the functor)."""
        return LogpSpec(LOGP_MODULE, int(dim), np.asarray(params, dtype=np.float64), module_path=str(module_path))

    @staticmethod
    def host_callback(dim, logp, threads=0):
        """A `CpuLogpFunc` on the host (the reference's own density interface, src/math/cpu_math.rs:885-891): `logp(chain,
        position: ndarray) -> (logp, gradient)`, or raise `RecoverableLogpError` / any other exception (unrecoverable).
        Evaluated once per leapfrog through the engine's mailbox path: the slow, fully general route."""
        def trampoline(ctx, chain, d, px, pg, plogp):
            try:
                x = np.ctypeslib.as_array(px, shape=(d,))
                lp, g = logp(int(chain), x.copy())
                np.ctypeslib.as_array(pg, shape=(d,))[:] = g
                plogp[0] = lp
                return 0
            except RecoverableLogpError:
                return 1
            except BaseException:     # noqa: BLE001 — an unrecoverable LogpError
                return 2
        spec = LogpSpec(LOGP_HOST_CALLBACK, int(dim), np.zeros(0))
        spec.host_fn = _lib.HOST_LOGP_FN(trampoline)
        spec.host_threads = threads
        return spec

    def to_c(self):
        self._keep = np.ascontiguousarray(self.params, dtype=np.float64)
        path = self.module_path.encode() if self.module_path else None
        c = NmLogpSpec(self.kind, self.dim, len(self._keep), self._keep.ctypes.data if len(self._keep) else None, path)
        if self.host_fn is not None:
            c.host_fn = C.cast(self.host_fn, C.c_void_p)
            c.host_threads = self.host_threads
        return c


class ChainBatch:
    """`n_chains` NUTS chains on one GPU: the batched `settings.new_chain(...)` of the reference."""

    def __init__(self, settings: DiagNutsSettings, logp: LogpSpec, n_chains: Optional[int] = None,
                 chain_id_offset: int = 0, device: int = -1, dims_per_lane: int = 0, waves_per_chain: int = 0,
                 grid_blocks: int = 0, lane_groups: int = 0, lowrank_max_rank: int = 0, chain_tiles: int = 0, lane_chains: int = 0):
        self.settings = settings
        self.logp = logp
        self.n_chains = int(n_chains if n_chains is not None else settings.num_chains)
        self.chain_id_offset = chain_id_offset
        L = _lib.load()
        cfg = NmEngineConfig()
        L.nm_engine_config_default(C.byref(cfg))
        cfg.device, cfg.chain_id_offset, cfg.dims_per_lane = device, chain_id_offset, dims_per_lane
        cfg.waves_per_chain, cfg.grid_blocks, cfg.lane_groups = waves_per_chain, grid_blocks, lane_groups
        cfg.lowrank_max_rank, cfg.chain_tiles, cfg.lane_chains = lowrank_max_rank, chain_tiles, lane_chains
        if logp.kind == LOGP_MODULE and os.environ.get("NUTS_AMD_SELFTEST", "1") != "0":
            # a USER module is compiled by the user's compiler into this engine's kernels, and the parity suite does not travel with it: before
            # the first such engine of a process, the built library must reproduce its known answers (nuts_rs_amd/selftest.py; ~0.1 s)
            from . import selftest
            selftest.run_once(device=device)
        self._cs = settings.to_c()
        self._cl = logp.to_c()
        h = C.c_void_p()
        check(L.nm_engine_create(C.byref(self._cs), C.byref(self._cl), self.n_chains, C.byref(cfg), C.byref(h)))
        self._h = h
        if logp.kind < LOGP_MODULE and os.environ.get("NUTS_AMD_SELFTEST", "1") != "0":
            # every kernel instantiation has its own known answer (selftest_instantiations.json, oracle-generated data): the runs of the ONE
            # instantiation this engine launches are checked the first time a process creates such an engine (0.1 - 0.3 s)
            from . import selftest
            try:
                selftest.first_use(self, device=device)
            except BaseException:
                self.close()
                raise

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().nm_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def dim(self):
        return self.logp.dim

    def threads_per_chain(self):
        """64 x waves cooperating on a chain; fixes the reduction order over dim (oracle: gpu_cfg(threads_per_chain))."""
        return int(_lib.load().nm_engine_threads_per_chain(self._h))

    def blocks_per_chain(self):
        """dim > 4096: co-resident blocks that share one chain, each owning a 4096-element slice (else 1)."""
        return int(_lib.load().nm_engine_blocks_per_chain(self._h))

    def dims_per_lane(self):
        return int(_lib.load().nm_engine_dims_per_lane(self._h))

    def group_launches(self) -> int:
        """Draw launches served by the 8-chains-per-wavefront kernel (small chains, after warm-up)."""
        return int(_lib.load().nm_engine_group_launches(self._h))

    def lane_launches(self) -> int:
        """Draw launches served by the one-chain-per-lane kernels (dim <= 16, nuts_lane.hpp)."""
        return int(_lib.load().nm_engine_lane_launches(self._h))

    def host_logp_calls(self) -> int:
        """Calls of the host density function so far (LogpSpec.host_callback)."""
        return int(_lib.load().nm_engine_host_logp_calls(self._h))

    def tile_launches(self) -> int:
        """Draw launches served by the 16-chains-per-block matrix-core kernel (shared transformation, nuts_tile.hpp)."""
        return int(_lib.load().nm_engine_tile_launches(self._h))

    def lockstep_launches(self):
        """... of which by the lockstep form (nuts_lockstep.hpp)."""
        return int(_lib.load().nm_engine_lockstep_launches(self._h))

    def reduce_order(self):
        """How the draws sum over dim (nm_engine_reduce_order): 0 wave kernels, 1 matrix-core tile kernel, 2 lockstep kernel —
        the oracle's gpu_cfg(threads_per_chain, lr_seq_dots=reduce_order())."""
        return int(_lib.load().nm_engine_reduce_order(self._h))

    def init_positions_uniform(self):
        """x0 ~ U(-1,1) from each chain's outer generator: `CpuMath::init_position` in Sampler order."""
        x0 = np.empty((self.n_chains, self.logp.dim))
        check(_lib.load().nm_init_positions_uniform(self.settings.seed, self.chain_id_offset, self.n_chains,
                                                    self.logp.dim, x0.ctypes.data))
        return x0

    def set_position(self, x0, raise_on_error=True, mask=None):
        """`Chain::set_position` for every chain; x0 is [n_chains, dim].  Returns per-chain status codes.
        mask (bool [n_chains]): only these chains are (re-)initialised, the others keep their whole state."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        if x0.shape != (self.n_chains, self.logp.dim):
            raise ValueError(f"x0 must have shape {(self.n_chains, self.logp.dim)}")
        status = np.zeros(self.n_chains, dtype=np.uint64)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        rc = _lib.load().nm_engine_set_positions_masked(self._h, x0.ctypes.data, m.ctypes.data if m is not None else None,
                                                        status.ctypes.data)
        if rc != 0 and raise_on_error:
            check(rc)
        return status

    def init_with_retries(self, x0=None, max_tries=500, raise_on_error=True):
        """The init loop of the reference's ChainProcess (src/sampler.rs:1126-1147): every chain whose initial point fails
        with BadInitGrad draws its next init_position and tries again (at most 500 times), without touching the chains
        that started.  Returns (status [n_chains], tries [n_chains])."""
        status, tries = np.zeros(self.n_chains, dtype=np.uint64), np.zeros(self.n_chains, dtype=np.uint64)
        x = None if x0 is None else np.ascontiguousarray(x0, dtype=np.float64)
        rc = _lib.load().nm_engine_init_positions_retry(self._h, x.ctypes.data if x is not None else None, max_tries,
                                                        status.ctypes.data, tries.ctypes.data)
        if rc != 0 and raise_on_error:
            check(rc)
        return status, tries

    def draw(self):
        """One `Chain::draw` per chain -> (positions [n_chains, dim], [Progress])."""
        pos, st = self.draw_many(1)
        prog = [Progress(int(r["draw"]), int(r["chain"]), bool(r["diverging"]), bool(r["tuning"]),
                         float(r["step_size"]), int(r["n_steps"])) for r in st[0]]
        return pos[0], prog

    def draw_many(self, n_draws, positions=True, stats=True, raise_on_error=True, out=None):
        """n_draws draws of every chain; returns host arrays ([n_draws, n_chains, dim], stats [n_draws, n_chains]).
        Chains that stopped with an error (or never started: BadInitGrad) raise NM_ERR_LOGP_FAILURE after the healthy
        chains' results are in the arrays; raise_on_error=False returns them anyway (rows of failed chains are unwritten).
        `out=(positions, stats)`: arrays of those shapes to fill (reused arrays are copied into at the PCIe rate; fresh ones
        are bound by their page faults)."""
        if out is not None:
            pos, st = out
            assert pos is None or (pos.shape == (n_draws, self.n_chains, self.logp.dim) and pos.dtype == np.float64 and pos.flags.c_contiguous)
            assert st is None or (st.shape == (n_draws, self.n_chains) and st.dtype == STATS_DTYPE and st.flags.c_contiguous)
            positions, stats = pos is not None, st is not None
        else:
            pos = np.empty((n_draws, self.n_chains, self.logp.dim)) if positions else None
            st = np.zeros((n_draws, self.n_chains), dtype=STATS_DTYPE) if stats else None
        rc = _lib.load().nm_engine_draw_to_host(self._h, n_draws, pos.ctypes.data if positions else None,
                                                st.ctypes.data if stats else None)
        if rc != _lib.NM_ERR_LOGP_FAILURE or raise_on_error:
            check(rc)
        return pos, st

    def stored_vectors(self):
        """The vector statistics the settings' store_* switches select (reference src/sampler.rs:218-226, :236)."""
        s = self.settings
        names = []
        if s.store_gradient:
            names.append("gradient")
        if s.store_transformed:
            names += ["transformed_position", "transformed_gradient"]
        if s.adapt_options.mass_matrix_options.store_mass_matrix:
            names += ["mass_matrix_inv", "transformation_mu"]
            if isinstance(s.adapt_options.mass_matrix_options, LowRankSettings):
                names += ["mass_matrix_eigvals"]
        if s.store_divergences:
            names += ["divergence_start", "divergence_start_gradient", "divergence_end"]
        return names

    def expanded_draw_many(self, n_draws, vectors=None):
        """`Chain::expanded_draw` x n_draws (src/chain.rs:190-204): positions, scalar stats and a dict of the
        vector-valued statistics ([n_draws, n_chains, dim]; rows of events that did not happen are NaN).
        `vectors` defaults to what the settings' store_* switches select."""
        names = list(self.stored_vectors() if vectors is None else vectors)
        bad = [k for k in names if k not in VECTOR_STATS]
        if bad:
            raise ValueError(f"unknown vector statistics {bad}; known: {VECTOR_STATS}")
        shape = (n_draws, self.n_chains, self.logp.dim)
        pos = np.empty(shape)
        st = np.zeros(shape[:2], dtype=STATS_DTYPE)
        vec = {k: np.empty(shape) for k in names}
        out = NmDrawOutputs()
        out.d_positions, out.d_stats = pos.ctypes.data, st.ctypes.data
        for k, a in vec.items():
            setattr(out, "d_" + k, a.ctypes.data)
        check(_lib.load().nm_engine_draw_ex_to_host(self._h, n_draws, C.byref(out)))
        return pos, st, vec

    def draw_device(self, n_draws, d_positions=0, d_stats=0, sync=True):
        """n_draws draws with results left in caller-provided device buffers (raw pointers, e.g. tensor.data_ptr())."""
        L = _lib.load()
        fn = L.nm_engine_draw if sync else L.nm_engine_draw_async
        check(fn(self._h, n_draws, C.c_void_p(d_positions or None), C.c_void_p(d_stats or None)))

    def draw_device_ex(self, n_draws, positions=0, stats=0, **vectors):
        """`expanded_draw` x n_draws with every result left in caller-provided DEVICE buffers (raw pointers): positions,
        stats and any of VECTOR_STATS by name (e.g. gradient=tensor.data_ptr())."""
        out = NmDrawOutputs()
        out.d_positions, out.d_stats = positions or None, stats or None
        for k, ptr in vectors.items():
            if k not in VECTOR_STATS:
                raise ValueError(f"unknown vector statistic {k}")
            setattr(out, "d_" + k, ptr or None)
        check(_lib.load().nm_engine_draw_ex(self._h, n_draws, C.byref(out)))

    def synchronize(self):
        check(_lib.load().nm_engine_synchronize(self._h))

    # ---- the low-rank transformation (LowRankNutsSettings; reference src/transform/low_rank.rs) ----
    def set_transform(self, stds, mean, vals, vecs, mu_low_rank):
        """`LowRankMassMatrix::update(stds, mean, vals, vecs, mean_low_rank)` for every chain.  vecs: [n_eig, dim] (one
        eigenvector per row); give every array a leading [n_chains] axis for per-chain transformations."""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        stds, mean, vals, vecs, mu = f(stds), f(mean), f(vals), f(vecs), f(mu_low_rank)
        per_chain = int(stds.ndim == 2)
        n_eig = int(vals.shape[-1]) if vals.size else 0
        want = (self.n_chains, self.logp.dim) if per_chain else (self.logp.dim,)
        if stds.shape != want or mean.shape != want or mu.shape != want:
            raise ValueError(f"stds, mean, mu_low_rank must have shape {want}")
        if vecs.size != (self.n_chains if per_chain else 1) * n_eig * self.logp.dim:
            raise ValueError("vecs must be [n_eig, dim] (or [n_chains, n_eig, dim])")
        self._keep_tr = (stds, mean, vals, vecs, mu)
        check(_lib.load().nm_engine_set_transform(self._h, per_chain, n_eig, stds.ctypes.data, mean.ctypes.data,
                                                  vals.ctypes.data if n_eig else None, vecs.ctypes.data if n_eig else None,
                                                  mu.ctypes.data))

    def set_lowrank_estimator(self, fn=None, n_threads=0):
        """Replace the built-in host estimator of `compute_update` (a _lib.LOWRANK_ESTIMATOR_FN; None restores it)."""
        self._keep_est = fn
        check(_lib.load().nm_engine_set_lowrank_estimator(self._h, C.cast(fn, C.c_void_p) if fn is not None else None, None, n_threads))

    def set_lowrank_estimator_place(self, place):
        """Where the built-in estimator runs (nm_engine_set_lowrank_estimator_place): 0 / "auto" the device where the block
        algorithm takes the window (dim <= 512, <= 1024 draws), else host threads; 1 / "host"; 2 / "device" (or an error)."""
        place = {"auto": 0, "host": 1, "device": 2}.get(place, place)
        check(_lib.load().nm_engine_set_lowrank_estimator_place(self._h, int(place)))

    def lowrank_device_updates(self):
        """Estimator calls that ran on the device so far."""
        return int(_lib.load().nm_engine_lowrank_device_updates(self._h))

    def lowrank(self):
        """Current low-rank part per chain: (n_eig [n_chains], lambda^(1/2) [n_chains, max_rank], vecs [n_chains, max_rank, dim],
        mu_low_rank [n_chains, dim])."""
        L = _lib.load()
        R = int(L.nm_engine_lowrank_max_rank(self._h))
        n = np.zeros(self.n_chains, dtype=np.uint64)
        vals, vecs = np.zeros((self.n_chains, R)), np.zeros((self.n_chains, R, self.logp.dim))
        mu = np.zeros((self.n_chains, self.logp.dim))
        check(L.nm_engine_get_lowrank(self._h, n.ctypes.data, vals.ctypes.data, vecs.ctypes.data, mu.ctypes.data))
        return n, vals, vecs, mu

    def positions(self):
        out = np.empty((self.n_chains, self.logp.dim))
        check(_lib.load().nm_engine_get_positions(self._h, out.ctypes.data))
        return out

    def gradients(self):
        out = np.empty((self.n_chains, self.logp.dim))
        check(_lib.load().nm_engine_get_gradients(self._h, out.ctypes.data))
        return out

    def mass_matrix(self):
        sd, mu = np.empty((self.n_chains, self.logp.dim)), np.empty((self.n_chains, self.logp.dim))
        check(_lib.load().nm_engine_get_mass_matrix(self._h, sd.ctypes.data, mu.ctypes.data))
        return sd, mu

    def step_sizes(self):
        out = np.empty(self.n_chains)
        check(_lib.load().nm_engine_get_step_sizes(self._h, out.ctypes.data))
        return out

    def counters(self):
        steps, draws, launches = C.c_uint64(), C.c_uint64(), C.c_uint64()
        ms = C.c_double()
        check(_lib.load().nm_engine_get_counters(self._h, C.byref(steps), C.byref(draws), C.byref(ms),
                                                 C.byref(launches)))
        return dict(total_leapfrogs=steps.value, total_draws=draws.value, kernel_ms=ms.value,
                    kernel_launches=launches.value)

    def reset_counters(self):
        check(_lib.load().nm_engine_reset_counters(self._h))

    def stream(self):
        return _lib.load().nm_engine_stream(self._h)


def sample(settings: DiagNutsSettings, logp: LogpSpec, x0=None, chain_id_offset=0, device=-1, chunk_bytes=0):
    """The reference's `Sampler` loop for every chain (src/sampler.rs:1120-1199): init, num_tune + num_draws draws.

    Returns (positions [num_tune+num_draws, n_chains, dim], stats)."""
    batch = ChainBatch(settings, logp, settings.num_chains, chain_id_offset, device)
    batch.init_with_retries(x0)                      # up to 500 initial points per chain, like the reference
    total = settings.num_tune + settings.num_draws
    # one trace, filled in place: draw_many cuts the launch into chunks itself (bounded device staging, copies under the next
    # chunk's kernel); `chunk_bytes` > 0 additionally bounds the draws per call (e.g. to poll for interrupts in between)
    pos = np.empty((total, batch.n_chains, logp.dim))
    st = np.zeros((total, batch.n_chains), dtype=STATS_DTYPE)
    per_draw = batch.n_chains * (logp.dim * 8 + STATS_DTYPE.itemsize)
    chunk = max(1, min(total, chunk_bytes // per_draw)) if chunk_bytes else total
    done = 0
    try:
        while done < total:
            n = min(chunk, total - done)
            batch.draw_many(n, out=(pos[done:done + n], st[done:done + n]))
            done += n
    finally:
        batch.close()
    return pos, st
