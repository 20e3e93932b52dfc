"""ctypes loader for libnuts_amd.so — the C ABI of include/nuts_amd.h.

There is no CPU fallback: if the shared library is missing this module raises, and the library itself refuses
to create an engine without a HIP device (NM_ERR_NO_DEVICE).
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NUTS_AMD_LIB", os.path.join(_HERE, "libnuts_amd.so"))   # env override: tuning builds

NM_OK = 0
NM_ERR_LOGP_FAILURE = 6
STATUS_NAMES = {0: "NM_OK", 1: "NM_ERR_INVALID_ARG", 2: "NM_ERR_NO_DEVICE", 3: "NM_ERR_HIP", 4: "NM_ERR_UNSUPPORTED",
                5: "NM_ERR_BAD_INIT", 6: "NM_ERR_LOGP_FAILURE", 7: "NM_ERR_STATE"}

# every symbol include/nuts_amd.h declares
ABI_SYMBOLS = [
    "nm_settings_default", "nm_engine_config_default", "nm_engine_create", "nm_engine_destroy",
    "nm_engine_set_positions", "nm_init_positions_uniform", "nm_engine_draw", "nm_engine_draw_async",
    "nm_engine_synchronize", "nm_engine_draw_to_host", "nm_host_register", "nm_host_unregister", "nm_engine_draw_ex", "nm_engine_draw_ex_async",
    "nm_engine_draw_ex_to_host", "nm_engine_get_positions", "nm_engine_get_gradients",
    "nm_engine_get_mass_matrix", "nm_engine_get_step_sizes", "nm_engine_get_counters", "nm_engine_reset_counters",
    "nm_engine_dim", "nm_engine_num_chains", "nm_engine_threads_per_chain", "nm_engine_dims_per_lane", "nm_engine_blocks_per_chain", "nm_engine_group_launches", "nm_engine_lane_launches", "nm_engine_stream", "nm_leapfrog_batch", "nm_turning_batch",
    "nm_scalar_math_batch", "nm_standard_normal_batch", "nm_chain_rng_key", "nm_last_error", "nm_abi_version",
    "nm_pick_tiling", "nm_probe_bandwidth", "nm_probe_issue", "nm_settings_default_low_rank", "nm_settings_default_mclmc", "nm_engine_set_lowrank_estimator",
    "nm_lowrank_compute_update", "nm_lowrank_test_spd_mean", "nm_lowrank_test_estimate_mass_matrix", "nm_engine_set_lowrank_estimator_place", "nm_engine_lowrank_device_updates", "nm_lowrank_block_twin", "nm_lowrank_test_block_device", "nm_engine_set_transform", "nm_engine_get_lowrank", "nm_engine_lowrank_max_rank",
    "nm_lowrank_transform_batch", "nm_engine_set_positions_masked", "nm_engine_init_positions_retry",
    "nm_init_positions_uniform_at", "nm_engine_tile_launches", "nm_engine_lockstep_launches", "nm_engine_reduce_order", "nm_engine_host_logp_calls", "nm_pooled_partials", "nm_pooled_exchange", "nm_pooled_finish", "nm_pooled_last_error",
    # the per-vector `Math` seam
    "nm_math_create", "nm_math_destroy", "nm_math_dim", "nm_math_threads", "nm_math_last_error", "nm_vec_new", "nm_vec_free", "nm_vec_read_from_slice", "nm_vec_write_to_slice", "nm_vec_copy_into", "nm_vec_fill_array", "nm_vec_array_recip", "nm_vec_axpy_out", "nm_vec_axpy", "nm_vec_array_mult", "nm_vec_array_vector_dot", "nm_vec_scalar_prods3", "nm_vec_array_gaussian", "nm_vec_array_update_variance", "nm_vec_array_update_var_inv_std_draw_grad", "nm_vec_array_update_var_inv_std_grad", "nm_vec_array_update_var_inv_std_draw", "nm_vec_array_sum_ln", "nm_vec_array_all_finite", "nm_vec_logp_array", "nm_vec_sq_norm_sum", "nm_vec_std_norm_flow", "nm_vec_std_norm_grad_flow", "nm_vec_esh_momentum_update", "nm_vec_array_normalize",
]


class NmSettings(C.Structure):
    """nm_settings == the reference's DiagNutsSettings, field for field (src/sampler.rs:199-239)."""
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


class NmLogpSpec(C.Structure):
    _fields_ = [("kind", C.c_uint64), ("dim", C.c_uint64), ("n_params", C.c_uint64), ("h_params", C.c_void_p),
                ("module_path", C.c_char_p), ("host_fn", C.c_void_p), ("host_ctx", C.c_void_p), ("host_threads", C.c_uint64)]


# nm_host_logp_fn: (ctx, chain, dim, position*, gradient*, logp*) -> 0 ok | 1 recoverable error | 2 fatal
HOST_LOGP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                           C.POINTER(C.c_double))


class NmEngineConfig(C.Structure):
    _fields_ = [("device", C.c_int64), ("chain_id_offset", C.c_uint64), ("dims_per_lane", C.c_uint64),
                ("waves_per_chain", C.c_uint64), ("grid_blocks", C.c_uint64), ("lane_groups", C.c_uint64),
                ("chain_tiles", C.c_uint64), ("lowrank_max_rank", C.c_uint64), ("lane_chains", C.c_uint64)]


STATS_DTYPE = np.dtype([
    ("draw", "<u8"), ("chain", "<u8"), ("depth", "<u8"), ("maxdepth_reached", "<u8"), ("diverging", "<u8"),
    ("tuning", "<u8"), ("n_steps", "<u8"), ("index_in_trajectory", "<i8"), ("transformation_index", "<i8"),
    ("step_size", "<f8"), ("step_size_bar", "<f8"), ("mean_tree_accept", "<f8"), ("mean_tree_accept_sym", "<f8"),
    ("max_energy_error", "<f8"), ("logp", "<f8"), ("energy", "<f8"), ("energy_error", "<f8"),
    ("fisher_distance", "<f8"), ("divergence_energy_error", "<f8"), ("chain_status", "<u8"),
    ("transformation_update_id", "<i8"), ("num_eigenvalues", "<u8"), ("energy_change", "<f8"), ("average_step_size", "<f8"),
])

# nm_draw_outputs: the draws, the scalar statistics and the vector-valued statistics (reference stat names)
VECTOR_STATS = ("gradient", "transformed_position", "transformed_gradient", "mass_matrix_inv", "transformation_mu",
                "divergence_start", "divergence_start_gradient", "divergence_end", "mass_matrix_eigvals")


class NmDrawOutputs(C.Structure):
    _fields_ = ([("d_positions", C.c_void_p), ("d_stats", C.c_void_p)] + [("d_" + k, C.c_void_p) for k in VECTOR_STATS]
                + [("reserved", C.c_uint64 * 5)])


# nm_lowrank_estimator_fn
LOWRANK_ESTIMATOR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))

_lib = None


class NutsAmdError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7) and asks for it by the
    name `libamdhip64.so`; libnuts_amd.so asks for `libamdhip64.so.7`.  When torch comes first the loader hands this library
    torch's copy (SONAME match); when this library comes first it brings /opt/rocm's copy and torch then loads a SECOND runtime —
    the one that initialises later sees no device (found in round 4: build() followed by smoke() in one process).  So torch's
    copy, where torch is installed, is loaded here by path before the engine — without importing torch."""
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


def load():
    """Load the shared library (no compute, no GPU needed)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension with `python -m nuts_rs_amd.build` "
            "(nuts_rs_amd has no CPU fallback)")
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u64, dbl = C.c_void_p, C.c_uint64, C.c_double
    L.nm_settings_default.argtypes = [C.POINTER(NmSettings)]
    L.nm_settings_default.restype = None
    L.nm_engine_config_default.argtypes = [C.POINTER(NmEngineConfig)]
    L.nm_engine_config_default.restype = None
    L.nm_engine_create.argtypes = [C.POINTER(NmSettings), C.POINTER(NmLogpSpec), u64, C.POINTER(NmEngineConfig),
                                   C.POINTER(vp)]
    L.nm_engine_destroy.argtypes = [vp]
    L.nm_engine_destroy.restype = None
    L.nm_engine_set_positions.argtypes = [vp, vp, vp]
    L.nm_init_positions_uniform.argtypes = [u64, u64, u64, u64, vp]
    L.nm_init_positions_uniform_at.argtypes = [u64, u64, u64, u64, u64, vp]
    L.nm_engine_set_positions_masked.argtypes = [vp, vp, vp, vp]
    L.nm_engine_init_positions_retry.argtypes = [vp, vp, u64, vp, vp]
    L.nm_engine_draw.argtypes = [vp, u64, vp, vp]
    L.nm_engine_draw_async.argtypes = [vp, u64, vp, vp]
    L.nm_engine_synchronize.argtypes = [vp]
    L.nm_engine_draw_to_host.argtypes = [vp, u64, vp, vp]
    L.nm_host_register.argtypes = [vp, u64]
    L.nm_host_unregister.argtypes = [vp]
    L.nm_engine_draw_ex.argtypes = [vp, u64, C.POINTER(NmDrawOutputs)]
    L.nm_engine_draw_ex_async.argtypes = [vp, u64, C.POINTER(NmDrawOutputs)]
    L.nm_engine_draw_ex_to_host.argtypes = [vp, u64, C.POINTER(NmDrawOutputs)]
    L.nm_engine_get_positions.argtypes = [vp, vp]
    L.nm_engine_get_gradients.argtypes = [vp, vp]
    L.nm_engine_get_mass_matrix.argtypes = [vp, vp, vp]
    L.nm_engine_get_step_sizes.argtypes = [vp, vp]
    L.nm_engine_get_counters.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(dbl), C.POINTER(u64)]
    L.nm_engine_reset_counters.argtypes = [vp]
    L.nm_engine_dim.argtypes = [vp]
    L.nm_engine_dim.restype = u64
    L.nm_engine_num_chains.argtypes = [vp]
    L.nm_engine_num_chains.restype = u64
    L.nm_engine_threads_per_chain.argtypes = [vp]
    L.nm_engine_threads_per_chain.restype = u64
    L.nm_engine_blocks_per_chain.argtypes = [vp]
    L.nm_engine_blocks_per_chain.restype = u64
    L.nm_engine_dims_per_lane.argtypes = [vp]
    L.nm_engine_dims_per_lane.restype = u64
    L.nm_engine_group_launches.argtypes = [vp]
    L.nm_engine_group_launches.restype = u64
    L.nm_engine_lane_launches.argtypes = [vp]
    L.nm_engine_lane_launches.restype = u64
    L.nm_engine_stream.argtypes = [vp]
    L.nm_engine_stream.restype = vp
    L.nm_leapfrog_batch.argtypes = [C.POINTER(NmLogpSpec), u64, u64] + [vp] * 16 + [vp]
    L.nm_turning_batch.argtypes = [u64, u64, u64, vp, vp, vp, vp, vp, vp]
    L.nm_scalar_math_batch.argtypes = [u64, u64, vp, vp, vp, vp]
    L.nm_standard_normal_batch.argtypes = [u64, u64, vp, vp, vp, vp]
    L.nm_chain_rng_key.argtypes = [u64, u64, vp]
    L.nm_pick_tiling.argtypes = [u64, u64, u64, C.POINTER(u64), C.POINTER(u64)]
    L.nm_settings_default_low_rank.argtypes = [C.POINTER(NmSettings)]
    L.nm_settings_default_low_rank.restype = None
    L.nm_settings_default_mclmc.argtypes = [C.POINTER(NmSettings)]
    L.nm_settings_default_mclmc.restype = None
    L.nm_engine_set_lowrank_estimator.argtypes = [vp, vp, vp, u64]
    L.nm_lowrank_compute_update.argtypes = [vp, u64, u64, vp, vp, dbl, dbl, vp, vp, C.POINTER(u64), vp, vp, vp]
    L.nm_lowrank_test_spd_mean.argtypes = [u64, vp, vp, vp, u64]
    L.nm_lowrank_test_estimate_mass_matrix.argtypes = [u64, u64, vp, vp, dbl, vp, vp, u64]
    L.nm_engine_set_lowrank_estimator_place.argtypes = [vp, u64]
    L.nm_engine_lowrank_device_updates.argtypes = [vp]
    L.nm_engine_lowrank_device_updates.restype = u64
    L.nm_lowrank_block_twin.argtypes = [vp, u64, u64, vp, vp, dbl, dbl, vp, vp, C.POINTER(u64), vp, vp, vp]
    L.nm_lowrank_test_block_device.argtypes = [u64, u64, u64, vp, vp, dbl, dbl, vp, vp, vp, vp, vp, vp, vp, vp]
    L.nm_engine_set_transform.argtypes = [vp, u64, u64, vp, vp, vp, vp, vp]
    L.nm_engine_get_lowrank.argtypes = [vp, vp, vp, vp, vp]
    L.nm_engine_lowrank_max_rank.argtypes = [vp]
    L.nm_engine_lowrank_max_rank.restype = u64
    L.nm_engine_tile_launches.argtypes = [vp]
    L.nm_engine_tile_launches.restype = u64
    for fn in (L.nm_engine_lockstep_launches, L.nm_engine_reduce_order):
        fn.argtypes = [vp]
        fn.restype = u64
    L.nm_engine_host_logp_calls.argtypes = [vp]
    L.nm_engine_host_logp_calls.restype = u64
    pd, pu = C.POINTER(dbl), C.POINTER(u64)
    L.nm_math_create.argtypes = [C.POINTER(NmLogpSpec), C.POINTER(vp)]
    L.nm_math_destroy.argtypes = [vp]
    L.nm_math_destroy.restype = None
    L.nm_math_dim.argtypes = [vp]
    L.nm_math_dim.restype = u64
    L.nm_math_threads.argtypes = [vp]
    L.nm_math_threads.restype = u64
    L.nm_math_last_error.restype = C.c_char_p
    L.nm_vec_new.argtypes = [vp, C.POINTER(vp)]
    L.nm_vec_free.argtypes = [vp]
    L.nm_vec_free.restype = None
    L.nm_vec_read_from_slice.argtypes = [vp, vp, vp]
    L.nm_vec_write_to_slice.argtypes = [vp, vp, vp]
    L.nm_vec_copy_into.argtypes = [vp, vp, vp]
    L.nm_vec_fill_array.argtypes = [vp, vp, dbl]
    L.nm_vec_array_recip.argtypes = [vp, vp, vp]
    L.nm_vec_axpy_out.argtypes = [vp, vp, vp, dbl, vp]
    L.nm_vec_axpy.argtypes = [vp, vp, vp, dbl]
    L.nm_vec_array_mult.argtypes = [vp, vp, vp, vp]
    L.nm_vec_array_vector_dot.argtypes = [vp, vp, vp, pd]
    L.nm_vec_scalar_prods3.argtypes = [vp, vp, vp, vp, vp, vp, pd]
    L.nm_vec_array_gaussian.argtypes = [vp, vp, pu, vp, vp]
    L.nm_vec_array_update_variance.argtypes = [vp, vp, vp, vp, dbl]
    L.nm_vec_array_update_var_inv_std_draw_grad.argtypes = [vp, vp, vp, vp, vp, u64, dbl, dbl, dbl]
    L.nm_vec_array_update_var_inv_std_grad.argtypes = [vp, vp, vp, vp, dbl, dbl, dbl]
    L.nm_vec_array_update_var_inv_std_draw.argtypes = [vp, vp, vp, vp, dbl, u64, dbl, dbl, dbl]
    L.nm_vec_array_sum_ln.argtypes = [vp, vp, pd]
    L.nm_vec_array_all_finite.argtypes = [vp, vp, u64, pu]
    L.nm_vec_logp_array.argtypes = [vp, vp, vp, pd, pu]
    L.nm_vec_sq_norm_sum.argtypes = [vp, vp, vp, pd]
    L.nm_vec_std_norm_flow.argtypes = [vp, vp, vp, vp, dbl]
    L.nm_vec_std_norm_grad_flow.argtypes = [vp, vp, vp, vp, vp, dbl]
    L.nm_vec_esh_momentum_update.argtypes = [vp, vp, vp, dbl, pd]
    L.nm_vec_array_normalize.argtypes = [vp, vp]
    L.nm_lowrank_transform_batch.argtypes = [u64, u64, u64, u64, u64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.nm_probe_bandwidth.argtypes = [u64, u64, u64, C.POINTER(dbl), C.POINTER(u64), C.POINTER(u64)]
    L.nm_probe_issue.argtypes = [u64, u64, C.POINTER(dbl)]
    L.nm_pooled_partials.argtypes = [u64, u64, vp, vp, vp, vp, vp]
    L.nm_pooled_exchange.argtypes = [vp, u64, u64, vp, vp, vp]
    L.nm_pooled_finish.argtypes = [u64, u64, vp, vp, vp, vp, vp]
    L.nm_pooled_last_error.restype = C.c_char_p
    L.nm_last_error.restype = C.c_char_p
    L.nm_abi_version.restype = u64
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("nm_settings_default",):
            fn.restype = C.c_int
    _lib = L
    return L


def check(status):
    if status != NM_OK:
        raise NutsAmdError(status, load().nm_last_error().decode())


def check_status(status, message_fn):
    """like check(), for entry points that keep their own error text (nm_math_*, nm_pooled_*)"""
    if status != NM_OK:
        raise NutsAmdError(status, message_fn().decode())
