/*
 * nuts_amd.h — C ABI of the MI355X-native many-chain NUTS engine (libnuts_amd.so).
 *
 * This is the drop-in boundary for ONE hot path of pymc-devs/nuts-rs: the per-chain driver seam
 * `Settings::new_chain -> Chain::{set_position, draw}` (reference src/sampler.rs:53-63,
 * src/chain.rs:24-42, :137-188), batched over many independent chains.  Everything the reference
 * does inside `NutsChain::draw` — momentum refresh, the recursive-doubling tree with the fused
 * leapfrog + logp/grad (src/nuts.rs:108-388, src/dynamics/transformed_hamiltonian.rs:524-736),
 * the diagonal mass-matrix and dual-averaging adaptation (src/adapt_strategy.rs:121-222,
 * src/transform/adapt/diagonal.rs, src/stepsize/) — runs inside hand-written HIP kernels for
 * gfx950; the host only launches and copies results.
 *
 * Conventions
 *   - plain C, no torch types; every struct field is 8 bytes wide (no padding surprises for FFI).
 *   - per-chain vectors cross the ABI in the reference's own layout: `[chain][dim]` row-major
 *     (one contiguous `&[f64]` of length dim per chain, reference src/chain.rs:31,34).
 *   - pointers named `h_*` are host pointers, `d_*` are device (HBM) pointers.
 *   - every function returns an nm_status; nm_last_error() gives the message for the calling
 *     thread.  A handle is not thread-safe: one driver thread per engine (same contract as the
 *     reference's `!Sync` chain, src/chain.rs:44-61).
 *   - the library has NO CPU fallback: creating an engine without a usable HIP device fails with
 *     NM_ERR_NO_DEVICE.
 */
#ifndef NUTS_AMD_H
#define NUTS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NM_ABI_VERSION 16

typedef enum nm_status {
    NM_OK = 0,
    NM_ERR_INVALID_ARG = 1,
    NM_ERR_NO_DEVICE = 2,      /* no HIP device / kernel image not loadable: never falls back to CPU */
    NM_ERR_HIP = 3,            /* a HIP runtime call failed */
    NM_ERR_UNSUPPORTED = 4,    /* valid reference setting that this engine does not implement */
    NM_ERR_BAD_INIT = 5,       /* NutsError::BadInitGrad for >=1 chain (reference src/nuts.rs:21) */
    NM_ERR_LOGP_FAILURE = 6,   /* NutsError::LogpFailure for >=1 chain (reference src/nuts.rs:15) */
    NM_ERR_STATE = 7           /* call order violated (e.g. draw before set_positions) */
} nm_status;

/* Per-chain status codes written by the kernels (reference error taxonomy). */
#define NM_CHAIN_OK 0
#define NM_CHAIN_BAD_INIT 1     /* BadInitGrad: non-finite x/g or zero whitened gradient (transformed_hamiltonian.rs:310-324) */
#define NM_CHAIN_LOGP_FATAL 2   /* unrecoverable logp error / invalid Bernoulli probability */

/* ---------------------------------------------------------------------------------------------
 * Settings: mirrors `DiagNutsSettings = NutsSettings<EuclideanAdaptOptions<DiagAdaptExpSettings>>`
 * (reference src/sampler.rs:199-239, defaults :507-531 and :630-634;
 *  src/adapt_strategy.rs:41-69; src/stepsize/adapt.rs:308-329; src/stepsize/dual_avg.rs:12-31;
 *  src/transform/adapt/diagonal.rs:92-106).  Field names are the reference's.
 * Booleans are uint64_t (0/1); Option<f64> is a (has_x, x) pair.
 * ------------------------------------------------------------------------------------------- */
typedef struct nm_settings {
    uint64_t num_tune;                       /* 400 */
    uint64_t num_draws;                      /* 1000 */
    uint64_t maxdepth;                       /* 10 */
    uint64_t mindepth;                       /* 0 */
    double   max_energy_error;               /* 1000.0 */
    uint64_t check_turning;                  /* 1 */
    uint64_t extra_doublings;                /* 0 */
    uint64_t seed;                           /* 0 */
    uint64_t num_chains;                     /* 6 */
    uint64_t store_gradient;                 /* 0 */
    uint64_t store_unconstrained;            /* 0 */
    uint64_t store_transformed;              /* 0 */
    uint64_t store_divergences;              /* 0 */
    uint64_t has_target_integration_time;    /* 0 (None) */
    double   target_integration_time;
    /* adapt_options: EuclideanAdaptOptions */
    double   early_window;                   /* 0.3 */
    double   step_size_window;               /* 0.15 */
    uint64_t mass_matrix_switch_freq;        /* 80 */
    uint64_t early_mass_matrix_switch_freq;  /* 10 */
    uint64_t mass_matrix_update_freq;        /* 1 */
    double   mass_matrix_window_growth;      /* 1.5 */
    /* adapt_options.mass_matrix_options: DiagAdaptExpSettings */
    uint64_t store_mass_matrix;              /* 0 */
    uint64_t use_grad_based_estimate;        /* 1 */
    /* adapt_options.step_size_settings: StepSizeSettings */
    double   target_accept;                  /* 0.8 */
    double   initial_step;                   /* 0.1 */
    uint64_t has_jitter;                     /* 1 (Some) */
    double   jitter;                         /* 0.1 */
    uint64_t step_size_method;               /* NM_STEP_DUAL_AVERAGE */
    double   fixed_step_size;                /* StepSizeAdaptMethod::Fixed(val) */
    /* adapt_options.step_size_settings.adapt_options.dual_average: DualAverageOptions */
    double   da_k;                           /* 0.75 */
    double   da_t0;                          /* 10 */
    double   da_gamma;                       /* 0.05 */
    double   da_max_step_size;               /* pi */
    /* adapt_options.step_size_settings.adapt_options.adam: AdamOptions (src/stepsize/adam.rs:12-34) */
    double   adam_beta1;                     /* 0.9 */
    double   adam_beta2;                     /* 0.999 */
    double   adam_epsilon;                   /* 1e-8 */
    double   adam_learning_rate;             /* 0.05 */
    /* which MassMatrixAdaptStrategy drives the transformation: `DiagNutsSettings` or `LowRankNutsSettings`
     * (reference src/sampler.rs:243-245, :651: GlobalStrategy<M, DiagAdaptStrategy<M>> / <M, LowRankMassMatrixStrategy>) */
    uint64_t adaptation;                     /* NM_ADAPT_DIAG */
    /* adapt_options.mass_matrix_options: LowRankSettings (src/transform/low_rank.rs:188-203); store_mass_matrix above is shared */
    double   lr_gamma;                       /* 1e-5 */
    double   lr_eigval_cutoff;               /* 2.0 */
    /* engine knob, not a reference setting: the transformation is the one given by nm_engine_set_transform and the
     * mass-matrix estimator steps of GlobalStrategy::adapt (update_estimators / switch / adapt) are skipped; the
     * step-size schedule runs as usual */
    uint64_t freeze_transform;               /* 0 */
    /* trajectory_kind: KineticEnergyKind (src/sampler.rs:224-232, src/dynamics/transformed_hamiltonian.rs:27-50): which
     * integrator the tree's leapfrog is.  NM_TRAJ_EXACT_NORMAL: the geodesic leapfrog that is exact for a standard-normal
     * potential (std_norm_flow / std_norm_grad_flow, src/math/util.rs:507-741).  NM_TRAJ_MICROCANONICAL: the isokinetic ESH
     * leapfrog (esh_momentum_update, src/math/cpu_math.rs:505-551; unit-sphere momentum, the point's kinetic energy is the
     * accumulated change, a divergence is |energy error| >= max_energy_error; needs dim >= 2).  The non-Euclidean kinds run
     * with either adaptation (DiagNutsSettings / LowRankNutsSettings) on the one-chain-per-block kernels (built-in densities and
     * NM_LOGP_HOST_CALLBACK). */
    uint64_t trajectory_kind;                /* NM_TRAJ_EUCLIDEAN */
    /* `MclmcSettings` (src/sampler.rs:266-317; experimental upstream): sampler = NM_SAMPLER_MCLMC replaces the NUTS tree by
     * `MclmcChain` (src/mclmc.rs:212-409): per draw round(subsample_frequency L / eps) leapfrogs of the current kinetic kind,
     * a partial momentum refresh (isokinetic Langevin / Ornstein-Uhlenbeck, transformed_hamiltonian.rs:770-825) on both sides
     * of each, and with dynamic_step_size the halve-and-retry ladder on a divergence (at most 10 halvings).  The step size is
     * the constant mclmc_step_size (the engine sets step_size_method = NM_STEP_FIXED, fixed_step_size = mclmc_step_size, as
     * DiagMclmcSettings::new_chain does, sampler.rs:421-423); geometry adapts as usual: NM_ADAPT_DIAG (DiagMclmcSettings) or
     * NM_ADAPT_LOW_RANK (LowRankMclmcSettings).  Built-in densities and NM_LOGP_HOST_CALLBACK, dim >= 2.  nm_draw_stats: depth = leapfrogs taken, energy_change, average_step_size. */
    uint64_t sampler;                        /* NM_SAMPLER_NUTS */
    double   mclmc_step_size;                /* 0.5 */
    double   momentum_decoherence_length;    /* 3.0 */
    double   subsample_frequency;            /* 1.0 */
    uint64_t dynamic_step_size;              /* 1 */
    uint64_t mclmc_trajectory_kind;          /* NM_MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL */
    double   trajectory_switch_fraction;     /* 0.3 */
} nm_settings;

#define NM_SAMPLER_NUTS 0
#define NM_SAMPLER_MCLMC 1
/* MclmcTrajectoryKind (src/mclmc.rs:44-70) */
#define NM_MCLMC_MICROCANONICAL 0
#define NM_MCLMC_EUCLIDEAN 1
#define NM_MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL 2

#define NM_TRAJ_EUCLIDEAN 0
#define NM_TRAJ_EXACT_NORMAL 1
#define NM_TRAJ_MICROCANONICAL 2

#define NM_ADAPT_DIAG 0
#define NM_ADAPT_LOW_RANK 1   /* reference src/transform/low_rank.rs, src/transform/adapt/low_rank.rs */

#define NM_STEP_DUAL_AVERAGE 0
#define NM_STEP_ADAM 1      /* reference src/stepsize/adam.rs:42-112 */
#define NM_STEP_FIXED 2

/* Fill `s` with `DiagNutsSettings::default()` (reference src/sampler.rs:630-634). */
void nm_settings_default(nm_settings* s);
/* Fill `s` with `LowRankNutsSettings::default()` (reference src/sampler.rs:636-642: num_tune 800, mass_matrix_update_freq 20). */
void nm_settings_default_low_rank(nm_settings* s);
/* Fill `s` with `DiagMclmcSettings::default()` (reference src/sampler.rs:342-374: step size 0.5, L 3, num_tune 400, 6 chains). */
void nm_settings_default_mclmc(nm_settings* s);

/* ---------------------------------------------------------------------------------------------
 * Log-density registry.  The reference takes an arbitrary `CpuLogpFunc::logp(&[f64], &mut [f64])
 * -> Result<f64, E>` (src/math/cpu_math.rs:885-891).  A host closure cannot be fused into a
 * device kernel, so densities are device functors selected by `kind` with a parameter blob.
 * ------------------------------------------------------------------------------------------- */
#define NM_LOGP_IID_NORMAL 0     /* params[0]=mu.  logp = sum -0.5*(x-mu)^2, g = -(x-mu)
                                    (reference benches/sample.rs:49-62, src/math/test_logps.rs:49-58) */
#define NM_LOGP_DIAG_NORMAL 1    /* params[0..dim)=precision diag p_i.  logp = -0.5 sum p_i x_i^2 + norm, g=-p_i x_i
                                    (the diagonal-P case of the MvNormal fixture, src/transform/mod.rs:98-112) */
#define NM_LOGP_FUNNEL 2         /* Neal's funnel, dim = 1 + n (defined by this repo) */
#define NM_LOGP_EIGHT_SCHOOLS 3  /* non-centered 8 schools, dim = 10 (defined by this repo) */
#define NM_LOGP_MVN_PREC 4       /* params[dim*dim] = symmetric precision P, row-major.  logp = -0.5 x'Px (no constant),
                                    g = -Px, (Px)_d = sum_j fma(P[j][d], x_j, .) with j ascending (the
                                    full-P form of the MvNormal fixture, src/transform/mod.rs:98-112; defined by this repo) */

#define NM_LOGP_MODULE 5         /* a user density compiled into its own shared object (see below) */
#define NM_LOGP_HOST_CALLBACK 6  /* the slow path: a HOST function with the reference's own shape, evaluated once per leapfrog
                                    through a mailbox in pinned host memory (see nm_host_logp_fn) */

/* `CpuLogpFunc::logp(&mut self, position: &[f64], gradient: &mut [f64]) -> Result<f64, Self::LogpError>`
 * (reference src/math/cpu_math.rs:885-891) as a C function pointer: fills gradient[dim] and *logp and returns
 *   0  Ok(logp)
 *   1  Err(e) with e.is_recoverable() == true   -> the leapfrog is a divergence with `energy_error: None`
 *                                                  (src/math/math.rs:9-13, src/dynamics/transformed_hamiltonian.rs:562-578)
 *   2  Err(e) with e.is_recoverable() == false  -> the chain stops: NutsError::LogpFailure (src/nuts.rs:15, :231),
 *                                                  NM_CHAIN_LOGP_FATAL in its status
 * `chain` is the global chain id (the reference builds one density per chain, src/sampler.rs:1121-1124: a callback may
 * keep per-chain state in ctx).  Called from the engine's service threads, concurrently for DIFFERENT chains, never
 * concurrently for one chain.  Besides the leapfrogs the engine evaluates the chosen point of every draw once more
 * (it keeps positions, not gradients, of the tree's candidates). */
typedef int (*nm_host_logp_fn)(void* ctx, uint64_t chain, uint64_t dim, const double* position, double* gradient, double* logp);

typedef struct nm_logp_spec {
    uint64_t      kind;
    uint64_t      dim;           /* 0 is served for NM_LOGP_IID_NORMAL: every draw returns the initial point (src/nuts.rs:322-326) */
    uint64_t      n_params;
    const double* h_params;      /* host pointer, n_params doubles, copied at engine creation */
    const char*   module_path;   /* NM_LOGP_MODULE: path of the density module (.so); NULL otherwise */
    nm_host_logp_fn host_fn;     /* NM_LOGP_HOST_CALLBACK: the function and its context; NULL otherwise */
    void*         host_ctx;
    uint64_t      host_threads;  /* NM_LOGP_HOST_CALLBACK: service threads calling host_fn (0 = min(cores, 16)) */
} nm_logp_spec;

/* ---------------------------------------------------------------------------------------------
 * User densities (the device-side answer to `CpuLogpFunc`, reference src/math/cpu_math.rs:885-970).
 * A density is a small HIP functor with the interface of the built-in ones (nuts_rs_amd/csrc/nuts_kernels.hpp,
 * e.g. `struct DiagNormal`): `init(params, dim, reducer)`, `set_lds(ptr)`, and
 *     template <int DPL, int W> double eval(const Tile<DPL>& x, Tile<DPL>& grad, int dim, Reducer<W>& R) const
 * returning logp and filling the gradient (thread t of the 64*W threads owns elements 2(m*64W + t) + {0,1}).  It is
 * compiled together with the engine's kernels into a module:
 *     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
 *           -DNM_MODULE_DENSITY=MyDensity -DNM_MODULE_HEADER='"my_density.hpp"' -DNM_MODULE_DPL=<d> -DNM_MODULE_W=<w>
 *           -I <repo>/nuts_rs_amd/csrc -I <repo>/include  <repo>/nuts_rs_amd/csrc/density_module.hip -o my_density.so
 * for the tiling nm_pick_tiling(dim, ...) names (one tiling per module keeps the build at ~20 s), and selected with
 * kind = NM_LOGP_MODULE, module_path = "my_density.so".  `python -m nuts_rs_amd.build` has a helper
 * (nuts_rs_amd.build.build_density_module).  The module exports nm_module_launch / nm_module_info; the engine checks
 * that it was built against the same kernel-parameter layout.
 * Errors (optional; the `Result<f64, E>` of `CpuLogpFunc::logp` with `LogpError::is_recoverable`, src/math/math.rs:9-13): a density
 * that declares `static constexpr bool kCanFail = true;`, an `int status` member and `bind(const KParams&, uint64_t chain)` sets
 * `status` in every eval — 0, 1 (recoverable: the leapfrog is a divergence without an energy error,
 * src/dynamics/transformed_hamiltonian.rs:562-578) or 2 (unrecoverable: the chain stops with NM_CHAIN_LOGP_FATAL) — the same value in
 * every thread of the chain (tests/user_density/my_walled_normal.hpp).  One-chain kernels only (no group form).
 * Wide chains (dim > 4096: several blocks per chain, see nm_engine_blocks_per_chain): the module is compiled with -DNM_CLUSTER_MODE=1
 * -DNM_MODULE_DPL=16 -DNM_MODULE_W=4 (build_density_module does it) and the density also defines
 * `init_slice(params, dim, gdim, goff, reducer)`: its block holds elements [goff, goff + dim) of a chain of gdim elements, `eval` sees
 * that slice, and every reducer sum spans the whole chain (tests/user_density/my_diag_normal.hpp).
 * Group form (optional, dim <= 64): many small chains are drawn several per wavefront (nm_engine_config.lane_groups).
 * A module takes part if it also defines `template <class L> struct MyDensityGroup` with `set_lds(ptr)`,
 * `init(params, dim)` and `double eval(const double (&x)[2], double (&grad)[2], int dim) const`, where this lane holds
 * elements 2 L::lane(), 2 L::lane() + 1 and L::sum / L::bcast combine the chain's lanes, and is built with
 * -DNM_MODULE_GROUP_DENSITY=MyDensityGroup -DNM_MODULE_GS=<8|16|32 for dim <= 16|32|64> (tests/user_density/ has one).
 * Lane form (optional, dim <= 10): one chain per lane (nm_engine_config.lane_chains): `template <int NP> struct MyDensityLane` with
 * `init(params, dim)` and `double eval(const double (&x)[2 NP], double (&gx)[2 NP], int dim) const`, -DNM_MODULE_LANE_DENSITY=MyDensityLane.
 * Other samplers (dim <= 4096): -DNM_MODULE_VARIANTS=<bits> compiles the SAME functor into the kernels that carry the low-rank
 * transformation (bit 0: adaptation = NM_ADAPT_LOW_RANK, `LowRankNutsSettings`) and into those with the non-Euclidean trajectory kinds
 * and MCLMC (bit 1); build_density_module(..., variants=("low_rank", "kinetic")).  Without them the engine answers
 * NM_ERR_UNSUPPORTED for those settings with this module.
 * ------------------------------------------------------------------------------------------- */
/* The tiling the engine uses for `dim` (requested_* = 0: automatic): doubles per lane and waves per chain. */
nm_status nm_pick_tiling(uint64_t dim, uint64_t requested_dims_per_lane, uint64_t requested_waves_per_chain,
                         uint64_t* dims_per_lane, uint64_t* waves_per_chain);

/* ---------------------------------------------------------------------------------------------
 * Per-draw, per-chain statistics.  Union of the reference's `Progress` (src/sampler.rs:165-174)
 * and the scalar fields of `NutsStats` and its flattened parts (src/chain.rs:215-232,
 * src/stepsize/adapt.rs:274-306, src/dynamics/transformed_hamiltonian.rs:96-112,
 * src/dynamics/hamiltonian.rs:38-55).  Names are the reference's stat names.
 * ------------------------------------------------------------------------------------------- */
typedef struct nm_draw_stats {
    uint64_t draw;                 /* Progress.draw: index of this draw (pre-increment, chain.rs:175) */
    uint64_t chain;                /* global chain id */
    uint64_t depth;                /* SampleInfo.depth */
    uint64_t maxdepth_reached;
    uint64_t diverging;
    uint64_t tuning;               /* strategy.is_tuning() after adapt (chain.rs:178) */
    uint64_t n_steps;              /* Progress.num_steps = leapfrogs of this draw incl. a divergent one */
    int64_t  index_in_trajectory;  /* of the chosen point */
    int64_t  transformation_index; /* transform_id of the chosen point */
    double   step_size;            /* AFTER adapt = step size of the NEXT draw (chain.rs:179) */
    double   step_size_bar;
    double   mean_tree_accept;
    double   mean_tree_accept_sym;
    double   max_energy_error;
    double   logp;
    double   energy;
    double   energy_error;
    double   fisher_distance;      /* sum (z+g_z)^2, math.rs:92 */
    double   divergence_energy_error; /* NaN when not diverging or unknown */
    uint64_t chain_status;         /* NM_CHAIN_* */
    int64_t  transformation_update_id; /* DiagMassMatrixStats.transformation_update_id (src/transform/diagonal.rs:32-71):
                                      the mass-matrix version if it differs from the one seen at the previous draw's
                                      statistics (first draw: compared with -1, src/sampler.rs:795), else -1 (= None) */
    uint64_t num_eigenvalues;      /* MatrixStats.num_eigenvalues (src/transform/low_rank.rs:205-229) on the draws where
                                      transformation_update_id >= 0 (0 while the transformation has no low-rank part), else 0 */
    double   energy_change;        /* MclmcStats.energy_change (= log_weight, src/mclmc.rs:91-124, :438-439); NaN for NUTS draws */
    double   average_step_size;    /* MclmcStats.average_step_size: trajectory time / leapfrogs taken; NaN for NUTS draws */
} nm_draw_stats;
/* Notes on reference naming.  `NutsStats.draw` (src/chain.rs:215-232) is read AFTER `draw_count += 1`
 * (src/chain.rs:184, :195) and therefore equals nm_draw_stats.draw + 1; `divergence_draw` likewise.
 * `divergence_message` is a pure function of divergence_energy_error (src/dynamics/hamiltonian.rs:86-98):
 * NaN -> "Divergence due to NaN energy error", else "Divergence due to large energy error: {:.4}". */

/* Where one call of nm_engine_draw_ex records its results.  Every pointer is a DEVICE pointer or NULL (not
 * recorded).  Vector outputs are [n_draws][n_chains][dim] row-major; the reference's store_* settings
 * (src/sampler.rs:218-226, :236) decide which of them a caller allocates:
 *   store_unconstrained -> d_positions (PointStats.unconstrained_draw == the draw itself)
 *   store_gradient      -> d_gradient
 *   store_transformed   -> d_transformed_position, d_transformed_gradient
 *   store_mass_matrix   -> d_mass_matrix_inv, d_transformation_mu   (event rows)
 *   store_divergences   -> d_divergence_start, d_divergence_start_gradient, d_divergence_end   (event rows)
 * Event rows are written only on the draws where the event happens (stats.transformation_update_id >= 0, resp.
 * stats.diverging != 0); other rows are left untouched.  `divergence_momentum` is always None on this path
 * (start_momentum: None, src/dynamics/transformed_hamiltonian.rs:567, :596) and has no buffer. */
typedef struct nm_draw_outputs {
    double*        d_positions;               /* the draws (State::write_position, src/chain.rs:163-164) */
    nm_draw_stats* d_stats;                   /* [n_draws][n_chains] */
    double*        d_gradient;                /* PointStats.gradient: untransformed gradient at the draw */
    double*        d_transformed_position;    /* PointStats.transformed_position  z = (x - mu)/sigma */
    double*        d_transformed_gradient;    /* PointStats.transformed_gradient  g_z = g_x * sigma */
    double*        d_mass_matrix_inv;         /* DiagMassMatrixStats.mass_matrix_inv = the `stds` vector (diagonal.rs:52-56) */
    double*        d_transformation_mu;       /* DiagMassMatrixStats.transformation_mu = the `mean` vector */
    double*        d_divergence_start;        /* DivergenceInfo.start_location (position the divergent leapfrog started from) */
    double*        d_divergence_start_gradient;
    double*        d_divergence_end;          /* DivergenceInfo.end_location */
    double*        d_mass_matrix_eigvals;     /* MatrixStats.mass_matrix_eigvals (low_rank.rs:232-243): lambda^(1/2) of the low-rank
                                                 part, NaN beyond num_eigenvalues; event row (NM_ADAPT_LOW_RANK only;
                                                 d_mass_matrix_inv then carries MatrixStats.mass_matrix_stds) */
    uint64_t       reserved[5];
} nm_draw_outputs;

typedef struct nm_engine nm_engine;

/* Engine-side execution knobs (not reference settings). */
typedef struct nm_engine_config {
    int64_t  device;               /* HIP device ordinal; -1 = current device */
    uint64_t chain_id_offset;      /* global id of local chain 0 (multi-GPU sharding: rank*r n_local) */
    uint64_t dims_per_lane;        /* 0 = auto.  doubles per lane of every live vector: 2, 4, 8 or 16 */
    uint64_t waves_per_chain;      /* 0 = auto.  1, 2 or 4 wavefronts cooperate on one chain (dim <= 64*waves*dims_per_lane) */
    uint64_t grid_blocks;          /* 0 = auto (resident blocks of the chip).  Blocks stride over the chains */
    uint64_t lane_groups;          /* chains with dim <= 16 / 32 / 64: draw them 8 / 4 / 2 per wavefront instead of one per wavefront, same results.
                                    * 0 = auto (when there are more chains than resident wavefronts, ~2048), 1 = never, 2 = whenever the kernel applies */
    uint64_t chain_tiles;          /* the full-precision normal at dim <= 256: draw 16 chains per block with the dense
                                    * products on the matrix cores (v_mfma_f64_16x16x4_f64), same results.  (a) DiagNutsSettings: P x of the density,
                                    * every chain keeps its own adapting mass matrix; (b) one low-rank transformation shared by all chains
                                    * (nm_engine_set_transform per_chain = 0, freeze_transform; dim and rank multiples of 8): U'z, U s and P x.
                                    * 0 = auto ((a) from 256 chains on, (b) whenever it applies), 1 = never, 2 = whenever it applies */
    uint64_t lowrank_max_rank;     /* NM_ADAPT_LOW_RANK: eigenvector slots per chain (HBM: (max_rank + 1) x dim f64 per chain).
                                    * 0 = auto: min(dim, 2 (num_tune + 1)) — the most the reference's estimator can return — or dim with freeze_transform */
    uint64_t lane_chains;          /* chains with dim <= 10: ONE CHAIN PER LANE, 64 chains per wavefront (nuts_lane.hpp), same results.  DiagNutsSettings,
                                    * Euclidean NUTS, maxdepth + extra_doublings <= 10, the built-in iid / diagonal normal, funnel and (dim 10) 8-schools densities.
                                    * 0 = auto (dim <= 4 from 32768 chains on, dim <= 10 from 49152, never for Neal's funnel: the crossovers against the
                                    * 8-lane kernels measured by tools/crossover_sweep.py, profiles/r05m_crossovers.json), 1 = never, 2 = whenever the kernel
                                    * applies.  Takes precedence over lane_groups. */
} nm_engine_config;
void nm_engine_config_default(nm_engine_config* c);

/* Create an engine that owns `n_chains` chains (replaces `settings.new_chain(chain, math, rng)` for
 * chain = chain_id_offset .. chain_id_offset+n_chains, reference src/sampler.rs:745-772).
 * Chain RNGs follow the `Sampler` seeding (reference src/sampler.rs:1105-1106, :761):
 * outer = ChaCha8(seed_from_u64(settings.seed), stream = chain_id+1); chain rng = ChaCha8 keyed by
 * the first 32 bytes of outer. */
nm_status nm_engine_create(const nm_settings* settings, const nm_logp_spec* logp,
                           uint64_t n_chains, const nm_engine_config* cfg, nm_engine** out);
void      nm_engine_destroy(nm_engine* e);

/* `Chain::set_position` for every chain (reference src/chain.rs:137-149).  h_x0 is [n_chains][dim].
 * h_chain_status (optional) receives NM_CHAIN_* per chain.  Returns NM_ERR_BAD_INIT /
 * NM_ERR_LOGP_FAILURE if any chain failed (the other chains are still initialised). */
nm_status nm_engine_set_positions(nm_engine* e, const double* h_x0, uint64_t* h_chain_status);
/* The same for the chains with h_mask[c] != 0 only (h_mask NULL = all): the others keep their state untouched — their
 * adapted mass matrix, step-size adaptation and random stream.  This is the per-chain `Chain::set_position` the
 * reference's init loop needs: a chain whose initial point fails with BadInitGrad draws another one and tries again,
 * up to 500 times (src/sampler.rs:1133-1147), without disturbing the chains that started.  Calling it again for a
 * chain does what the reference's second `set_position` does: the mass-matrix estimators take the new point as one
 * more sample (src/transform/adapt/diagonal.rs:209-231), the mass matrix is re-derived from its gradient, the
 * step-size search runs again, the random stream continues. */
nm_status nm_engine_set_positions_masked(nm_engine* e, const double* h_x0, const uint8_t* h_mask, uint64_t* h_chain_status);
/* The whole init loop of the reference's ChainProcess (src/sampler.rs:1126-1147): attempt 0 is h_x0 (NULL: the uniform
 * init_position), every failed chain is retried with its next init_position, at most max_tries (reference: 500) in all.
 * h_tries (optional) receives the attempts used per chain.  Returns NM_ERR_BAD_INIT if a chain never started. */
nm_status nm_engine_init_positions_retry(nm_engine* e, const double* h_x0, uint64_t max_tries, uint64_t* h_chain_status, uint64_t* h_tries);

/* The `init_position` of the reference's CpuMath (src/math/cpu_math.rs:171-199): x0 ~ U(-1,1) drawn
 * from each chain's OUTER generator right after the 32 seed bytes (Sampler order, src/sampler.rs:1126-1136).
 * Fills h_x0 [n_chains][dim]; pure host helper. */
nm_status nm_init_positions_uniform(uint64_t seed, uint64_t chain_id_offset, uint64_t n_chains,
                                    uint64_t dim, double* h_x0);
/* The `attempt`-th init_position of every chain (attempt 0 is nm_init_positions_uniform): the outer generator keeps
 * running through the retries of the init loop (src/sampler.rs:1133-1143). */
nm_status nm_init_positions_uniform_at(uint64_t seed, uint64_t chain_id_offset, uint64_t n_chains, uint64_t dim,
                                       uint64_t attempt, double* h_x0);

/* Advance ALL chains by n_draws draws (`Chain::draw` x n_draws per chain, reference src/chain.rs:151-188).
 * The engine launches on its OWN non-blocking stream (nm_engine_stream): output buffers must not have work pending on
 * another stream (e.g. a framework's asynchronous zero-fill) when this is called.
 * The whole loop runs on the device; this call enqueues the launches and returns after they finish.
 *   d_positions : device buffer [n_draws][n_chains][dim] or NULL (positions not recorded)
 *   d_stats     : device buffer [n_draws][n_chains] of nm_draw_stats or NULL
 * Both may also be fetched afterwards with the nm_engine_read_* helpers. */
nm_status nm_engine_draw(nm_engine* e, uint64_t n_draws, double* d_positions, nm_draw_stats* d_stats);

/* `Chain::expanded_draw` x n_draws (reference src/chain.rs:190-204): nm_engine_draw plus the vector-valued
 * statistics selected by the non-NULL pointers of `out`.  Asynchronous variant: returns after enqueueing. */
nm_status nm_engine_draw_ex(nm_engine* e, uint64_t n_draws, const nm_draw_outputs* out);
nm_status nm_engine_draw_ex_async(nm_engine* e, uint64_t n_draws, const nm_draw_outputs* out);

/* Same, asynchronous on the engine's stream: returns after enqueueing. */
nm_status nm_engine_draw_async(nm_engine* e, uint64_t n_draws, double* d_positions, nm_draw_stats* d_stats);
nm_status nm_engine_synchronize(nm_engine* e);

/* Run n_draws and deliver the results to host buffers ([n_draws][n_chains][dim] / [n_draws][n_chains]) — the shape of the
 * reference's `Chain::draw() -> (Box<[f64]>, Stats)` consumers.  The launch is cut into chunks that pass through two sets of
 * device staging buffers: the kernel of chunk i + 1 runs while chunk i crosses PCIe on a copy stream (same draws as one launch).
 * The copies reach the PCIe rate when the destination is pinned (nm_host_register) or has been written before; fresh pageable
 * memory is bound by its page faults. */
nm_status nm_engine_draw_to_host(nm_engine* e, uint64_t n_draws, double* h_positions, nm_draw_stats* h_stats);
/* Pin / unpin a host array (hipHostRegister) that *_to_host calls will fill repeatedly. */
nm_status nm_host_register(void* h_ptr, uint64_t bytes);
nm_status nm_host_unregister(void* h_ptr);

/* nm_engine_draw_ex with HOST destinations: the pointers of `h_out` are host arrays of the same shapes; event
 * rows that were not written read as NaN. */
nm_status nm_engine_draw_ex_to_host(nm_engine* e, uint64_t n_draws, const nm_draw_outputs* h_out);

/* ---------------------------------------------------------------------------------------------
 * The low-rank transformation (NM_ADAPT_LOW_RANK; reference `LowRankMassMatrix`, src/transform/low_rank.rs:95-186):
 *     F(y) = sigma . (I + U (diag(lambda)^1/2 - I) U') (y + mu_lr) + mean
 * applied inside the fused leapfrog (x' = F(z'), g_z' = J_F' g_x'; src/transform/low_rank.rs:325-398,
 * src/math/cpu_math.rs:332-425).  Its adaptation (`LowRankMassMatrixStrategy`, src/transform/adapt/low_rank.rs) keeps
 * the window of draws / gradients on the device; the dense linear algebra of `compute_update` (thin SVDs, pivoted
 * QR, three symmetric eigendecompositions — faer in the reference) runs on the host between kernel launches: a chain
 * whose schedule asks for a new matrix pauses, the host estimates it from the chain's window and uploads it, the
 * next launch resumes the chain at the same point of GlobalStrategy::adapt.  nm_engine_draw* do all of that.
 * ------------------------------------------------------------------------------------------- */
/* The estimator: draws / grads are [n_draws][dim] (oldest first); outputs: stds[dim], mean[dim], *n_eig <= min(dim, 2 n_draws),
 * vals[*n_eig], vecs[*n_eig][dim] (one eigenvector per row), mu_low_rank[dim].  Returns 0 (Some) or 1 (None: no update).
 * Called from several host threads at once (one chain each).  The default is the built-in C++ restatement of
 * compute_update (nuts_rs_amd/csrc/lowrank_host.cpp). */
typedef int (*nm_lowrank_estimator_fn)(void* ctx, uint64_t dim, uint64_t n_draws, const double* draws, const double* grads,
                                       double gamma, double eigval_cutoff, double* stds, double* mean, uint64_t* n_eig,
                                       double* vals, double* vecs, double* mu_low_rank);
nm_status nm_engine_set_lowrank_estimator(nm_engine* e, nm_lowrank_estimator_fn fn, void* ctx, uint64_t n_threads /* 0 = all cores */);
/* The built-in estimator itself (host only, no device needed): for tests and for callers that adapt elsewhere. */
int nm_lowrank_compute_update(void* unused, uint64_t dim, uint64_t n_draws, const double* draws, const double* grads,
                              double gamma, double eigval_cutoff, double* stds, double* mean, uint64_t* n_eig,
                              double* vals, double* vecs, double* mu_low_rank);
/* Test hooks into the built-in estimator (host only): the two routines the reference's own unit tests exercise
 * (`spd_mean`, `estimate_mass_matrix`: src/transform/adapt/low_rank.rs:228-290, tests :354-407), so that its vectors run
 * against THIS implementation and not only against the oracle's LAPACK restatement.  Matrices are column-major.
 * spd_mean: n x n inputs -> out n x n.  estimate_mass_matrix: draws / grads are (rows x n_draws) -> vals[rows] ascending,
 * vecs rows x rows (one eigenvector per column).  force_base: 0 the widest ISA build, 1 the baseline-ISA build even where AVX2 is
 * available, 2 the block form's twin (= the device kernel's arithmetic, see nm_lowrank_block_twin).
 * Return 0 (Some) or 1 (None). */
int nm_lowrank_test_spd_mean(uint64_t n, const double* cov_draws, const double* cov_grads, double* out, uint64_t force_base);
int nm_lowrank_test_estimate_mass_matrix(uint64_t rows, uint64_t n_draws, const double* draws, const double* grads, double gamma,
                                         double* vals, double* vecs, uint64_t force_base);

/* WHERE the built-in estimator runs.  The reference runs `compute_update` (src/transform/adapt/low_rank.rs:73-142) in the thread
 * that runs the chain; here the default (NM_LR_PLACE_AUTO) is the device that runs the chains: one 256-thread block per paused
 * chain works on the chain's window where the draw kernel left it (csrc/lowrank_device.hip, the block form of the same
 * algorithm: csrc/lowrank_block.hpp), for dim <= 512 and windows of <= 1024 draws; other shapes, and any estimator set with
 * nm_engine_set_lowrank_estimator, run on host threads.  NM_LR_PLACE_HOST: always the host threads.  NM_LR_PLACE_DEVICE: the
 * device or NM_ERR_UNSUPPORTED at the first window it does not take.  The device form sums in another order than the host form:
 * the two agree to rounding on full-rank windows and within the reference algorithm's own conditioning on rank-deficient ones
 * (tests/test_lowrank_estimator_builtin.py states the tolerances for both against the literal reference algorithm). */
#define NM_LR_PLACE_AUTO 0
#define NM_LR_PLACE_HOST 1
#define NM_LR_PLACE_DEVICE 2
nm_status nm_engine_set_lowrank_estimator_place(nm_engine* e, uint64_t place);
uint64_t  nm_engine_lowrank_device_updates(const nm_engine* e);   /* estimator calls that ran on the device so far */
/* The device form's host TWIN (nm_lowrank_estimator_fn's signature; host only): the same steps walked by one thread with the
 * kernel's reduction trees — bit for bit what the kernel computes (tests/test_gpu_lowrank.py), so CPU tests of the twin are
 * statements about the kernel.  Returns 1 (None) also for shapes the block algorithm does not take. */
int nm_lowrank_block_twin(void* unused, uint64_t dim, uint64_t n_draws, const double* draws, const double* grads,
                          double gamma, double eigval_cutoff, double* stds, double* mean, uint64_t* n_eig,
                          double* vals, double* vecs, double* mu_low_rank);
/* Test hook: the estimator KERNEL on a batch of windows of one shape.  draws / grads [n_windows][n_draws][dim]; outputs carry a
 * leading [n_windows] axis, vals [kmax] and vecs [kmax][dim] with kmax = min(dim, 2 n_draws); status[w] = 0 (Some) / 1 (None);
 * logdet_bits[w] (optional) = the bits of -1/2 sum ln lambda.  Returns a hipError_t (0 = success). */
int nm_lowrank_test_block_device(uint64_t dim, uint64_t n_draws, uint64_t n_windows, const double* draws, const double* grads,
                                 double gamma, double eigval_cutoff, double* stds, double* mean, uint64_t* n_eig, double* vals,
                                 double* vecs, double* mu_low_rank, uint64_t* status, uint64_t* logdet_bits);

/* `LowRankMassMatrix::update(stds, mean, vals, vecs, mean_low_rank)` (src/transform/low_rank.rs:155-186) for every chain,
 * from the host: the transformation version moves on and the current points are re-whitened lazily at their next
 * trajectory (src/dynamics/transformed_hamiltonian.rs:706-720).  per_chain = 0: one transformation for all chains
 * (h_stds[dim], h_mean[dim], h_vals[n_eig], h_vecs[n_eig][dim], h_mu_low_rank[dim]); 1: every array has a leading
 * [n_chains] axis.  Chains whose input is not finite keep their transformation (the reference returns early).
 * Needs settings.adaptation == NM_ADAPT_LOW_RANK; with settings.freeze_transform the schedule never replaces it. */
nm_status nm_engine_set_transform(nm_engine* e, uint64_t per_chain, uint64_t n_eig, const double* h_stds, const double* h_mean,
                                  const double* h_vals, const double* h_vecs, const double* h_mu_low_rank);
/* Current low-rank part per chain: h_n_eig[n_chains]; optional h_vals [n_chains][max_rank] (lambda^(1/2)),
 * h_vecs [n_chains][max_rank][dim], h_mu_low_rank [n_chains][dim]. */
nm_status nm_engine_get_lowrank(nm_engine* e, uint64_t* h_n_eig, double* h_vals_sqrt, double* h_vecs, double* h_mu_low_rank);
uint64_t  nm_engine_lowrank_max_rank(const nm_engine* e);
/* draw launches served by the 16-chains-per-block matrix-core kernel so far (nm_engine_config.chain_tiles) */
uint64_t  nm_engine_tile_launches(const nm_engine* e);
/* ... of which by its LOCKSTEP form (nuts_lockstep.hpp: the 16 chains of a block advance together, draws not synchronised) */
uint64_t  nm_engine_lockstep_launches(const nm_engine* e);
/* The order in which this engine's draws sum over dim, for callers that compare with the CPU oracle bit for bit: 0 = the wave
 * kernels' (threads_per_chain), 1 = the matrix-core tile kernel (the same, low-rank products as sequential dots), 2 = the lockstep
 * kernel's stripe order from nm_engine_set_transform on.  Fixed when the shared transformation is set. */
uint64_t  nm_engine_reduce_order(const nm_engine* e);
/* calls of the host density function so far (NM_LOGP_HOST_CALLBACK) */
uint64_t  nm_engine_host_logp_calls(const nm_engine* e);

/* Current per-chain quantities, host copies ([n_chains][dim] unless noted). */
nm_status nm_engine_get_positions(nm_engine* e, double* h_x);
nm_status nm_engine_get_gradients(nm_engine* e, double* h_gx);
nm_status nm_engine_get_mass_matrix(nm_engine* e, double* h_stds, double* h_mean);   /* sigma, mu */
nm_status nm_engine_get_step_sizes(nm_engine* e, double* h_step_size /*[n_chains]*/);

/* Totals since creation (for the metric): sum over chains of leapfrog steps, and device time in ms
 * spent inside the draw kernels (HIP events on the engine's stream). */
nm_status nm_engine_get_counters(nm_engine* e, uint64_t* total_leapfrogs, uint64_t* total_draws,
                                 double* kernel_ms, uint64_t* kernel_launches);
nm_status nm_engine_reset_counters(nm_engine* e);

uint64_t  nm_engine_dim(const nm_engine* e);
uint64_t  nm_engine_num_chains(const nm_engine* e);
/* The tiling the engine chose: threads cooperating on one chain (64 * waves_per_chain) and doubles per lane.
 * The reduction order over dim (and so the last bits of every sum) is a function of threads_per_chain;
 * the oracle reproduces it with gpu_cfg(threads_per_chain). */
uint64_t  nm_engine_threads_per_chain(const nm_engine* e);
uint64_t  nm_engine_dims_per_lane(const nm_engine* e);
/* dim > 4096: a chain is spread over ceil(dim / 4096) co-resident blocks of 256 threads that each own a 4096-element slice
 * and exchange their block sums (1 for dim <= 4096).  Sums are then slice totals added in slice order; the oracle reproduces
 * it with gpu_cfg(threads_per_chain, slice = 4096).  NM_LOGP_IID_NORMAL, NM_LOGP_DIAG_NORMAL, NM_LOGP_MODULE (a fused user density built for
 * wide chains) and NM_LOGP_HOST_CALLBACK (any density); NUTS with every
 * trajectory_kind and NM_SAMPLER_MCLMC, with the diagonal adaptation (not the low-rank one); dim <= 131072.  The blocks of a chain wait for each other inside the kernel: the grid never exceeds what the
 * device holds at once, which assumes the device is not shared with another such engine running at the same time. */
uint64_t  nm_engine_blocks_per_chain(const nm_engine* e);
/* draw launches served by the several-chains-per-wavefront kernels so far (nm_engine_config.lane_groups) */
uint64_t  nm_engine_group_launches(const nm_engine* e);
/* draw launches served by the one-chain-per-lane kernels (nm_engine_config.lane_chains) */
uint64_t  nm_engine_lane_launches(const nm_engine* e);
/* The HIP stream the engine launches on (a hipStream_t), so callers can order their own work. */
void*     nm_engine_stream(nm_engine* e);

/* ---------------------------------------------------------------------------------------------
 * Batched `Math` primitives on device vectors ([n][dim] row-major, one row per chain): the
 * per-vector seam of the reference (src/math/math.rs:15-314), exported for unit-parity tests of the
 * kernels' building blocks.  All pointers are DEVICE pointers; `stream` is a hipStream_t or NULL.
 * ------------------------------------------------------------------------------------------- */
/* One fused leapfrog for n independent chains (reference transformed_hamiltonian.rs:524-615 with
 * DiagMassMatrix, src/transform/diagonal.rs:196-209).  In: z,v,g_z,sigma,mu [n][dim]; eps[n], logdet[n],
 * initial_energy[n].  Out: z',v',g_z',x',g_x' [n][dim]; logp'[n], kinetic'[n], energy_error[n]. */
nm_status nm_leapfrog_batch(const nm_logp_spec* logp, uint64_t n, uint64_t dims_per_lane,
                            const double* d_z, const double* d_v, const double* d_gz,
                            const double* d_sigma, const double* d_mu,
                            const double* d_eps, const double* d_logdet, const double* d_initial_energy,
                            double* d_z_out, double* d_v_out, double* d_gz_out,
                            double* d_x_out, double* d_gx_out,
                            double* d_logp_out, double* d_kinetic_out, double* d_energy_error_out,
                            void* stream);

/* The three maps of the low-rank transformation for n independent chains (unit parity of low_rank.rs:325-398):
 * which 0 compute_transformed_position (x -> z), 1 compute_untransformed_position (z -> x), 2 compute_transformed_gradient
 * (g_x -> g_z).  d_stds, d_mean, d_mu_low_rank, d_in, d_out: [n][dim]; d_vals [n][n_eig] (the RAW eigenvalues lambda),
 * d_vecs [n][n_eig][dim]. */
nm_status nm_lowrank_transform_batch(uint64_t which, uint64_t n, uint64_t dim, uint64_t n_eig, uint64_t dims_per_lane,
                                     const double* d_stds, const double* d_mean, const double* d_vals, const double* d_vecs,
                                     const double* d_mu_low_rank, const double* d_in, double* d_out, void* stream);

/* U-turn criterion `is_turning` (reference transformed_hamiltonian.rs:617-638 via scalar_prods3,
 * src/math/util.rs:221-347) for n pairs: out_t[2*i] = (z_end - z_start).v_start, out_t[2*i+1] = (..).v_end. */
nm_status nm_turning_batch(uint64_t n, uint64_t dim, uint64_t dims_per_lane,
                           const double* d_z_start, const double* d_v_start,
                           const double* d_z_end, const double* d_v_end,
                           double* d_out_t, void* stream);

/* Scalar special functions used on the device (deterministic restatements):
 * op 0 exp, 1 ln, 2 ln_1p, 3 logaddexp(a,b), 4 sqrt, 5 a/b, 7 exp_m1, 8 sin, 9 cos.  d_a, d_b, d_out are device arrays [n]. */
nm_status nm_scalar_math_batch(uint64_t op, uint64_t n, const double* d_a, const double* d_b,
                               double* d_out, void* stream);

/* Standard-normal stream of a chain generator (reference array_gaussian, src/math/cpu_math.rs:561-577):
 * fills d_out[n][count] with the first `count` N(0,1) variates of ChaCha8(key=h_keys[i], stream 0). */
nm_status nm_standard_normal_batch(uint64_t n, uint64_t count, const uint8_t* h_keys /*[n][32]*/,
                                   double* d_out, uint64_t* h_words_consumed /*[n] or NULL*/, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The per-vector seam: `trait Math` (reference src/math/math.rs:15-314) over device vectors, one function per hot-path
 * method.  An nm_math is one `CpuMath<F>` (src/math/cpu_math.rs:19-41: a density, its dim, a
 * stream), an nm_vec its `M::Vector` (device resident, opaque).  One chain per Math, one small launch per method: on a
 * GPU this seam is launch-latency bound — it exists so that code written against `Math` has a target and so that the
 * fused kernels' building blocks can be tested one by one; throughput lives in the batched engine above.  Every method
 * runs the engine's arithmetic: FMAs where the reference writes mul_add, reductions in the engine's documented order
 * (the oracle's gpu_cfg(nm_math_threads)).  Built-in densities only (their `LogpErr` cannot occur).
 * ------------------------------------------------------------------------------------------- */
typedef struct nm_math nm_math;
typedef struct nm_vec nm_vec;
nm_status nm_math_create(const nm_logp_spec* logp, nm_math** out);                    /* CpuMath::new (cpu_math.rs:32-41) */
void      nm_math_destroy(nm_math* m);
uint64_t  nm_math_dim(const nm_math* m);                                              /* Math::dim (math.rs:22) */
uint64_t  nm_math_threads(const nm_math* m);                                          /* threads per vector: fixes the reduction order */
const char* nm_math_last_error(void);
nm_status nm_vec_new(nm_math* m, nm_vec** out);                                       /* M13 new_array: zeros (math.rs:24) */
void      nm_vec_free(nm_vec* v);
nm_status nm_vec_read_from_slice(nm_math* m, nm_vec* dest, const double* h_source);  /* M13 (math.rs:101) */
nm_status nm_vec_write_to_slice(nm_math* m, const nm_vec* source, double* h_dest);   /* M13 (math.rs:103) */
nm_status nm_vec_copy_into(nm_math* m, const nm_vec* source, nm_vec* dest);          /* M13 (math.rs:94) */
nm_status nm_vec_fill_array(nm_math* m, nm_vec* dest, double value);                 /* M13 (math.rs:119) */
nm_status nm_vec_array_recip(nm_math* m, const nm_vec* a, nm_vec* dest);             /* M13 (math.rs:125) */
nm_status nm_vec_axpy_out(nm_math* m, const nm_vec* x, const nm_vec* y, double a, nm_vec* out);     /* M1 out = fma(a, x, y) (math.rs:98) */
nm_status nm_vec_axpy(nm_math* m, const nm_vec* x, nm_vec* y, double a);             /* M2 y = fma(a, x, y) (math.rs:99) */
nm_status nm_vec_array_mult(nm_math* m, const nm_vec* a, const nm_vec* b, nm_vec* dest);            /* M3 (math.rs:123-124; dest may be a) */
nm_status nm_vec_array_vector_dot(nm_math* m, const nm_vec* a, const nm_vec* b, double* out);       /* M4 (math.rs:212) */
nm_status nm_vec_scalar_prods3(nm_math* m, const nm_vec* positive1, const nm_vec* negative1, const nm_vec* positive2,
                               const nm_vec* x, const nm_vec* y, double out[2]);     /* M5 (math.rs:75-82) */
/* M6 array_gaussian (math.rs:213-218): dest_i = stds_i * N(0,1) from ChaCha8(key) at *stream_pos (u32 words), which is advanced */
nm_status nm_vec_array_gaussian(nm_math* m, const uint8_t key[32], uint64_t* stream_pos, nm_vec* dest, const nm_vec* stds);
nm_status nm_vec_array_update_variance(nm_math* m, nm_vec* mean, nm_vec* variance, const nm_vec* value, double diff_scale);   /* M7 (math.rs:227-233) */
nm_status nm_vec_array_update_var_inv_std_draw_grad(nm_math* m, nm_vec* inv_std, nm_vec* std_, const nm_vec* draw_var, const nm_vec* grad_var,
                                                    uint64_t has_fill_invalid, double fill_invalid, double clamp_lo, double clamp_hi);   /* M8 (math.rs:243-251) */
nm_status nm_vec_array_update_var_inv_std_grad(nm_math* m, nm_vec* inv_std, nm_vec* std_, const nm_vec* gradient, double fill_invalid,
                                               double clamp_lo, double clamp_hi);   /* M9 (math.rs:253-260) */
nm_status nm_vec_array_update_var_inv_std_draw(nm_math* m, nm_vec* inv_std, nm_vec* std_, const nm_vec* draw_var, double scale,
                                               uint64_t has_fill_invalid, double fill_invalid, double clamp_lo, double clamp_hi);   /* M10 (math.rs:234-242) */
nm_status nm_vec_array_sum_ln(nm_math* m, const nm_vec* a, double* out);             /* M11 (math.rs:113-117) */
nm_status nm_vec_array_all_finite(nm_math* m, const nm_vec* a, uint64_t and_nonzero, uint64_t* out);   /* M12 (math.rs:121-122) */
/* M14 logp_array (math.rs:46-50): fills `gradient`, *logp; *status = 0 ok / 1 recoverable / 2 fatal (always 0 here) */
nm_status nm_vec_logp_array(nm_math* m, const nm_vec* position, nm_vec* gradient, double* logp, uint64_t* status);
nm_status nm_vec_sq_norm_sum(nm_math* m, const nm_vec* x, const nm_vec* y, double* out);             /* M15 (math.rs:92) */
/* the Math methods of the non-Euclidean KineticEnergyKinds (src/math/math.rs:155-200; src/math/util.rs:507-741,
 * src/math/cpu_math.rs:496-551).  `vel_out` may be `vel` (std_norm_grad_flow_inplace). */
nm_status nm_vec_std_norm_flow(nm_math* m, const nm_vec* pos, nm_vec* pos_out, nm_vec* vel, double epsilon);
nm_status nm_vec_std_norm_grad_flow(nm_math* m, const nm_vec* pos, const nm_vec* grad, const nm_vec* vel, nm_vec* vel_out, double epsilon);
nm_status nm_vec_esh_momentum_update(nm_math* m, const nm_vec* gradient, nm_vec* momentum, double step_size, double* kinetic_energy_change);
nm_status nm_vec_array_normalize(nm_math* m, nm_vec* v);

/* Chain RNG key derivation (host helper; reference src/sampler.rs:1105-1106, :761). */
nm_status nm_chain_rng_key(uint64_t seed, uint64_t chain_id, uint8_t key_out[32]);

/* ---------------------------------------------------------------------------------------------
 * HBM streaming probes: copy / triad / read / write with the engine's access shape
 * (16 B per lane, 1 KiB per wave instruction) over arrays of `bytes_per_array` bytes each, `iters` timed launches
 * after one warm-up launch (HIP events on the probe's own stream).  They give the roofline its measured
 * denominator on the box at hand, and they are the known-size kernels the rocprofv3 FETCH_SIZE / WRITE_SIZE
 * counters are calibrated on (tools/hbm_probe.py, profiles/r02*_hbm_calibration.json).
 * ------------------------------------------------------------------------------------------- */
#define NM_PROBE_COPY 0      /* c[i] = a[i]                 reads 1 array, writes 1 */
#define NM_PROBE_TRIAD 1     /* c[i] = fma(s, b[i], a[i])   reads 2 arrays, writes 1 */
#define NM_PROBE_READ 2      /* sum of a                    reads 1 array */
#define NM_PROBE_WRITE 3     /* c[i] = const                writes 1 array */
#define NM_PROBE_COPY_NT 4   /* copy with non-temporal stores (the cache policy of the engine's candidate / per-draw stores) */
nm_status nm_probe_bandwidth(uint64_t kind, uint64_t bytes_per_array, uint64_t iters, double* ms_per_iter,
                             uint64_t* bytes_read_per_iter, uint64_t* bytes_written_per_iter);
/* Fixed-work issue-rate calibration: `waves` one-wavefront blocks (0 = one per SIMD of the device) each run a dependent chain of `chain`
 * v_fma_f64; *ns_per_instruction = launch time / chain.  The engine's BASELINE kernels run one wavefront per SIMD and are bound by the
 * instructions that wavefront issues (one per ~4.3 cycles): this figure is the box-dependent factor of their speed (bench.py prints it
 * beside every config).  No reference counterpart (measurement support). */
nm_status nm_probe_issue(uint64_t waves, uint64_t chain, double* ns_per_instruction);

/* ---------------------------------------------------------------------------------------------
 * The per-rank half of the OPT-IN pooled adaptation (north_star's "RCCL cross-chain Welford reduction"; NOT reference
 * behaviour — every reference chain adapts alone, src/adapt_strategy.rs:24-39 — and never the default; nuts_rs_amd/pooled.py
 * drives it).  Reduces one window of recorded draws and gradients (device buffers [n_rows][dim] as nm_engine_draw_ex leaves
 * them, n_rows = window draws x local chains) to d_out[2][1 + 2 dim] = {count, mean[dim], M2[dim]} for the draws and for the
 * gradients, on `stream`, asynchronously.  With d_stats (the window's nm_draw_stats rows, same row order) a row counts only if
 * its chain is healthy and the draw is one the reference's DrawGradCollector keeps (is_good, src/transform/adapt/diagonal.rs:
 * 57-84).  The ranks exchange d_out with one all_gather (RCCL) and merge the partials in rank order (Chan).  Deterministic.
 * ------------------------------------------------------------------------------------------- */
nm_status nm_pooled_partials(uint64_t n_rows, uint64_t dim, const double* d_positions, const double* d_gradients,
                             const nm_draw_stats* d_stats, double* d_out, void* stream);
/* The exchange and the merge of the pooled adaptation, for callers that do not go through nuts_rs_amd/pooled.py (a Rust host with
 * its own RCCL communicator).  nm_pooled_exchange: one all_gather of the rank's payload d_payload[2][1 + 2 dim] into
 * d_gathered[world][2][1 + 2 dim] over `rccl_comm` (an ncclComm_t; librccl is resolved at run time) on `stream`; with world == 1 or a
 * null communicator the payload is copied.  nm_pooled_finish: merges the gathered partials in RANK ORDER (Chan; empty partials are
 * skipped) and writes the pooled diagonal transformation — d_sigma[dim] = (var draws / var grads)^(1/4) clamped to [1e-10, 1e10]
 * (1 where that is not finite and positive), d_mean[dim] = mean draws + sigma^2 mean grads, *d_count = pooled draws — what the
 * caller hands to nm_engine_set_transform (n_eig = 0) when the count is at least 3.  Both asynchronous on `stream`. */
nm_status nm_pooled_exchange(void* rccl_comm, uint64_t world, uint64_t dim, const double* d_payload, double* d_gathered, void* stream);
nm_status nm_pooled_finish(uint64_t world, uint64_t dim, const double* d_gathered, double* d_sigma, double* d_mean, double* d_count, void* stream);
const char* nm_pooled_last_error(void);

const char* nm_last_error(void);
uint64_t    nm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NUTS_AMD_H */
