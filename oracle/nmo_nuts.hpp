// nmo_nuts.hpp — CPU oracle of one NUTS chain with diagonal mass-matrix + dual-averaging adaptation.
// TEST INFRASTRUCTURE ONLY (see nmo_math.hpp).  Scalar, deterministic, one chain per object, written
// to follow the reference op-for-op and in the reference's own (recursive) control structure:
//
//   NutsTree / draw            src/nuts.rs:94-388
//   TransformedHamiltonian     src/dynamics/transformed_hamiltonian.rs:161-262, :310-351, :524-736 (Euclidean)
//   DiagMassMatrix             src/transform/diagonal.rs:85-265
//   RunningVariance / Strategy src/transform/adapt/diagonal.rs:17-236
//   GlobalStrategy             src/adapt_strategy.rs:77-222
//   stepsize::Strategy         src/stepsize/adapt.rs:52-272
//   DualAverage / collectors   src/stepsize/dual_avg.rs:34-166
//   NutsChain                  src/chain.rs:137-188
#pragma once
#include <cstdint>
#include <deque>
#include <memory>
#include <string>
#include <vector>
#include "nmo_math.hpp"
#include "nmo_rng.hpp"

namespace nmo {

typedef std::vector<double> Vec;

// Same field order as nm_settings (include/nuts_amd.h); declared independently on purpose.
struct Settings {
    uint64_t num_tune, num_draws, maxdepth, mindepth;
    double max_energy_error;
    uint64_t check_turning, extra_doublings, seed, num_chains;
    uint64_t store_gradient, store_unconstrained, store_transformed, store_divergences;
    uint64_t has_target_integration_time;
    double target_integration_time;
    double early_window, step_size_window;
    uint64_t mass_matrix_switch_freq, early_mass_matrix_switch_freq, mass_matrix_update_freq;
    double mass_matrix_window_growth;
    uint64_t store_mass_matrix, use_grad_based_estimate;
    double target_accept, initial_step;
    uint64_t has_jitter;
    double jitter;
    uint64_t step_size_method;
    double fixed_step_size;
    double da_k, da_t0, da_gamma, da_max_step_size;
    double adam_beta1, adam_beta2, adam_epsilon, adam_learning_rate;
    // which MassMatrixAdaptStrategy: 0 = DiagAdaptStrategy, 1 = LowRankMassMatrixStrategy (reference src/sampler.rs:651, :245)
    uint64_t adaptation;
    double lr_gamma, lr_eigval_cutoff;          // LowRankSettings (src/transform/low_rank.rs:188-203)
    uint64_t freeze_transform;                  // engine knob (not a reference setting): the transformation is given, never adapted
    uint64_t trajectory_kind;                   // KineticEnergyKind (src/dynamics/transformed_hamiltonian.rs:27-50, NutsSettings::trajectory_kind)
    // MclmcSettings (src/sampler.rs:266-317): sampler = 1 runs MclmcChain (src/mclmc.rs) instead of the NUTS tree
    uint64_t sampler;
    double mclmc_step_size, momentum_decoherence_length, subsample_frequency;
    uint64_t dynamic_step_size, mclmc_trajectory_kind;
    double trajectory_switch_fraction;
};
enum { TRAJ_EUCLIDEAN = 0, TRAJ_EXACT_NORMAL = 1, TRAJ_MICROCANONICAL = 2 };
enum { SAMPLER_NUTS = 0, SAMPLER_MCLMC = 1 };
enum { MCLMC_MICROCANONICAL = 0, MCLMC_EUCLIDEAN = 1, MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL = 2 };   // MclmcTrajectoryKind (src/mclmc.rs:44-70)

struct DrawStats {   // same field order as nm_draw_stats
    uint64_t draw, chain, depth, maxdepth_reached, diverging, tuning, n_steps;
    int64_t index_in_trajectory, transformation_index;
    double step_size, step_size_bar, mean_tree_accept, mean_tree_accept_sym, max_energy_error;
    double logp, energy, energy_error, fisher_distance, divergence_energy_error;
    uint64_t chain_status;
    int64_t transformation_update_id;   // DiagMassMatrixStats.transformation_update_id, -1 = None
    uint64_t num_eigenvalues;           // MatrixStats.num_eigenvalues (low_rank.rs:205-216) on update draws, else 0
    double energy_change, average_step_size;   // MclmcStats (src/mclmc.rs:91-124); NaN for NUTS draws
};

// Optional vector-valued statistics of one draw (rows of length dim; nullptr = not wanted).  The reference's
// PointStats (transformed_hamiltonian.rs:96-112), DiagMassMatrixStats (transform/diagonal.rs:32-71) and
// DivergenceStats (dynamics/hamiltonian.rs:38-55).  Event rows are written only on the draws where the event happens.
struct DrawVectors {
    double* gradient = nullptr;
    double* transformed_position = nullptr;
    double* transformed_gradient = nullptr;
    double* mass_matrix_inv = nullptr;          // where transformation_update_id >= 0
    double* transformation_mu = nullptr;
    double* divergence_start = nullptr;         // where diverging
    double* divergence_start_gradient = nullptr;
    double* divergence_end = nullptr;
    double* mass_matrix_eigvals = nullptr;      // MatrixStats.mass_matrix_eigvals: lambda^(1/2), NaN-padded to dim (low_rank.rs:232-243)
};

enum { LOGP_IID_NORMAL = 0, LOGP_DIAG_NORMAL = 1, LOGP_FUNNEL = 2, LOGP_EIGHT_SCHOOLS = 3, LOGP_MVN_PREC = 4,
       LOGP_HOST_CALLBACK = 100 };
enum { ST_OK = 0, ST_BAD_INIT = 1, ST_LOGP_FATAL = 2 };

// CpuLogpFunc::logp shape (reference src/math/cpu_math.rs:885-891): 0 ok, 1 recoverable, 2 fatal
typedef int (*host_logp_fn)(void* ctx, uint64_t dim, const double* x, double* grad, double* logp);

struct Density {
    int64_t kind = LOGP_IID_NORMAL;
    size_t dim = 0;
    Vec params;
    host_logp_fn cb = nullptr;
    void* cb_ctx = nullptr;

    int logp(const Ctx& m, const double* x, double* g, double* out) const {
        const size_t n = dim;
        switch (kind) {
        case LOGP_IID_NORMAL: {   // reference benches/sample.rs:49-62
            const double mu = params[0];
            Vec t(n);
            for (size_t i = 0; i < n; ++i) {
                double diff = x[i] - mu;
                g[i] = -diff;
                t[i] = -0.5 * diff * diff;
            }
            *out = m.sum_terms(t.data(), n);
            return 0;
        }
        case LOGP_DIAG_NORMAL: {  // diagonal-P case of MvNormal, reference src/transform/mod.rs:98-112
            Vec t(n), lt(n);
            for (size_t i = 0; i < n; ++i) {
                double px = params[i] * x[i];
                g[i] = -px;
                t[i] = x[i] * px;
                lt[i] = m.ln(params[i]);
            }
            double quad = -0.5 * m.sum_terms(t.data(), n);
            double log_det_p = m.sum_terms(lt.data(), n);
            double norm = -0.5 * ((double)n * m.ln(6.283185307179586) - log_det_p);
            *out = quad + norm;
            return 0;
        }
        case LOGP_MVN_PREC: {     // params = symmetric precision P [n][n]; logp = -x'Px/2, g = -Px
            Vec t(n);
            for (size_t d = 0; d < n; ++d) {
                double y = 0.0;
                for (size_t j = 0; j < n; ++j) y = std::fma(params[j * n + d], x[j], y);   // column d, j ascending
                g[d] = -y;
                t[d] = x[d] * y;
            }
            *out = -0.5 * m.sum_terms(t.data(), n);
            return 0;
        }
        case LOGP_FUNNEL: {       // v~N(0,3^2), x_i|v ~ N(0, e^v), i=1..n-1
            const double v = x[0];
            const size_t k = n - 1;
            Vec t(n);
            t[0] = 0.0;
            for (size_t i = 1; i < n; ++i) t[i] = x[i] * x[i];
            double ss = m.sum_terms(t.data(), n);
            double ev = m.exp(-v);
            *out = -v * v / 18.0 - 0.5 * (double)k * v - 0.5 * ev * ss;
            g[0] = -v / 9.0 - 0.5 * (double)k + 0.5 * ev * ss;
            for (size_t i = 1; i < n; ++i) g[i] = -ev * x[i];
            return 0;
        }
        case LOGP_EIGHT_SCHOOLS: { // (mu, log tau, theta_tilde[8]); params = y[8], sigma[8]
            const double mu = x[0], lt = x[1];
            const double tau = m.exp(lt);
            Vec t(n);
            double gmu = 0.0, gtau_lin = 0.0;
            t[0] = -mu * mu / 50.0;                                    // mu ~ N(0, 5^2)
            // tau ~ HalfCauchy(5) with log-Jacobian: -ln(1+(tau/5)^2) + lt
            t[1] = lt - m.ln_1p((tau / 5.0) * (tau / 5.0));
            for (size_t i = 0; i < 8; ++i) {
                double th = x[2 + i];
                double r = (params[i] - (mu + tau * th)) / params[8 + i];
                t[2 + i] = -0.5 * th * th - 0.5 * r * r;
                double dr = r / params[8 + i];                          // d(-r^2/2)/d(mu+tau*th)
                gmu += dr;
                gtau_lin += dr * th;
                g[2 + i] = -th + dr * tau;
            }
            g[0] = -mu / 25.0 + gmu;
            g[1] = 1.0 - 2.0 * (tau / 5.0) * (tau / 5.0) / (1.0 + (tau / 5.0) * (tau / 5.0)) + gtau_lin * tau;
            *out = m.sum_terms(t.data(), n);
            return 0;
        }
        case LOGP_HOST_CALLBACK:
            return cb(cb_ctx, n, x, g, out);
        }
        return 2;
    }
};

// ---------------------------------------------------------------------------------------------
// TransformedPoint (reference transformed_hamiltonian.rs:56-77)
// ---------------------------------------------------------------------------------------------
struct Point {
    Vec x, gx, z, gz, v;
    int64_t index_in_trajectory = 0;
    double logp = 0, logdet = 0, kinetic_energy = 0, initial_energy = 0;
    int64_t transform_id = -1;
    explicit Point(size_t n) : x(n, 0.0), gx(n, 0.0), z(n, 0.0), gz(n, 0.0), v(n, 0.0) {}
    double energy() const { return kinetic_energy - (logp + logdet); }          // :349-351
    double energy_error() const { return energy() - initial_energy; }           // hamiltonian.rs:134-136
};
typedef std::shared_ptr<Point> State;

// ---------------------------------------------------------------------------------------------
// DiagMassMatrix (reference src/transform/diagonal.rs)
// ---------------------------------------------------------------------------------------------
struct DiagMassMatrix {
    Vec mean, inv_stds, stds;
    double logdet = 0;
    int64_t id = -1;
    explicit DiagMassMatrix(size_t n) : mean(n, 0.0), inv_stds(n, 0.0), stds(n, 0.0) {}

    // update_diag_draw_grad :107-131 with array_update_var_inv_std_draw_grad cpu_math.rs:671-708
    void update_diag_draw_grad(const Ctx& m, const Vec& draw_mean, const Vec& grad_mean, const Vec& draw_var,
                               const Vec& grad_var, bool has_fill, double fill, double lo, double hi) {
        const size_t n = mean.size();
        for (size_t i = 0; i < n; ++i) {
            double val = std::sqrt(draw_var[i] / grad_var[i]);
            if (!std::isfinite(val) || val == 0.0) {
                if (has_fill) { stds[i] = std::sqrt(fill); inv_stds[i] = std::sqrt(1.0 / fill); }
            } else {
                val = clampd(val, lo, hi);
                stds[i] = std::sqrt(val);
                inv_stds[i] = std::sqrt(1.0 / val);
            }
        }
        Vec var(n);
        multiply(stds.data(), stds.data(), var.data(), n);
        multiply(var.data(), grad_mean.data(), mean.data(), n);
        axpy(draw_mean.data(), mean.data(), 1.0, n);
        logdet = m.sum_ln(inv_stds.data(), n);
        id += 1;
    }
    // update_diag_draw :85-105 with array_update_var_inv_std_draw cpu_math.rs:633-669
    void update_diag_draw(const Ctx& m, const Vec& draw_mean, const Vec& draw_var, double scale, bool has_fill,
                          double fill, double lo, double hi) {
        const size_t n = mean.size();
        for (size_t i = 0; i < n; ++i) {
            double dv = draw_var[i] * scale;
            if (!std::isfinite(dv) || dv == 0.0) {
                if (has_fill) { stds[i] = std::sqrt(fill); inv_stds[i] = std::sqrt(1.0 / fill); }
            } else {
                double val = clampd(dv, lo, hi);
                stds[i] = std::sqrt(val);
                inv_stds[i] = std::sqrt(1.0 / val);
            }
        }
        mean = draw_mean;
        logdet = m.sum_ln(inv_stds.data(), n);
        id += 1;
    }
    // update_diag_grad :133-154 with array_update_var_inv_std_grad cpu_math.rs:710-738
    void update_diag_grad(const Ctx& m, const Vec& position, const Vec& gradient, double fill, double lo, double hi) {
        const size_t n = mean.size();
        for (size_t i = 0; i < n; ++i) {
            double val = 1.0 / clampd(std::fabs(gradient[i]), lo, hi);
            if (!std::isfinite(val)) val = fill;
            stds[i] = std::sqrt(val);
            inv_stds[i] = std::sqrt(1.0 / val);
        }
        Vec var(n);
        multiply(stds.data(), stds.data(), var.data(), n);
        multiply(var.data(), gradient.data(), mean.data(), n);
        axpy(position.data(), mean.data(), 1.0, n);
        logdet = m.sum_ln(inv_stds.data(), n);
        id += 1;
    }
    // set_transform :155-161
    void set_transform(const Ctx& m, const Vec& stds_, const Vec& mean_) {
        stds = stds_; mean = mean_;
        for (size_t i = 0; i < stds.size(); ++i) inv_stds[i] = 1.0 / stds[i];   // array_recip cpu_math.rs:328-330
        logdet = m.sum_ln(inv_stds.data(), stds.size());
        id += 1;
    }
    // :233-265
    void compute_transformed_position(const Vec& x, Vec& z) const {
        axpy_out(mean.data(), x.data(), -1.0, z.data(), x.size());
        for (size_t i = 0; i < x.size(); ++i) z[i] = inv_stds[i] * z[i];
    }
    void compute_untransformed_position(const Vec& z, Vec& x) const {
        multiply(z.data(), stds.data(), x.data(), z.size());
        axpy(mean.data(), x.data(), 1.0, z.size());
    }
    void compute_transformed_gradient(const Vec& gx, Vec& gz) const {
        multiply(gx.data(), stds.data(), gz.data(), gx.size());
    }
};

// ---------------------------------------------------------------------------------------------
// LowRankMassMatrix (reference src/transform/low_rank.rs:95-186, :325-404; apply_lowrank_transform*
// src/math/cpu_math.rs:332-425).  F(y) = sigma . (I + U (diag(lambda)^1/2 - I) U') (y + mu_lr) + mean.
// The base class is the `diag` member; the base's id / logdet double as the OUTER id / logdet (both ids start at -1 and
// move together: update_from_grad and update bump diag.id and self.id in the same call, low_rank.rs:139-186).
// Summation order of the two skinny matrix products: faer's matmul has no reproducible order (the reference's own
// tests use 1e-12); here U'v is one ordered dot product per eigenvector (Ctx::vector_dot: the reference's SIMD order or
// the engine's lane order) and v + U s is accumulated eigenvector by eigenvector with one fma per term.
// ---------------------------------------------------------------------------------------------
struct MassMatrix : DiagMassMatrix {
    bool has_inner = false;
    size_t rank = 0;
    Vec vecs;              // [rank][n]: eigenvector k contiguous (column k of the reference's U)
    Vec vals_sqrt, vals_sqrt_inv, mu_lr;
    double logdet_contribution = 0;
    explicit MassMatrix(size_t n) : DiagMassMatrix(n), mu_lr(n, 0.0) {}

    // update_from_grad low_rank.rs:139-153
    void update_from_grad(const Ctx& m, const Vec& pos, const Vec& grad, double fill, double lo, double hi) {
        has_inner = false; rank = 0;
        update_diag_grad(m, pos, grad, fill, lo, hi);          // logdet = diag.logdet, id += 1
    }
    // update low_rank.rs:155-186 with InnerMatrix::new :55-92.  vecs_rows: [n_eig][n].  Returns false when it bails out.
    bool update(const Ctx& m, const Vec& stds_, const Vec& mean_, const Vec& vals, const Vec& vecs_rows, const Vec& mu_low_rank) {
        const size_t n = mean.size();
        if (!all_finite(stds_.data(), n) || !all_finite(mean_.data(), n)) return false;
        if (!all_finite(vals.data(), vals.size()) || !all_finite(vecs_rows.data(), vecs_rows.size())) return false;
        set_transform(m, stds_, mean_);
        double ld = -0.0;                                       // f64::sum of the mapped iterator: sequential
        for (double v : vals) ld += -0.5 * m.ln(v);
        logdet_contribution = ld;
        rank = vals.size();
        vecs = vecs_rows;
        vals_sqrt.resize(rank); vals_sqrt_inv.resize(rank);
        for (size_t k = 0; k < rank; ++k) { vals_sqrt[k] = std::sqrt(vals[k]); vals_sqrt_inv[k] = 1.0 / vals_sqrt[k]; }
        mu_lr = mu_low_rank;
        logdet = logdet_contribution + logdet;                  // inner.logdet() + diag.logdet()
        has_inner = true;
        return true;
    }
    // apply_lowrank_transform_inplace cpu_math.rs:383-425:  v += U ((vals - 1) . (U' v))
    void apply_inplace(const Ctx& m, const Vec& vals, Vec& v) const {
        if (rank == 0) return;
        const size_t n = v.size();
        Vec sc(rank);
        for (size_t k = 0; k < rank; ++k) sc[k] = m.lowrank_dot(&vecs[k * n], v.data(), n);
        for (size_t k = 0; k < rank; ++k) sc[k] *= vals[k] - 1.0;
        for (size_t k = 0; k < rank; ++k)
            for (size_t i = 0; i < n; ++i) v[i] = std::fma(vecs[k * n + i], sc[k], v[i]);
    }
    // low_rank.rs:325-398
    void compute_transformed_position(const Ctx& m, const Vec& x, Vec& z) const {
        axpy_out(mean.data(), x.data(), -1.0, z.data(), x.size());
        for (size_t i = 0; i < x.size(); ++i) z[i] = z[i] * inv_stds[i];
        if (has_inner) {
            axpy(mu_lr.data(), z.data(), -1.0, z.size());
            apply_inplace(m, vals_sqrt_inv, z);
        }
    }
    void compute_untransformed_position(const Ctx& m, const Vec& z, Vec& x) const {
        if (!has_inner) multiply(z.data(), stds.data(), x.data(), z.size());
        else {
            x = z;                                              // apply_lowrank_transform: dest = rhs, then += U scratch
            apply_inplace(m, vals_sqrt, x);
            axpy(mu_lr.data(), x.data(), 1.0, x.size());
            for (size_t i = 0; i < x.size(); ++i) x[i] = x[i] * stds[i];
        }
        axpy(mean.data(), x.data(), 1.0, z.size());
    }
    void compute_transformed_gradient(const Ctx& m, const Vec& gx, Vec& gz) const {
        multiply(gx.data(), stds.data(), gz.data(), gx.size());
        if (has_inner) apply_inplace(m, vals_sqrt, gz);
    }
};

// What LowRankMassMatrixStrategy::compute_update needs (reference src/transform/adapt/low_rank.rs:73-142): thin SVDs, a
// pivoted QR and three symmetric eigendecompositions.  The oracle delegates them to a callback (numpy / LAPACK in
// oracle/lowrank.py, following the reference step by step).  draws / grads: [ndraws][ndim].  Returns 0 = Some(...), 1 = None.
typedef int (*lowrank_estimator_fn)(void* ctx, uint64_t ndim, uint64_t ndraws, const double* draws, const double* grads,
                                    double gamma, double eigval_cutoff, double* stds, double* mean, uint64_t* n_eig,
                                    double* vals /*[min(ndim, 2 ndraws)]*/, double* vecs /*[.][ndim]*/, double* mu_low_rank);

// AcceptanceRateCollector (reference src/stepsize/dual_avg.rs:83-166)
struct AcceptanceRateCollector {
    double initial_energy = 0;
    double mean_sum = 0, mean_sym_sum = 0;
    uint64_t mean_count = 0, mean_sym_count = 0;
    double max_energy_error = 0;
    void register_init(double e0) { initial_energy = e0; mean_sum = mean_sym_sum = 0; mean_count = mean_sym_count = 0; max_energy_error = 0; }
    void register_leapfrog(const Ctx& m, const Point* end, bool divergent) {
        if (divergent) {
            mean_count++; mean_sym_count++;            // add(0.)
            mean_sum += 0.; mean_sym_sum += 0.;
            max_energy_error = -INFINITY;
        } else {
            double diff = initial_energy - end->energy();
            double e = m.exp(std::fmin(diff, 0.));
            mean_sum += e; mean_count++;
            mean_sym_sum += 2. * e / (1. + m.exp(diff)); mean_sym_count++;
            if (std::fabs(diff) > std::fabs(max_energy_error)) max_energy_error = diff;
        }
    }
    double mean() const { return mean_sum / (double)mean_count; }
    double mean_sym() const { return mean_sym_sum / (double)mean_sym_count; }
};

// DrawGradCollector (reference src/transform/adapt/diagonal.rs:57-84)
struct DrawGradCollector {
    Vec draw, grad;
    bool is_good = true;
    explicit DrawGradCollector(size_t n) : draw(n, 0.0), grad(n, 0.0) {}
    void register_draw(const Point& p, bool diverging) {
        draw = p.x; grad = p.gx;
        int64_t idx = p.index_in_trajectory;
        is_good = diverging ? (std::llabs(idx) > 4) : (idx != 0);
    }
};

struct Collector {   // CombinedCollector (reference src/adapt_strategy.rs:286-350)
    AcceptanceRateCollector acc;
    DrawGradCollector dg;
    explicit Collector(size_t n) : dg(n) {}
};

struct DivergenceInfo {   // dynamics/hamiltonian.rs:20-36 (start_momentum is always None on this path)
    bool present = false; bool has_energy_error = false; double energy_error = 0;
    Vec start_location, start_gradient, end_location;
    bool has_end = false;
    int64_t start_idx = 0, end_idx = 0;
};
enum LeapfrogKind { LF_OK, LF_DIVERGENCE, LF_ERR };
struct LeapfrogResult { LeapfrogKind kind; State state; DivergenceInfo info; };

// ---------------------------------------------------------------------------------------------
// TransformedHamiltonian<DiagMassMatrix | LowRankMassMatrix>; the three KineticEnergyKinds (:27-50)
// ---------------------------------------------------------------------------------------------
struct Hamiltonian {
    const Ctx* m;
    const Density* dens;
    MassMatrix mm;
    double step_size = 0;
    size_t n;
    int64_t kind = TRAJ_EUCLIDEAN;
    double decoherence_length = INFINITY;       // momentum_decoherence_length: Some(L) for MCLMC (:441-443)
    Hamiltonian(const Ctx* m_, const Density* d) : m(m_), dens(d), mm(d->dim), n(d->dim) {}

    // leapfrog :524-615.  `acc` may be null (no collector).
    LeapfrogResult leapfrog(const State& start, int sign, double step_size_factor, double energy_baseline,
                            double max_energy_error, AcceptanceRateCollector* acc) {
        State out = std::make_shared<Point>(n);
        Point& o = *out;
        const Point& s = *start;
        o.initial_energy = s.initial_energy;
        o.transform_id = s.transform_id;
        const double epsilon = (double)sign * step_size * step_size_factor;
        const double sqrt_n = std::sqrt((double)n);
        if (kind == TRAJ_EXACT_NORMAL) {                                         // first_velocity_halfstep :169-177
            m->std_norm_grad_flow(s.z.data(), s.gz.data(), s.v.data(), o.v.data(), epsilon / 2., n);
            m->std_norm_flow(s.z.data(), o.z.data(), o.v.data(), epsilon, n);    // position_step :206-213
        } else if (kind == TRAJ_MICROCANONICAL) {                                // :186-198
            o.v = s.v;
            o.kinetic_energy = s.kinetic_energy + m->esh_momentum_update(s.gz.data(), o.v.data(), sqrt_n * epsilon / 2., n);
            axpy_out(o.v.data(), s.z.data(), epsilon * sqrt_n, o.z.data(), n);   // position_step :214-226
        } else {
            axpy_out(s.gz.data(), s.v.data(), epsilon / 2., o.v.data(), n);      // first_velocity_halfstep :178-184
            axpy_out(o.v.data(), s.z.data(), epsilon, o.z.data(), n);            // position_step :220-225
        }
        // init_from_transformed_position (diagonal.rs:196-209)
        mm.compute_untransformed_position(*m, o.z, o.x);
        double logp = 0;
        int st = dens->logp(*m, o.x.data(), o.gx.data(), &logp);
        if (st != 0) {
            if (st == 2) return {LF_ERR, nullptr, {}};
            DivergenceInfo info; info.present = true;
            info.start_location = s.x; info.start_gradient = s.gx; info.start_idx = s.index_in_trajectory;
            if (acc) acc->register_leapfrog(*m, out.get(), true);
            return {LF_DIVERGENCE, nullptr, info};
        }
        mm.compute_transformed_gradient(*m, o.gx, o.gz);
        o.logp = logp;
        o.logdet = mm.logdet;
        o.transform_id = mm.id;
        if (kind == TRAJ_EXACT_NORMAL)                                           // second_velocity_halfstep :235-258
            m->std_norm_grad_flow(o.z.data(), o.gz.data(), o.v.data(), o.v.data(), epsilon / 2., n);
        else if (kind == TRAJ_MICROCANONICAL)
            o.kinetic_energy = o.kinetic_energy + m->esh_momentum_update(o.gz.data(), o.v.data(), sqrt_n * epsilon / 2., n);
        else
            axpy(o.gz.data(), o.v.data(), epsilon / 2., n);
        if (kind != TRAJ_MICROCANONICAL)
            o.kinetic_energy = 0.5 * m->vector_dot(o.v.data(), o.v.data(), n);  // :260-262
        o.index_in_trajectory = s.index_in_trajectory + sign;
        const double energy_error = o.energy() - energy_baseline;
        const bool bad_energy = kind == TRAJ_MICROCANONICAL ? std::fabs(energy_error) >= max_energy_error
                                                            : energy_error > max_energy_error;   // :583-589
        if (bad_energy | !std::isfinite(energy_error)) {                         // :590-610
            DivergenceInfo info; info.present = true; info.has_energy_error = true; info.energy_error = energy_error;
            info.start_location = s.x; info.start_gradient = s.gx; info.end_location = o.x; info.has_end = true;
            info.start_idx = s.index_in_trajectory; info.end_idx = o.index_in_trajectory;
            if (acc) acc->register_leapfrog(*m, out.get(), true);
            return {LF_DIVERGENCE, nullptr, info};
        }
        if (acc) acc->register_leapfrog(*m, out.get(), false);
        return {LF_OK, out, {}};
    }

    // is_turning :617-638
    bool is_turning(const State& s1, const State& s2) const {
        const Point *start, *end;
        if (s1->index_in_trajectory < s2->index_in_trajectory) { start = s1.get(); end = s2.get(); }
        else { start = s2.get(); end = s1.get(); }
        Vec zeros(n, 0.0);
        double t1, t2;
        m->scalar_prods3(end->z.data(), start->z.data(), zeros.data(), start->v.data(), end->v.data(), n, &t1, &t2);
        return (t1 < 0.) | (t2 < 0.);
    }

    // check_all :310-324
    bool check_all(const Point& p) const {
        return all_finite(p.z.data(), n) && all_finite_and_nonzero(p.gz.data(), n) &&
               all_finite(p.gx.data(), n) && all_finite(p.x.data(), n);
    }

    // init_state :640-661.  returns status
    int init_state(const double* init, State* out) {
        State st = std::make_shared<Point>(n);
        Point& p = *st;
        for (size_t i = 0; i < n; ++i) p.x[i] = init[i];
        double logp = 0;
        int rc = dens->logp(*m, p.x.data(), p.gx.data(), &logp);               // init_from_untransformed_position
        if (rc != 0) return ST_LOGP_FATAL;
        mm.compute_transformed_position(*m, p.x, p.z);
        mm.compute_transformed_gradient(*m, p.gx, p.gz);
        p.logp = logp; p.logdet = mm.logdet; p.transform_id = mm.id;
        if (!check_all(p)) return ST_BAD_INIT;
        *out = st;
        return ST_OK;
    }

    // init_state_untransformed :663-685
    int init_state_untransformed(const double* init, State* out) {
        State st = std::make_shared<Point>(n);
        Point& p = *st;
        for (size_t i = 0; i < n; ++i) p.x[i] = init[i];
        double logp = 0;
        int rc = dens->logp(*m, p.x.data(), p.gx.data(), &logp);
        if (rc != 0) return ST_LOGP_FATAL;
        p.logp = logp;
        p.transform_id = -1;
        if (!(all_finite(p.gx.data(), n) && all_finite(p.x.data(), n))) return ST_BAD_INIT;
        *out = st;
        return ST_OK;
    }

    // initialize_trajectory :687-736 (resample_velocity = true)
    void initialize_trajectory(Point& p, ChaCha8Rng& rng, bool resample_velocity = true) {
        if (resample_velocity) {
            for (size_t i = 0; i < n; ++i) p.v[i] = 1.0 * standard_normal(rng, *m);  // array_gaussian cpu_math.rs:561-577
            if (kind == TRAJ_MICROCANONICAL) m->array_normalize(p.v.data(), n);  // the momentum lives on the unit sphere :700-703
        }
        if (mm.id != p.transform_id) {                                           // inv_transform_normalize diagonal.rs:210-221
            mm.compute_transformed_position(*m, p.x, p.z);
            mm.compute_transformed_gradient(*m, p.gx, p.gz);
            p.logdet = mm.logdet;
            p.transform_id = mm.id;
        }
        if (kind == TRAJ_MICROCANONICAL) p.kinetic_energy = 0.0;                 // accumulated change, none yet :722-727
        else p.kinetic_energy = 0.5 * m->vector_dot(p.v.data(), p.v.data(), n);
        p.index_in_trajectory = 0;
        p.initial_energy = p.energy();
    }

    // partial_momentum_refresh :770-825 (MCLMC only: momentum_decoherence_length is Some(L))
    void partial_momentum_refresh(Point& p, const Vec& noise, double factor) {
        const double half_step = step_size * factor / 2.0;
        if (kind == TRAJ_MICROCANONICAL) {       // isokinetic Langevin: p <- (p + nu z) / |p + nu z|
            const double nn = (double)n;
            const double nu = std::sqrt(m->exp_m1(2.0 * half_step / decoherence_length) / nn);
            axpy(noise.data(), p.v.data(), nu, n);
            m->array_normalize(p.v.data(), n);
        } else {                                 // Ornstein-Uhlenbeck: p <- alpha p + sqrt(1 - alpha^2) z
            const double alpha = m->exp(-half_step / decoherence_length);
            const double beta = std::sqrt(1.0 - alpha * alpha);
            Vec zeros(n, 0.0), nv(n);
            axpy_out(p.v.data(), zeros.data(), alpha, nv.data(), n);
            axpy(noise.data(), nv.data(), beta, n);
            p.v = nv;
            p.kinetic_energy = 0.5 * m->vector_dot(p.v.data(), p.v.data(), n);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// NUTS tree (reference src/nuts.rs)
// ---------------------------------------------------------------------------------------------
struct NutsOptions { uint64_t maxdepth, mindepth; bool check_turning; uint64_t extra_doublings; double max_energy_error;
                     bool has_target_time; double target_time; };
struct SampleInfo { uint64_t depth = 0; DivergenceInfo divergence; bool reached_maxdepth = false; };

struct NutsTree {
    State left, right, draw;
    double log_size = 0;
    uint64_t depth = 0;
    bool is_main = true;
};
enum ExtendKind { EX_OK, EX_ERR, EX_TURNING, EX_DIVERGING };
struct ExtendResult { ExtendKind kind; NutsTree tree; DivergenceInfo info; };

struct TreeBuilder {
    const Ctx* m; Hamiltonian* h; ChaCha8Rng* rng; Collector* coll; int* fatal;

    // single_step :209-245
    LeapfrogResult single_step(const NutsTree& self, int dir, const NutsOptions& opt) {
        const State& start = dir > 0 ? self.right : self.left;
        return h->leapfrog(start, dir, 1.0, start->initial_energy, opt.max_energy_error, &coll->acc);
    }
    // merge_into :172-207
    void merge_into(NutsTree& self, NutsTree& other, int dir) {
        if (dir > 0) self.right = other.right; else self.left = other.left;
        double log_size = m->logaddexp(self.log_size, other.log_size);
        double self_log_size = self.is_main ? self.log_size : log_size;
        bool take = other.log_size >= self_log_size;
        if (!take) {
            int b = random_bool(*rng, m->exp(other.log_size - self_log_size));
            if (b < 0) { *fatal = 1; b = 0; }
            take = b == 1;
        }
        if (take) self.draw = other.draw;
        self.depth += 1;
        self.log_size = log_size;
    }
    // extend :108-170
    ExtendResult extend(NutsTree self, int dir, const NutsOptions& opt) {
        LeapfrogResult lf = single_step(self, dir, opt);
        if (lf.kind == LF_DIVERGENCE) return {EX_DIVERGING, self, lf.info};
        if (lf.kind == LF_ERR) return {EX_ERR, self, {}};
        NutsTree other;
        other.left = other.right = other.draw = lf.state;
        other.depth = 0; other.log_size = -lf.state->energy_error(); other.is_main = false;
        while (other.depth < self.depth) {
            ExtendResult r = extend(other, dir, opt);
            if (r.kind == EX_OK) other = r.tree;
            else if (r.kind == EX_TURNING) return {EX_TURNING, self, {}};
            else if (r.kind == EX_DIVERGING) return {EX_DIVERGING, self, r.info};
            else return {EX_ERR, self, {}};
        }
        const State& first = dir > 0 ? self.left : other.left;
        const State& last = dir > 0 ? other.right : self.right;
        bool turning = false;
        if (opt.check_turning) {
            turning = h->is_turning(first, last);
            if (self.depth > 0) {
                if (!turning) turning = h->is_turning(self.right, other.right);
                if (!turning) turning = h->is_turning(self.left, other.left);
            }
        }
        merge_into(self, other, dir);
        return {turning ? EX_TURNING : EX_OK, self, {}};
    }
};

// nuts::draw :281-388.  returns 0 ok, 2 fatal
static inline int nuts_draw(const Ctx& m, State& init, ChaCha8Rng& rng, Hamiltonian& h, const NutsOptions& options,
                            Collector& coll, State* out_state, SampleInfo* out_info) {
    h.initialize_trajectory(*init, rng);
    coll.acc.register_init(init->energy());
    NutsTree tree;
    tree.left = tree.right = tree.draw = init;
    uint64_t mindepth = options.mindepth, maxdepth = options.maxdepth;
    if (options.has_target_time) {                                                 // :300-320
        // `as u64` saturates (NaN -> 0).  For max_steps == 0 (a NaN step size, e.g. after a draw of zero leapfrogs)
        // the reference's `.log2().floor().to_u64().unwrap()` panics; here and on the device that case counts as
        // max_steps = 1 (depth limits 0), so the chain keeps returning its current point.
        const double q = std::ceil(options.target_time / h.step_size);
        uint64_t max_steps = q >= 18446744073709551616.0 ? ~0ull : (q > 0 ? (uint64_t)q : 0ull);
        if (max_steps == 0) max_steps = 1;
        uint64_t fl = 0;
        while ((max_steps >> (fl + 1)) != 0) ++fl;                                  // floor(log2)
        const uint64_t md = fl;
        mindepth = md > options.mindepth ? md : options.mindepth;
        uint64_t xd = (max_steps & (max_steps - 1)) == 0 ? fl : fl + 1;             // ceil(log2)
        xd = xd > mindepth ? xd : mindepth;
        maxdepth = xd < options.maxdepth ? xd : options.maxdepth;
    }
    int fatal = 0;
    TreeBuilder tb{&m, &h, &rng, &coll, &fatal};
    SampleInfo info;
    auto finish = [&](const NutsTree& t, bool maxd, const DivergenceInfo& div) {
        info.depth = t.depth; info.divergence = div; info.reached_maxdepth = maxd;
        coll.dg.register_draw(*t.draw, div.present);
        *out_state = t.draw; *out_info = info;
        return fatal ? 2 : 0;
    };
    if (h.n == 0) return finish(tree, false, {});
    NutsOptions no_check = options; no_check.check_turning = false;
    while (tree.depth < maxdepth) {
        int dir = rng.random_bool_std() ? 1 : -1;                                  // :334, hamiltonian.rs:111-118
        const NutsOptions& cur = tree.depth < mindepth ? no_check : options;
        ExtendResult r = tb.extend(tree, dir, cur);
        if (fatal) return 2;
        if (r.kind == EX_OK) { tree = r.tree; continue; }
        if (r.kind == EX_TURNING) {
            tree = r.tree;
            for (uint64_t e = 0; e < options.extra_doublings; ++e) {
                ExtendResult r2 = tb.extend(tree, dir, no_check);
                if (fatal) return 2;
                if (r2.kind == EX_OK || r2.kind == EX_TURNING) tree = r2.tree;
                else if (r2.kind == EX_DIVERGING) return finish(r2.tree, false, r2.info);
                else return 2;
            }
            return finish(tree, false, {});
        }
        if (r.kind == EX_DIVERGING) return finish(r.tree, false, r.info);
        return 2;
    }
    return finish(tree, true, {});
}

// ---------------------------------------------------------------------------------------------
// Adaptation
// ---------------------------------------------------------------------------------------------
struct RunningVariance {   // reference src/transform/adapt/diagonal.rs:17-55
    Vec mean, variance;
    uint64_t count = 0;
    explicit RunningVariance(size_t n) : mean(n, 0.0), variance(n, 0.0) {}
    void add_sample(const Vec& value) {
        count += 1;
        if (count == 1) { mean = value; return; }
        const double diff_scale = 1.0 / (double)count;
        for (size_t i = 0; i < mean.size(); ++i) {       // array_update_variance cpu_math.rs:605-631
            double diff = value[i] - mean[i];
            mean[i] += diff * diff_scale;
            variance[i] += diff * diff;
        }
    }
};

struct DualAverage {       // reference src/stepsize/dual_avg.rs:34-81
    double log_step, log_step_adapted, hbar, mu;
    uint64_t count;
    double k, t0, gamma, max_step_size;
    void reset(const Ctx& m, double initial_step) {
        log_step = m.ln(initial_step); log_step_adapted = m.ln(initial_step);
        hbar = 0.; mu = m.ln(10. * initial_step); count = 1;
    }
    void advance(const Ctx& m, double accept_stat, double target) {
        double w = 1. / ((double)count + t0);
        hbar = (1. - w) * hbar + w * (target - accept_stat);
        log_step = mu - hbar * std::sqrt((double)count) / gamma;
        log_step = std::fmin(log_step, m.ln(max_step_size));
        double mk = m.powf((double)count, -k);
        log_step_adapted = mk * log_step + (1. - mk) * log_step_adapted;
        count += 1;
    }
};

// f64::powi: the llvm.powi intrinsic is lowered to compiler-builtins' __powidf2 (square-and-multiply from the low bit)
static inline double powi(double a, int32_t b) {
    uint32_t pw = b < 0 ? 0u - (uint32_t)b : (uint32_t)b;
    double mul = 1.0;
    for (;;) {
        if (pw & 1u) mul *= a;
        pw >>= 1;
        if (pw == 0) break;
        a *= a;
    }
    return b < 0 ? 1.0 / mul : mul;
}

struct Adam {              // reference src/stepsize/adam.rs:42-112
    double log_step = 0, m1 = 0, v = 0;
    uint64_t t = 0;
    double beta1, beta2, epsilon, learning_rate;
    void reset(const Ctx& m, double initial_step) { log_step = m.ln(initial_step); m1 = 0.; v = 0.; t = 0; }
    void advance(double accept_stat, double target) {
        const double gradient = accept_stat - target;
        t += 1;
        m1 = beta1 * m1 + (1.0 - beta1) * gradient;
        v = beta2 * v + (1.0 - beta2) * gradient * gradient;
        const double m_hat = m1 / (1.0 - powi(beta1, (int32_t)t));
        const double v_hat = v / (1.0 - powi(beta2, (int32_t)t));
        log_step += learning_rate * m_hat / (std::sqrt(v_hat) + epsilon);
    }
};

struct Chain {
    Ctx m;
    Density dens;
    Settings s;
    uint64_t chain_id;
    ChaCha8Rng rng;
    Hamiltonian h;
    Collector coll;
    NutsOptions options;
    State state;
    SampleInfo last_info;
    uint64_t mclmc_switch_draw = 0;
    uint64_t draw_count = 0;
    // GlobalStrategy (adapt_strategy.rs:24-39)
    uint64_t num_tune, early_end, final_step_size_window, last_update = 0, current_window_size;
    bool tuning = true, has_initial_mass_matrix = true;
    // diag Strategy (adapt/diagonal.rs:108-115)
    RunningVariance var_draw, var_grad, var_draw_bg, var_grad_bg;
    // LowRankMassMatrixStrategy (adapt/low_rank.rs:14-21): the window of draws / gradients and where its background starts
    std::deque<Vec> lr_draws, lr_grads;
    size_t lr_background_split = 0;
    lowrank_estimator_fn lr_estimator = nullptr;
    void* lr_estimator_ctx = nullptr;
    // stepsize::Strategy (stepsize/adapt.rs:52-65)
    DualAverage da;
    Adam adam;
    double last_mean_tree_accept = 0, last_sym_mean_tree_accept = 0, last_max_energy_error = 0;
    uint64_t last_n_steps = 0;
    int64_t stats_last_id = -1;
    size_t n;

    Chain(const Settings& s_, const Density& d, const MathCfg& cfg, uint64_t chain, const uint8_t key[32])
        : m{cfg}, dens(d), s(s_), chain_id(chain), rng(ChaCha8Rng::from_seed(key)), h(&m, &dens), coll(d.dim),
          var_draw(d.dim), var_grad(d.dim), var_draw_bg(d.dim), var_grad_bg(d.dim), n(d.dim) {
        h.m = &m; h.dens = &dens; h.kind = (int64_t)s.trajectory_kind;
        if (s.sampler == SAMPLER_MCLMC) {                                       // DiagMclmcSettings::new_chain sampler.rs:405-457
            h.kind = s.mclmc_trajectory_kind == MCLMC_MICROCANONICAL ? TRAJ_MICROCANONICAL : TRAJ_EUCLIDEAN;
            h.decoherence_length = s.momentum_decoherence_length;
            mclmc_switch_draw = (uint64_t)(s.trajectory_switch_fraction * (double)s.num_tune);
            s.step_size_method = 2; s.fixed_step_size = s.mclmc_step_size;      // StepSizeAdaptMethod::Fixed(self.step_size)
        }
        options = {s.maxdepth, s.mindepth, s.check_turning != 0, s.extra_doublings, s.max_energy_error,
                   s.has_target_integration_time != 0, s.target_integration_time};
        // GlobalStrategy::new adapt_strategy.rs:77-98
        num_tune = s.num_tune;
        double num_tune_f = (double)num_tune;
        uint64_t step_size_window = (uint64_t)(s.step_size_window * num_tune_f);
        early_end = (uint64_t)(s.early_window * num_tune_f);
        final_step_size_window = num_tune >= step_size_window ? num_tune - step_size_window : 0;
        current_window_size = s.mass_matrix_switch_freq;
        da.k = s.da_k; da.t0 = s.da_t0; da.gamma = s.da_gamma; da.max_step_size = s.da_max_step_size;
        da.reset(m, s.initial_step);                                            // stepsize/adapt.rs:69-72
        adam.beta1 = s.adam_beta1; adam.beta2 = s.adam_beta2; adam.epsilon = s.adam_epsilon;
        adam.learning_rate = s.adam_learning_rate;
        adam.reset(m, s.initial_step);
    }
    Chain(const Chain&) = delete;

    bool is_low_rank() const { return s.adaptation == 1; }
    // MassMatrixAdaptStrategy::{current_count, background_count} (adapt/diagonal.rs:150-159, adapt/low_rank.rs:339-345)
    uint64_t mm_current_count() const { return is_low_rank() ? lr_draws.size() : var_draw.count; }
    uint64_t mm_background_count() const { return is_low_rank() ? lr_draws.size() - lr_background_split : var_draw_bg.count; }
    // LowRankMassMatrixStrategy::adapt -> update (adapt/low_rank.rs:53-71, :347-353): true whenever count >= 3, whatever
    // the estimator or LowRankMassMatrix::update make of the window
    bool lowrank_adapt() {
        if (lr_draws.size() < 3) return false;
        const size_t nd = lr_draws.size();
        Vec D(nd * n), G(nd * n);
        for (size_t i = 0; i < nd; ++i)
            for (size_t j = 0; j < n; ++j) { D[i * n + j] = lr_draws[i][j]; G[i * n + j] = lr_grads[i][j]; }
        const size_t maxr = std::min(n, 2 * nd);
        Vec stds(n), mean(n), vals(maxr), vecs(maxr * n), mu(n);
        uint64_t n_eig = 0;
        if (!lr_estimator) return true;
        int rc = lr_estimator(lr_estimator_ctx, n, nd, D.data(), G.data(), s.lr_gamma, s.lr_eigval_cutoff, stds.data(),
                              mean.data(), &n_eig, vals.data(), vecs.data(), mu.data());
        if (rc != 0) return true;                                // compute_update returned None: no update, still `true`
        vals.resize(n_eig); vecs.resize(n_eig * n);
        (void)h.mm.update(m, stds, mean, vals, vecs, mu);
        return true;
    }
    bool is_fixed() const { return s.step_size_method == 2; }
    bool is_adam() const { return s.step_size_method == 1; }
    void adapt_reset(double step) { if (is_adam()) adam.reset(m, step); else da.reset(m, step); }

    // stepsize::Strategy::init  stepsize/adapt.rs:91-199
    int stepsize_init(const double* position) {
        if (is_fixed()) { h.step_size = s.fixed_step_size; return ST_OK; }
        State st;
        int rc = h.init_state(position, &st);
        if (rc != ST_OK) return rc;
        h.initialize_trajectory(*st, rng);
        AcceptanceRateCollector c;
        c.register_init(st->energy());
        h.step_size = s.initial_step;
        LeapfrogResult r = h.leapfrog(st, +1, 1.0, st->initial_energy, 1000.0, &c);
        if (r.kind == LF_ERR) return ST_OK;          // `let LeapfrogResult::Ok(_) = .. else { return Ok(()) }`
        if (r.kind != LF_OK) return ST_OK;
        double accept_stat = c.mean();
        int dir = accept_stat > s.target_accept ? +1 : -1;
        for (int it = 0; it < 100; ++it) {
            AcceptanceRateCollector c2;
            c2.register_init(st->energy());
            LeapfrogResult r2 = h.leapfrog(st, dir, 1.0, st->initial_energy, 1000.0, &c2);
            if (r2.kind != LF_OK) { h.step_size = s.initial_step; return ST_OK; }
            double a = c2.mean();
            if (dir > 0) {
                if ((a <= s.target_accept) | (h.step_size > 1e5)) { adapt_reset(h.step_size); return ST_OK; }
                h.step_size *= 2.;
            } else {
                if ((a >= s.target_accept) | (h.step_size < 1e-10)) { adapt_reset(h.step_size); return ST_OK; }
                h.step_size /= 2.;
            }
        }
        h.step_size = s.initial_step;
        return ST_OK;
    }

    // update_stepsize stepsize/adapt.rs:235-267
    void update_stepsize(bool use_best_guess) {
        double step = is_fixed() ? s.fixed_step_size
                      : is_adam() ? m.exp(adam.log_step)
                                  : (use_best_guess ? m.exp(da.log_step_adapted) : m.exp(da.log_step));
        if (s.has_jitter) {
            UniformF64 u = UniformF64::make(1.0 - s.jitter, 1.0 + s.jitter);
            double j = u.sample(rng);
            h.step_size = step * j;
        } else h.step_size = step;
    }
    void update_estimator(bool late) {
        if (is_fixed()) return;
        const double accept_stat = late ? last_sym_mean_tree_accept : last_mean_tree_accept;
        if (is_adam()) adam.advance(accept_stat, s.target_accept);
        else da.advance(m, accept_stat, s.target_accept);
    }

    // NutsChain::set_position chain.rs:137-149 -> GlobalStrategy::init adapt_strategy.rs:100-119
    int set_position(const double* position) {
        State st;
        int rc = h.init_state_untransformed(position, &st);
        if (rc != ST_OK) return rc;
        if (is_low_rank()) {
            // LowRankMassMatrixStrategy::init adapt/low_rank.rs:299-317: add_draw(point), update_from_grad
            lr_draws.push_back(st->x); lr_grads.push_back(st->gx);
            h.mm.update_from_grad(m, st->x, st->gx, 1.0, 1e-20, 1e20);
        } else {
            // DiagAdaptStrategy::init adapt/diagonal.rs:209-231
            var_draw.add_sample(st->x); var_draw_bg.add_sample(st->x);
            var_grad.add_sample(st->gx); var_grad_bg.add_sample(st->gx);
            h.mm.update_diag_grad(m, st->x, st->gx, 1.0, 1e-20, 1e20);
        }
        rc = stepsize_init(position);
        if (rc != ST_OK) return rc;
        rc = h.init_state(position, &state);
        if (rc == ST_OK && s.sampler == SAMPLER_MCLMC) h.initialize_trajectory(*state, rng, true);   // MclmcChain::set_position mclmc.rs:472-486
        return rc;
    }

    // GlobalStrategy::adapt adapt_strategy.rs:121-222
    int adapt(uint64_t draw, const State& st) {
        last_mean_tree_accept = coll.acc.mean();                                  // step_size.update, adapt.rs:201-209
        last_sym_mean_tree_accept = coll.acc.mean_sym();
        last_n_steps = coll.acc.mean_count;
        last_max_energy_error = coll.acc.max_energy_error;
        if (draw >= num_tune) { update_stepsize(true); tuning = false; return ST_OK; }
        if (draw < final_step_size_window) {
            bool is_early = draw < early_end;
            const bool frozen = s.freeze_transform != 0;                          // engine knob: no mass-matrix estimator at all
            if (!is_early && draw == early_end)
                current_window_size = std::max(current_window_size, frozen ? (uint64_t)0 : mm_background_count());
            uint64_t switch_freq = is_early ? s.early_mass_matrix_switch_freq : current_window_size;
            if (coll.dg.is_good && !frozen) {
                if (is_low_rank()) {                                              // update_estimators adapt/low_rank.rs:323-333
                    lr_draws.push_back(coll.dg.draw); lr_grads.push_back(coll.dg.grad);
                } else {                                                          // update_estimators adapt/diagonal.rs:134-141
                    var_draw.add_sample(coll.dg.draw); var_grad.add_sample(coll.dg.grad);
                    var_draw_bg.add_sample(coll.dg.draw); var_grad_bg.add_sample(coll.dg.grad);
                }
            }
            bool could_switch = !frozen && mm_background_count() >= switch_freq;
            uint64_t next_window_size;
            if (is_early) next_window_size = s.early_mass_matrix_switch_freq;
            else {
                uint64_t grown = (uint64_t)std::round((double)current_window_size * s.mass_matrix_window_growth);
                next_window_size = std::max(current_window_size + 1, grown);
            }
            bool is_late = next_window_size + draw > final_step_size_window;
            bool force_update = false;
            if (could_switch && !is_late) {
                if (is_low_rank()) {                                              // switch adapt/low_rank.rs:335-342
                    for (size_t i = 0; i < lr_background_split; ++i) { lr_draws.pop_front(); lr_grads.pop_front(); }
                    lr_background_split = lr_draws.size();
                } else {
                    var_draw = var_draw_bg; var_draw_bg = RunningVariance(n);     // switch adapt/diagonal.rs:143-148
                    var_grad = var_grad_bg; var_grad_bg = RunningVariance(n);
                }
                force_update = true;
                if (!is_early) current_window_size = next_window_size;
            }
            bool did_change = false;
            if (frozen) {
            } else if (is_low_rank()) {
                if (force_update | (draw - last_update >= s.mass_matrix_update_freq)) did_change = lowrank_adapt();
            } else if (force_update | (draw - last_update >= s.mass_matrix_update_freq)) {
                if (var_draw.count >= 3) {                                        // Strategy::adapt adapt/diagonal.rs:161-196
                    if (s.use_grad_based_estimate)
                        h.mm.update_diag_draw_grad(m, var_draw.mean, var_grad.mean, var_draw.variance, var_grad.variance,
                                                   false, 0.0, 1e-20, 1e20);
                    else
                        h.mm.update_diag_draw(m, var_draw.mean, var_draw.variance, 1.0 / (double)var_draw.count,
                                              false, 0.0, 1e-20, 1e20);
                    did_change = true;
                }
            }
            if (did_change) last_update = draw;
            update_estimator(is_late);
            if (did_change & has_initial_mass_matrix) {
                has_initial_mass_matrix = false;
                Vec position = st->x;
                return stepsize_init(position.data());
            }
            update_stepsize(false);
            return ST_OK;
        }
        update_estimator(true);
        update_stepsize(draw == num_tune - 1);
        return ST_OK;
    }

    // NutsChain::draw chain.rs:151-188 (+ the stats of expanded_draw :190-232)
    int draw(double* out_position, DrawStats* stats, const DrawVectors* vec = nullptr) {
        if (s.sampler == SAMPLER_MCLMC) return mclmc_draw(out_position, stats, vec);
        State chosen;
        SampleInfo info;
        int rc = nuts_draw(m, state, rng, h, options, coll, &chosen, &info);
        if (rc != 0) { if (stats) stats->chain_status = ST_LOGP_FATAL; return ST_LOGP_FATAL; }
        if (out_position) for (size_t i = 0; i < n; ++i) out_position[i] = chosen->x[i];
        int arc = adapt(draw_count, chosen);
        if (stats) {
            DrawStats& o = *stats;
            o.draw = draw_count; o.chain = chain_id; o.depth = info.depth; o.maxdepth_reached = info.reached_maxdepth;
            o.diverging = info.divergence.present; o.tuning = tuning; o.n_steps = last_n_steps;
            o.index_in_trajectory = chosen->index_in_trajectory; o.transformation_index = chosen->transform_id;
            o.step_size = h.step_size;
            o.step_size_bar = is_fixed() ? s.fixed_step_size : is_adam() ? m.exp(adam.log_step) : m.exp(da.log_step_adapted);
            o.mean_tree_accept = last_mean_tree_accept; o.mean_tree_accept_sym = last_sym_mean_tree_accept;
            o.max_energy_error = last_max_energy_error;
            o.logp = chosen->logp; o.energy = chosen->energy(); o.energy_error = chosen->energy_error();
            o.fisher_distance = m.sq_norm_sum(chosen->z.data(), chosen->gz.data(), n);
            o.divergence_energy_error = (info.divergence.present && info.divergence.has_energy_error)
                                            ? info.divergence.energy_error : NAN;
            o.chain_status = arc;
            // DiagMassMatrix::extract_stats (transform/diagonal.rs:48-70) against the id seen at the previous
            // extraction (chain.rs:195-200; starts at -1, sampler.rs:795)
            o.transformation_update_id = h.mm.id != stats_last_id ? h.mm.id : -1;
            o.num_eigenvalues = (h.mm.id != stats_last_id && h.mm.has_inner) ? h.mm.rank : 0;   // MatrixStats low_rank.rs:222-229
            o.energy_change = NAN; o.average_step_size = NAN;
        }
        if (vec) {
            auto put = [&](double* dst, const Vec& v) { if (dst) for (size_t i = 0; i < n; ++i) dst[i] = v[i]; };
            put(vec->gradient, chosen->gx);
            put(vec->transformed_position, chosen->z);
            put(vec->transformed_gradient, chosen->gz);
            if (h.mm.id != stats_last_id) {
                put(vec->mass_matrix_inv, h.mm.stds); put(vec->transformation_mu, h.mm.mean);
                if (vec->mass_matrix_eigvals && h.mm.has_inner)                    // low_rank.rs:232-243: lambda^(1/2), NaN beyond the rank
                    for (size_t i = 0; i < n; ++i) vec->mass_matrix_eigvals[i] = i < h.mm.rank ? h.mm.vals_sqrt[i] : NAN;
            }
            if (info.divergence.present) {
                put(vec->divergence_start, info.divergence.start_location);
                put(vec->divergence_start_gradient, info.divergence.start_gradient);
                if (info.divergence.has_end) put(vec->divergence_end, info.divergence.end_location);
            }
        }
        stats_last_id = h.mm.id;
        draw_count += 1;
        state = chosen;
        last_info = info;
        return arc;
    }

    // MclmcChain::draw + mclmc_kernel (reference src/mclmc.rs:212-409, :488-556): `num_steps` ESH / Euclidean leapfrogs with a
    // partial momentum refresh on both sides of each, halving the step size factor on a divergence (dynamic_step_size)
    int mclmc_draw(double* out_position, DrawStats* stats, const DrawVectors* vec) {
        bool resample_velocity = false;                                          // Euclidean -> Microcanonical switch :490-504
        if (s.mclmc_trajectory_kind == MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL && draw_count == mclmc_switch_draw &&
            h.kind != TRAJ_MICROCANONICAL) {
            h.kind = TRAJ_MICROCANONICAL;
            resample_velocity = true;
        }
        const double base_step_size = h.step_size;
        double ns = std::round(s.subsample_frequency * h.decoherence_length / base_step_size);   // f64::round: half away from zero
        ns = ns != ns ? 1.0 : std::fmax(ns, 1.0);                                // f64::max ignores a NaN
        ns = std::fmin(ns, 1e6);
        const uint64_t num_base_steps = (uint64_t)ns;
        const uint64_t max_halvings = s.dynamic_step_size ? 10 : 0;
        State current = std::make_shared<Point>(*state);
        h.initialize_trajectory(*current, rng, resample_velocity);
        Vec noise(n);
        auto gaussian = [&]() { for (size_t i = 0; i < n; ++i) noise[i] = 1.0 * standard_normal(rng, m); };
        gaussian();
        const double draw_start_energy = current->energy();
        coll.acc = AcceptanceRateCollector();                                    // adapt.new_collector: never register_init'ed (base energy 0)
        bool diverged = false;
        DivergenceInfo div;
        uint64_t steps_taken = 0;
        double factor = 1.0;
        std::vector<uint64_t> remaining_stack;
        uint64_t remaining = num_base_steps;
        double time = 0.0;
        while (remaining > 0) {
            Vec tmp_velocity = current->v;
            h.partial_momentum_refresh(*current, noise, factor);
            const double step_baseline = current->energy();
            LeapfrogResult r = h.leapfrog(current, +1, factor, step_baseline,
                                          s.max_energy_error * factor / (double)num_base_steps, &coll.acc);
            if (r.kind == LF_OK) {
                gaussian();
                h.partial_momentum_refresh(*r.state, noise, factor);
                gaussian();
                current = r.state;
                steps_taken += 1;
                remaining -= 1;
                time += factor * base_step_size;
                while (remaining == 0) {
                    if (remaining_stack.empty()) break;
                    remaining = remaining_stack.back() - 1;
                    remaining_stack.pop_back();
                    factor *= 2.0;
                }
            } else if (r.kind == LF_DIVERGENCE) {
                if (remaining_stack.size() >= max_halvings) { diverged = true; div = r.info; break; }
                factor *= 0.5;
                remaining_stack.push_back(remaining);
                remaining = 2;
                current->v = tmp_velocity;
            } else {
                if (stats) stats->chain_status = ST_LOGP_FATAL;
                return ST_LOGP_FATAL;
            }
        }
        State next;
        double energy_change;
        if (diverged) {      // stay at the pre-trajectory position with a fresh momentum :361-388
            next = std::make_shared<Point>(*state);
            h.initialize_trajectory(*next, rng, true);
            energy_change = current->energy() - draw_start_energy;
        } else {
            next = current;
            energy_change = current->energy_error();
        }
        coll.dg.register_draw(*current, diverged);
        const double average_step_size = time / (double)steps_taken;
        if (out_position) for (size_t i = 0; i < n; ++i) out_position[i] = next->x[i];
        int arc = adapt(draw_count, next);
        if (stats) {
            DrawStats& o = *stats;
            o.draw = draw_count; o.chain = chain_id; o.depth = steps_taken; o.maxdepth_reached = 0;
            o.diverging = diverged; o.tuning = tuning; o.n_steps = last_n_steps;
            o.index_in_trajectory = next->index_in_trajectory; o.transformation_index = next->transform_id;
            o.step_size = h.step_size; o.step_size_bar = s.fixed_step_size;
            o.mean_tree_accept = last_mean_tree_accept; o.mean_tree_accept_sym = last_sym_mean_tree_accept;
            o.max_energy_error = last_max_energy_error;
            o.logp = next->logp; o.energy = next->energy(); o.energy_error = next->energy_error();
            o.fisher_distance = m.sq_norm_sum(next->z.data(), next->gz.data(), n);
            o.divergence_energy_error = (diverged && div.has_energy_error) ? div.energy_error : NAN;
            o.chain_status = arc;
            o.transformation_update_id = h.mm.id != stats_last_id ? h.mm.id : -1;
            o.num_eigenvalues = (h.mm.id != stats_last_id && h.mm.has_inner) ? h.mm.rank : 0;   // MatrixStats low_rank.rs:222-229
            o.energy_change = energy_change; o.average_step_size = average_step_size;
        }
        if (vec) {
            auto put = [&](double* dst, const Vec& v) { if (dst) for (size_t i = 0; i < n; ++i) dst[i] = v[i]; };
            put(vec->gradient, next->gx);
            put(vec->transformed_position, next->z);
            put(vec->transformed_gradient, next->gz);
            if (h.mm.id != stats_last_id) {
                put(vec->mass_matrix_inv, h.mm.stds); put(vec->transformation_mu, h.mm.mean);
                if (vec->mass_matrix_eigvals && h.mm.has_inner)
                    for (size_t i = 0; i < n; ++i) vec->mass_matrix_eigvals[i] = i < h.mm.rank ? h.mm.vals_sqrt[i] : NAN;
            }
            if (diverged) {
                put(vec->divergence_start, div.start_location);
                put(vec->divergence_start_gradient, div.start_gradient);
                if (div.has_end) put(vec->divergence_end, div.end_location);
            }
        }
        stats_last_id = h.mm.id;
        draw_count += 1;
        state = next;
        return arc;
    }
};

// Chain RNG key: Sampler seeding (reference src/sampler.rs:1105-1106 then :761)
static inline ChaCha8Rng outer_rng(uint64_t seed, uint64_t chain_id) {
    ChaCha8Rng r = ChaCha8Rng::seed_from_u64(seed);
    r.set_stream(chain_id + 1);
    return r;
}

}  // namespace nmo
