// nuts_oracle.cpp — C API (ctypes) of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see nmo_math.hpp).
// Built by oracle/Makefile into oracle/libnuts_oracle.so with -O2 -ffp-contract=off (the reference's Rust
// never contracts a*b+c; FMAs appear only where src/math/util.rs writes mul_add).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include "nmo_nuts.hpp"

using namespace nmo;

extern "C" {

// DiagNutsSettings::default() (reference src/sampler.rs:507-531, :630-634; src/adapt_strategy.rs:56-69;
// src/stepsize/adapt.rs:320-329; src/stepsize/dual_avg.rs:22-31; src/transform/adapt/diagonal.rs:99-106)
void nmo_settings_default(Settings* s) {
    std::memset(s, 0, sizeof(*s));
    s->num_tune = 400; s->num_draws = 1000; s->maxdepth = 10; s->mindepth = 0;
    s->max_energy_error = 1000.0; s->check_turning = 1; s->extra_doublings = 0; s->seed = 0; s->num_chains = 6;
    s->early_window = 0.3; s->step_size_window = 0.15;
    s->mass_matrix_switch_freq = 80; s->early_mass_matrix_switch_freq = 10; s->mass_matrix_update_freq = 1;
    s->mass_matrix_window_growth = 1.5;
    s->store_mass_matrix = 0; s->use_grad_based_estimate = 1;
    s->target_accept = 0.8; s->initial_step = 0.1; s->has_jitter = 1; s->jitter = 0.1;
    s->step_size_method = 0; s->fixed_step_size = 0.0;
    s->da_k = 0.75; s->da_t0 = 10.; s->da_gamma = 0.05; s->da_max_step_size = 3.14159265358979323846;
    s->adam_beta1 = 0.9; s->adam_beta2 = 0.999; s->adam_epsilon = 1e-8; s->adam_learning_rate = 0.05;   // adam.rs:25-33
    s->adaptation = 0; s->lr_gamma = 1e-5; s->lr_eigval_cutoff = 2.0; s->freeze_transform = 0;          // low_rank.rs:195-203
    s->trajectory_kind = TRAJ_EUCLIDEAN;                                                                  // sampler.rs:528
    // default_mclmc_settings (sampler.rs:342-366), inert while sampler == SAMPLER_NUTS
    s->sampler = SAMPLER_NUTS; s->mclmc_step_size = 0.5; s->momentum_decoherence_length = 3.0; s->subsample_frequency = 1.0;
    s->dynamic_step_size = 1; s->mclmc_trajectory_kind = MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL; s->trajectory_switch_fraction = 0.3;
}
// DiagMclmcSettings::default() (reference src/sampler.rs:368-374): num_tune 400, 6 chains, max_energy_error 1000, Fixed(0.5)
void nmo_settings_default_mclmc(Settings* s) {
    nmo_settings_default(s);
    s->sampler = SAMPLER_MCLMC; s->step_size_method = 2; s->fixed_step_size = 0.5;
}
// LowRankNutsSettings::default() (reference src/sampler.rs:636-642): num_tune 800, mass_matrix_update_freq 20
void nmo_settings_default_low_rank(Settings* s) {
    nmo_settings_default(s);
    s->num_tune = 800; s->mass_matrix_update_freq = 20; s->adaptation = 1;
}
uint64_t nmo_settings_size(void) { return sizeof(Settings); }
uint64_t nmo_draw_stats_size(void) { return sizeof(DrawStats); }

static Density make_density(int64_t kind, uint64_t dim, const double* params, uint64_t n_params) {
    Density d;
    d.kind = kind; d.dim = dim;
    d.params.assign(params, params + n_params);
    return d;
}

void nmo_chain_key(uint64_t seed, uint64_t chain_id, uint8_t key_out[32]) {
    ChaCha8Rng outer = outer_rng(seed, chain_id);
    ChaCha8Rng c = outer.fork();
    for (int i = 0; i < 8; ++i) {
        key_out[4 * i] = (uint8_t)c.key[i]; key_out[4 * i + 1] = (uint8_t)(c.key[i] >> 8);
        key_out[4 * i + 2] = (uint8_t)(c.key[i] >> 16); key_out[4 * i + 3] = (uint8_t)(c.key[i] >> 24);
    }
}

// CpuMath::init_position (reference src/math/cpu_math.rs:171-199) from the outer generator, after the fork
void nmo_init_position_uniform(uint64_t seed, uint64_t chain_id, uint64_t dim, double* out) {
    ChaCha8Rng outer = outer_rng(seed, chain_id);
    (void)outer.fork();
    for (uint64_t i = 0; i < dim; ++i) out[i] = outer.random_f64() * 2.0 - 1.0;
}

void* nmo_chain_create(const Settings* s, int64_t kind, uint64_t dim, const double* params, uint64_t n_params,
                       const MathCfg* cfg, uint64_t chain_id, const uint8_t key[32]) {
    Density d = make_density(kind, dim, params, n_params);
    return new Chain(*s, d, *cfg, chain_id, key);
}
void* nmo_chain_create_callback(const Settings* s, uint64_t dim, host_logp_fn cb, void* cb_ctx, const MathCfg* cfg,
                                uint64_t chain_id, const uint8_t key[32]) {
    Density d;
    d.kind = LOGP_HOST_CALLBACK; d.dim = dim; d.cb = cb; d.cb_ctx = cb_ctx;
    return new Chain(*s, d, *cfg, chain_id, key);
}
void nmo_chain_destroy(void* c) { delete (Chain*)c; }
// LowRankMassMatrixStrategy's dense linear algebra (thin SVD / pivoted QR / eigh) is delegated to this callback
void nmo_chain_set_estimator(void* c, lowrank_estimator_fn fn, void* ctx) { ((Chain*)c)->lr_estimator = fn; ((Chain*)c)->lr_estimator_ctx = ctx; }
// LowRankMassMatrix::update(stds, mean, vals, vecs, mean_low_rank) on the chain's transformation (vecs: [n_eig][dim]);
// returns 1 if it was applied, 0 if update() bailed out on non-finite input
int nmo_chain_set_transform(void* cv, const double* stds, const double* mean, uint64_t n_eig, const double* vals,
                            const double* vecs, const double* mu_lr) {
    Chain* c = (Chain*)cv;
    const size_t n = c->n;
    const int ok = c->h.mm.update(c->m, Vec(stds, stds + n), Vec(mean, mean + n), Vec(vals, vals + n_eig), Vec(vecs, vecs + n_eig * n),
                                  Vec(mu_lr, mu_lr + n)) ? 1 : 0;
    if (c->m.cfg.lr_seq_dots == 2) c->m.cfg.tile_order = 1;     // the engine's lockstep kernel takes over from here (set_position ran on the wave kernels)
    return ok;
}
int nmo_chain_draw_ex(void* c, double* out_position, DrawStats* stats, const DrawVectors* vec) { return ((Chain*)c)->draw(out_position, stats, vec); }
int nmo_chain_set_position(void* c, const double* x0) { return ((Chain*)c)->set_position(x0); }
int nmo_chain_draw(void* c, double* out_position, DrawStats* stats) { return ((Chain*)c)->draw(out_position, stats); }
void nmo_chain_get_state(void* cv, double* x, double* gx, double* stds, double* mean, double* step_size,
                         uint64_t* rng_pos) {
    Chain* c = (Chain*)cv;
    for (size_t i = 0; i < c->n; ++i) {
        if (x) x[i] = c->state->x[i];
        if (gx) gx[i] = c->state->gx[i];
        if (stds) stds[i] = c->h.mm.stds[i];
        if (mean) mean[i] = c->h.mm.mean[i];
    }
    if (step_size) *step_size = c->h.step_size;
    if (rng_pos) *rng_pos = c->rng.pos;
}

// Many chains, one task per chain over `n_threads` host threads (the structure of the reference's Rayon
// Sampler, src/sampler.rs:1116, :1287-1326).  x0 [n][dim]; out_positions [n_draws][n][dim] or NULL;
// out_stats [n_draws][n] or NULL.  Returns the number of chains that failed.
// A transformation given from outside (nm_engine_set_transform on the engine): applied to every chain right after
// set_position with LowRankMassMatrix::update semantics.  per_chain = 0: one (stds, mean, vals, vecs, mu_lr) for all
// chains; 1: arrays are [n_chains][...].
struct RunExtras {
    const double *stds, *mean, *vals, *vecs, *mu_lr;
    uint64_t n_eig, per_chain;
    lowrank_estimator_fn estimator;
    void* estimator_ctx;
};
static RunExtras g_no_extras = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr};
int nmo_run_ex(const Settings* s, int64_t kind, uint64_t dim, const double* params, uint64_t n_params,
               const MathCfg* cfg, uint64_t n_chains, uint64_t chain_offset, const double* x0, uint64_t n_draws,
               double* out_positions, DrawStats* out_stats, uint64_t* out_total_steps, uint64_t n_threads,
               const DrawVectors* out_vec, const RunExtras* ex);
int nmo_run(const Settings* s, int64_t kind, uint64_t dim, const double* params, uint64_t n_params,
            const MathCfg* cfg, uint64_t n_chains, uint64_t chain_offset, const double* x0, uint64_t n_draws,
            double* out_positions, DrawStats* out_stats, uint64_t* out_total_steps, uint64_t n_threads,
            const DrawVectors* out_vec /* bases of [n_draws][n][dim] arrays, or NULL */) {
    return nmo_run_ex(s, kind, dim, params, n_params, cfg, n_chains, chain_offset, x0, n_draws, out_positions, out_stats,
                      out_total_steps, n_threads, out_vec, &g_no_extras);
}
int nmo_run_ex(const Settings* s, int64_t kind, uint64_t dim, const double* params, uint64_t n_params,
               const MathCfg* cfg, uint64_t n_chains, uint64_t chain_offset, const double* x0, uint64_t n_draws,
               double* out_positions, DrawStats* out_stats, uint64_t* out_total_steps, uint64_t n_threads,
               const DrawVectors* out_vec, const RunExtras* ex) {
    std::atomic<uint64_t> next{0}, steps{0};
    std::atomic<int> failed{0};
    auto work = [&]() {
        for (;;) {
            uint64_t c = next.fetch_add(1);
            if (c >= n_chains) break;
            uint8_t key[32];
            nmo_chain_key(s->seed, chain_offset + c, key);
            Density d = make_density(kind, dim, params, n_params);
            Chain ch(*s, d, *cfg, chain_offset + c, key);
            ch.lr_estimator = ex->estimator; ch.lr_estimator_ctx = ex->estimator_ctx;
            if (ch.set_position(x0 + c * dim) != ST_OK) { failed++; continue; }
            if (ex->stds) {
                const uint64_t k = ex->per_chain ? c : 0, r = ex->n_eig;
                (void)nmo_chain_set_transform(&ch, ex->stds + k * dim, ex->mean + k * dim, r, ex->vals + k * r,
                                              ex->vecs + k * r * dim, ex->mu_lr + k * dim);
            }
            uint64_t local = 0;
            for (uint64_t t = 0; t < n_draws; ++t) {
                DrawStats st;
                DrawVectors row;
                if (out_vec) {
                    const size_t off = (t * n_chains + c) * dim;
                    auto at = [&](double* b) { return b ? b + off : nullptr; };
                    row = {at(out_vec->gradient), at(out_vec->transformed_position), at(out_vec->transformed_gradient),
                           at(out_vec->mass_matrix_inv), at(out_vec->transformation_mu), at(out_vec->divergence_start),
                           at(out_vec->divergence_start_gradient), at(out_vec->divergence_end), at(out_vec->mass_matrix_eigvals)};
                }
                int rc = ch.draw(out_positions ? out_positions + (t * n_chains + c) * dim : nullptr, &st,
                                 out_vec ? &row : nullptr);
                if (out_stats) out_stats[t * n_chains + c] = st;
                local += st.n_steps;
                if (rc != ST_OK) { failed++; break; }
            }
            steps += local;
        }
    };
    if (n_threads <= 1) work();
    else {
        std::vector<std::thread> th;
        for (uint64_t i = 0; i < n_threads; ++i) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    if (out_total_steps) *out_total_steps = steps.load();
    return failed.load();
}

// Same as nmo_run but separates the first `n_warm` draws of every chain (warm-up) from the rest and reports the
// CPU seconds and leapfrog steps of the post-warm-up part summed over chains (bench.py's cpu_baseline leg).
int nmo_run_timed(const Settings* s, int64_t kind, uint64_t dim, const double* params, uint64_t n_params,
                  const MathCfg* cfg, uint64_t n_chains, uint64_t chain_offset, const double* x0, uint64_t n_warm,
                  uint64_t n_draws, uint64_t n_threads, double* out_warm_cpu_seconds, uint64_t* out_warm_steps,
                  double* out_cpu_seconds, uint64_t* out_steps) {
    std::atomic<uint64_t> next{0}, steps{0}, wsteps{0};
    std::atomic<int> failed{0};
    std::vector<double> secs(n_threads ? n_threads : 1, 0.0), wsecs(n_threads ? n_threads : 1, 0.0);
    auto work = [&](uint64_t tid) {
        for (;;) {
            uint64_t c = next.fetch_add(1);
            if (c >= n_chains) break;
            uint8_t key[32];
            nmo_chain_key(s->seed, chain_offset + c, key);
            Density d = make_density(kind, dim, params, n_params);
            Chain ch(*s, d, *cfg, chain_offset + c, key);
            if (ch.set_position(x0 + c * dim) != ST_OK) { failed++; continue; }
            DrawStats st;
            auto t0 = std::chrono::steady_clock::now();
            uint64_t w = 0;
            for (uint64_t t = 0; t < n_warm; ++t) { if (ch.draw(nullptr, &st) != ST_OK) { failed++; break; } w += st.n_steps; }
            auto t1 = std::chrono::steady_clock::now();
            uint64_t local = 0;
            for (uint64_t t = 0; t < n_draws; ++t) { if (ch.draw(nullptr, &st) != ST_OK) { failed++; break; } local += st.n_steps; }
            auto t2 = std::chrono::steady_clock::now();
            wsecs[tid] += std::chrono::duration<double>(t1 - t0).count();
            secs[tid] += std::chrono::duration<double>(t2 - t1).count();
            steps += local; wsteps += w;
        }
    };
    if (n_threads <= 1) work(0);
    else {
        std::vector<std::thread> th;
        for (uint64_t i = 0; i < n_threads; ++i) th.emplace_back(work, i);
        for (auto& t : th) t.join();
    }
    double tot = 0, wtot = 0;
    for (double v : secs) tot += v;
    for (double v : wsecs) wtot += v;
    if (out_cpu_seconds) *out_cpu_seconds = tot;
    if (out_warm_cpu_seconds) *out_warm_cpu_seconds = wtot;
    if (out_steps) *out_steps = steps.load();
    if (out_warm_steps) *out_warm_steps = wsteps.load();
    return failed.load();
}

// Wall-clock variant (bench.py's cpu_baseline): phase 1 = every chain's set_position + n_warm draws on n_threads host
// threads (one chain per task, the reference's Rayon structure), join; phase 2 = n_draws draws of every chain, again
// one chain per task.  Reports the wall seconds and leapfrog steps of each phase.
int nmo_run_wall(const Settings* s, int64_t kind, uint64_t dim, const double* params, uint64_t n_params,
                 const MathCfg* cfg, uint64_t n_chains, uint64_t chain_offset, const double* x0, uint64_t n_warm,
                 uint64_t n_draws, uint64_t n_threads, double* out_warm_wall, uint64_t* out_warm_steps,
                 double* out_wall, uint64_t* out_steps) {
    std::vector<std::unique_ptr<Chain>> chains(n_chains);
    std::atomic<int> failed{0};
    auto phase = [&](bool warm, double* wall, uint64_t* steps_out) {
        std::atomic<uint64_t> next{0}, steps{0};
        auto work = [&]() {
            for (;;) {
                uint64_t c = next.fetch_add(1);
                if (c >= n_chains) break;
                DrawStats st;
                uint64_t local = 0;
                if (warm) {
                    uint8_t key[32];
                    nmo_chain_key(s->seed, chain_offset + c, key);
                    Density d = make_density(kind, dim, params, n_params);
                    chains[c].reset(new Chain(*s, d, *cfg, chain_offset + c, key));
                    if (chains[c]->set_position(x0 + c * dim) != ST_OK) { failed++; chains[c].reset(); continue; }
                }
                if (!chains[c]) continue;
                const uint64_t n = warm ? n_warm : n_draws;
                for (uint64_t t = 0; t < n; ++t) {
                    if (chains[c]->draw(nullptr, &st) != ST_OK) { failed++; chains[c].reset(); break; }
                    local += st.n_steps;
                }
                steps += local;
            }
        };
        auto t0 = std::chrono::steady_clock::now();
        if (n_threads <= 1) work();
        else {
            std::vector<std::thread> th;
            for (uint64_t i = 0; i < n_threads; ++i) th.emplace_back(work);
            for (auto& t : th) t.join();
        }
        auto t1 = std::chrono::steady_clock::now();
        if (wall) *wall = std::chrono::duration<double>(t1 - t0).count();
        if (steps_out) *steps_out = steps.load();
    };
    phase(true, out_warm_wall, out_warm_steps);
    phase(false, out_wall, out_steps);
    return failed.load();
}

// Adam::new(initial_step) then advance(accept[i], target): log_step after every advance
void nmo_adam_sequence(const MathCfg* cfg, double initial_step, const double* accept, uint64_t n, double target,
                       double beta1, double beta2, double epsilon, double learning_rate, double* out_log_step) {
    Ctx m{*cfg};
    Adam a;
    a.beta1 = beta1; a.beta2 = beta2; a.epsilon = epsilon; a.learning_rate = learning_rate;
    a.reset(m, initial_step);
    for (uint64_t i = 0; i < n; ++i) { a.advance(accept[i], target); out_log_step[i] = a.log_step; }
}

// ----- primitives for known-answer tests ------------------------------------------------------
double nmo_logaddexp(const MathCfg* cfg, double a, double b) { Ctx m{*cfg}; return m.logaddexp(a, b); }
double nmo_scalar_fn(const MathCfg* cfg, int64_t op, double a, double b) {
    Ctx m{*cfg};
    switch (op) {
    case 0: return m.exp(a);
    case 1: return m.ln(a);
    case 2: return m.ln_1p(a);
    case 3: return m.logaddexp(a, b);
    case 4: return std::sqrt(a);
    case 5: return a / b;
    case 6: return m.powf(a, b);
    case 7: return m.exp_m1(a);
    case 8: return m.sin(a);
    case 9: return m.cos(a);
    }
    return NAN;
}
void nmo_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int64_t rounds, uint32_t out[16]) {
    chacha_block(key, counter, stream, (int)rounds, out);
}
void nmo_rng_words(const uint8_t key[32], uint64_t stream, uint64_t count, uint32_t* out) {
    ChaCha8Rng r = ChaCha8Rng::from_seed(key);
    r.set_stream(stream);
    for (uint64_t i = 0; i < count; ++i) out[i] = r.next_u32();
}
void nmo_seed_from_u64(uint64_t state, uint8_t key_out[32]) {
    ChaCha8Rng r = ChaCha8Rng::seed_from_u64(state);
    for (int i = 0; i < 8; ++i) for (int b = 0; b < 4; ++b) key_out[4 * i + b] = (uint8_t)(r.key[i] >> (8 * b));
}
uint64_t nmo_standard_normal_stream(const MathCfg* cfg, const uint8_t key[32], uint64_t count, double* out) {
    Ctx m{*cfg};
    ChaCha8Rng r = ChaCha8Rng::from_seed(key);
    for (uint64_t i = 0; i < count; ++i) out[i] = standard_normal(r, m);
    return r.pos;
}
void nmo_zig_tables(double* x257, double* f257) {
    const ZigguratTables& t = zig_tables();
    for (int i = 0; i < 257; ++i) { x257[i] = t.x[i]; f257[i] = t.f[i]; }
}
// distribution samplers on a fresh ChaCha8(key): kind 0 bool, 1 f64, 2 random_bool(p=a), 3 Uniform(a,b)
void nmo_rng_samples(const uint8_t key[32], int64_t kind, double a, double b, uint64_t count, double* out) {
    ChaCha8Rng r = ChaCha8Rng::from_seed(key);
    UniformF64 u = UniformF64::make(kind == 3 ? a : 0.0, kind == 3 ? b : 1.0);
    for (uint64_t i = 0; i < count; ++i) {
        switch (kind) {
        case 0: out[i] = r.random_bool_std() ? 1.0 : 0.0; break;
        case 1: out[i] = r.random_f64(); break;
        case 2: out[i] = (double)random_bool(r, a); break;
        case 3: out[i] = u.sample(r); break;
        }
    }
}
double nmo_vector_dot(const MathCfg* cfg, const double* a, const double* b, uint64_t n) { Ctx m{*cfg}; return m.vector_dot(a, b, n); }
void nmo_scalar_prods3(const MathCfg* cfg, const double* p1, const double* n1, const double* p2, const double* x,
                       const double* y, uint64_t n, double* out2) {
    Ctx m{*cfg};
    m.scalar_prods3(p1, n1, p2, x, y, n, &out2[0], &out2[1]);
}
void nmo_axpy(const double* x, double* y, double a, uint64_t n) { axpy(x, y, a, n); }
void nmo_axpy_out(const double* x, const double* y, double a, double* out, uint64_t n) { axpy_out(x, y, a, out, n); }
void nmo_multiply(const double* x, const double* y, double* out, uint64_t n) { multiply(x, y, out, n); }
int nmo_logp(const MathCfg* cfg, int64_t kind, uint64_t dim, const double* params, uint64_t n_params, const double* x,
             double* g, double* out) {
    Ctx m{*cfg};
    Density d = make_density(kind, dim, params, n_params);
    return d.logp(m, x, g, out);
}

// DiagMassMatrix known-answer harness (reference tests src/transform/mod.rs:175-377):
// update_diag_draw_grad(draw_mean, grad_mean, draw_var, grad_var, None, (1e-20,1e20)), then
// init_from_untransformed_position(x) and init_from_transformed_position(z) round trip.
int nmo_diag_kat(const MathCfg* cfg, uint64_t dim, const double* precision_diag, const double* draw_mean,
                 const double* grad_mean, const double* draw_var, const double* grad_var, const double* x,
                 double* z, double* gz, double* logp, double* logdet, double* x_rt, double* logp_rt, double* logdet_rt,
                 double* stds, double* inv_stds, double* mean) {
    Ctx m{*cfg};
    Density d = make_density(LOGP_DIAG_NORMAL, dim, precision_diag, dim);
    DiagMassMatrix mm(dim);
    Vec dm(draw_mean, draw_mean + dim), gm(grad_mean, grad_mean + dim), dv(draw_var, draw_var + dim), gv(grad_var, grad_var + dim);
    mm.update_diag_draw_grad(m, dm, gm, dv, gv, false, 0.0, 1e-20, 1e20);
    Vec X(x, x + dim), GX(dim), Z(dim), GZ(dim);
    int rc = d.logp(m, X.data(), GX.data(), logp);
    mm.compute_transformed_position(X, Z);
    mm.compute_transformed_gradient(GX, GZ);
    *logdet = mm.logdet;
    Vec XR(dim), GXR(dim), GZR(dim);
    mm.compute_untransformed_position(Z, XR);
    rc |= d.logp(m, XR.data(), GXR.data(), logp_rt);
    mm.compute_transformed_gradient(GXR, GZR);
    *logdet_rt = mm.logdet;
    for (uint64_t i = 0; i < dim; ++i) {
        z[i] = Z[i]; gz[i] = GZ[i]; x_rt[i] = XR[i];
        stds[i] = mm.stds[i]; inv_stds[i] = mm.inv_stds[i]; mean[i] = mm.mean[i];
    }
    return rc;
}

// LowRankMassMatrix known-answer harness (reference tests src/transform/mod.rs:391-674, src/transform/low_rank.rs:415-533):
// update(stds, mean, vals, vecs, mu_lr) then init_from_untransformed_position(x) and the init_from_transformed_position
// round trip; `which` >= 0 instead applies one map to `x`: 0 compute_transformed_position, 1 compute_untransformed_position,
// 2 compute_transformed_gradient (result in z).
int nmo_lowrank_kat(const MathCfg* cfg, uint64_t dim, const double* precision_diag, const double* stds, const double* mean,
                    uint64_t n_eig, const double* vals, const double* vecs, const double* mu_lr, int64_t which,
                    const double* x, double* z, double* gz, double* logp, double* logdet, double* x_rt, double* logp_rt,
                    double* logdet_rt) {
    Ctx m{*cfg};
    Density d = make_density(LOGP_DIAG_NORMAL, dim, precision_diag, dim);
    MassMatrix mm(dim);
    if (!mm.update(m, Vec(stds, stds + dim), Vec(mean, mean + dim), Vec(vals, vals + n_eig), Vec(vecs, vecs + n_eig * dim),
                   Vec(mu_lr, mu_lr + dim))) return 3;
    Vec X(x, x + dim), GX(dim), Z(dim), GZ(dim);
    if (which == 0) { mm.compute_transformed_position(m, X, Z); for (uint64_t i = 0; i < dim; ++i) z[i] = Z[i]; return 0; }
    if (which == 1) { mm.compute_untransformed_position(m, X, Z); for (uint64_t i = 0; i < dim; ++i) z[i] = Z[i]; return 0; }
    if (which == 2) { mm.compute_transformed_gradient(m, X, Z); for (uint64_t i = 0; i < dim; ++i) z[i] = Z[i]; return 0; }
    int rc = d.logp(m, X.data(), GX.data(), logp);
    mm.compute_transformed_position(m, X, Z);
    mm.compute_transformed_gradient(m, GX, GZ);
    *logdet = mm.logdet;
    Vec XR(dim), GXR(dim), GZR(dim);
    mm.compute_untransformed_position(m, Z, XR);
    rc |= d.logp(m, XR.data(), GXR.data(), logp_rt);
    mm.compute_transformed_gradient(m, GXR, GZR);
    *logdet_rt = mm.logdet;
    for (uint64_t i = 0; i < dim; ++i) { z[i] = Z[i]; gz[i] = GZ[i]; x_rt[i] = XR[i]; }
    return rc;
}

// One leapfrog from (z, v, gz) with mass matrix (sigma, mu, logdet): the unit that nm_leapfrog_batch fuses.
int nmo_leapfrog(const MathCfg* cfg, int64_t kind, uint64_t dim, const double* params, uint64_t n_params,
                 const double* z, const double* v, const double* gz, const double* sigma, const double* mu,
                 double eps, double logdet, double initial_energy,
                 double* z_out, double* v_out, double* gz_out, double* x_out, double* gx_out,
                 double* logp_out, double* kinetic_out, double* energy_error_out) {
    Ctx m{*cfg};
    Density d = make_density(kind, dim, params, n_params);
    Hamiltonian h(&m, &d);
    for (uint64_t i = 0; i < dim; ++i) { h.mm.stds[i] = sigma[i]; h.mm.mean[i] = mu[i]; h.mm.inv_stds[i] = 1.0 / sigma[i]; }
    h.mm.logdet = logdet; h.mm.id = 0;
    h.step_size = std::fabs(eps);
    State s = std::make_shared<Point>(dim);
    for (uint64_t i = 0; i < dim; ++i) { s->z[i] = z[i]; s->v[i] = v[i]; s->gz[i] = gz[i]; }
    s->initial_energy = initial_energy; s->transform_id = 0;
    LeapfrogResult r = h.leapfrog(s, eps < 0 ? -1 : +1, 1.0, initial_energy, INFINITY, nullptr);
    if (r.kind != LF_OK) return 1;
    for (uint64_t i = 0; i < dim; ++i) {
        z_out[i] = r.state->z[i]; v_out[i] = r.state->v[i]; gz_out[i] = r.state->gz[i];
        x_out[i] = r.state->x[i]; gx_out[i] = r.state->gx[i];
    }
    *logp_out = r.state->logp; *kinetic_out = r.state->kinetic_energy;
    *energy_error_out = r.state->energy() - initial_energy;
    return 0;
}

// The per-vector primitives of the non-Euclidean trajectory kinds (reference src/math/util.rs:507-741,
// src/math/cpu_math.rs:496-551), one call each: op 0 std_norm_flow(pos = a, vel = b) -> out1 = pos_out, out2 = vel;
// 1 std_norm_grad_flow(pos = a, grad = b, vel = c) -> out1; 2 esh_momentum_update(gradient = a, momentum = b, step = eps)
// -> out1 = momentum, *scalar = kinetic-energy change; 3 array_normalize(a) -> out1; 4 sin / cos of eps -> out1[0], out1[1].
int nmo_traj_kat(const MathCfg* cfg, int64_t op, uint64_t n, const double* a, const double* b, const double* c, double eps,
                 double* out1, double* out2, double* scalar) {
    Ctx m{*cfg};
    switch (op) {
    case 0: { Vec v(b, b + n); m.std_norm_flow(a, out1, v.data(), eps, n); for (uint64_t i = 0; i < n; ++i) out2[i] = v[i]; return 0; }
    case 1: m.std_norm_grad_flow(a, b, c, out1, eps, n); return 0;
    case 2: { for (uint64_t i = 0; i < n; ++i) out1[i] = b[i]; *scalar = m.esh_momentum_update(a, out1, eps, n); return 0; }
    case 3: { for (uint64_t i = 0; i < n; ++i) out1[i] = a[i]; m.array_normalize(out1, n); return 0; }
    case 4: out1[0] = m.sin(eps); out1[1] = m.cos(eps); return 0;
    }
    return 1;
}

// nmo_leapfrog with a KineticEnergyKind: `kinetic_start` is the start point's kinetic_energy (the accumulated change for
// the microcanonical kind); `max_energy_error` as in the leapfrog (returns 1 for a divergence).
int nmo_leapfrog_traj(const MathCfg* cfg, int64_t traj_kind, int64_t kind, uint64_t dim, const double* params, uint64_t n_params,
                      const double* z, const double* v, const double* gz, const double* sigma, const double* mu,
                      double eps, double logdet, double initial_energy, double kinetic_start, double max_energy_error,
                      double* z_out, double* v_out, double* gz_out, double* x_out, double* gx_out,
                      double* logp_out, double* kinetic_out, double* energy_error_out) {
    Ctx m{*cfg};
    Density d = make_density(kind, dim, params, n_params);
    Hamiltonian h(&m, &d);
    h.kind = traj_kind;
    for (uint64_t i = 0; i < dim; ++i) { h.mm.stds[i] = sigma[i]; h.mm.mean[i] = mu[i]; h.mm.inv_stds[i] = 1.0 / sigma[i]; }
    h.mm.logdet = logdet; h.mm.id = 0;
    h.step_size = std::fabs(eps);
    State s = std::make_shared<Point>(dim);
    for (uint64_t i = 0; i < dim; ++i) { s->z[i] = z[i]; s->v[i] = v[i]; s->gz[i] = gz[i]; }
    s->initial_energy = initial_energy; s->transform_id = 0; s->kinetic_energy = kinetic_start;
    LeapfrogResult r = h.leapfrog(s, eps < 0 ? -1 : +1, 1.0, initial_energy, max_energy_error, nullptr);
    if (r.kind != LF_OK) return 1;
    for (uint64_t i = 0; i < dim; ++i) {
        z_out[i] = r.state->z[i]; v_out[i] = r.state->v[i]; gz_out[i] = r.state->gz[i];
        x_out[i] = r.state->x[i]; gx_out[i] = r.state->gx[i];
    }
    *logp_out = r.state->logp; *kinetic_out = r.state->kinetic_energy;
    *energy_error_out = r.state->energy() - initial_energy;
    return 0;
}

}  // extern "C"
