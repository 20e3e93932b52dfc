// nuts_amd.hpp — C++ host side above the C ABI of nuts_amd.h (header only, no dependencies beyond the C ABI).
//
// The reference is Rust; no Rust toolchain exists in the build image, so this is the compiled-language mirror of the
// reference's interface for the accelerated path: same names, argument meaning and error behaviour.
//   nuts_amd::DiagNutsSettings (+ nested option structs)   <-> DiagNutsSettings and its nested Default impls
//                                                             (src/sampler.rs:199-239, :630-634; src/adapt_strategy.rs:41-69;
//                                                              src/stepsize/adapt.rs:20-50, :308-329; src/stepsize/dual_avg.rs:12-31;
//                                                              src/stepsize/adam.rs:12-34; src/transform/adapt/diagonal.rs:92-106)
//   nuts_amd::ChainBatch::{set_position, draw, expanded_draw}  <-> Settings::new_chain + Chain::{set_position, draw,
//                                                             expanded_draw} (src/sampler.rs:53-63, src/chain.rs:24-42, :137-204)
//                                                             for n_chains chains at once
//   nuts_amd::Progress, ChainProgress                      <-> src/sampler.rs:165-174, :1009-1051
//   nuts_amd::Sampler::{pause, resume, progress, inspect, abort, wait_timeout}  <-> src/sampler.rs:1229-1552
//   nuts_amd::NutsError                                    <-> NutsError / anyhow::Error (src/nuts.rs:12-23)
// Link with -lnuts_amd (nuts_rs_amd/libnuts_amd.so).  There is no CPU fallback: creating a ChainBatch without a HIP
// device throws NutsError{NM_ERR_NO_DEVICE}.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "nuts_amd.h"

namespace nuts_amd {

struct NutsError : std::runtime_error {
    nm_status status;
    NutsError(nm_status s, const std::string& msg) : std::runtime_error(msg), status(s) {}
};
inline void check(nm_status st) {
    if (st != NM_OK) throw NutsError(st, nm_last_error());
}

struct DualAverageOptions { double k = 0.75, t0 = 10.0, gamma = 0.05, max_step_size = 3.14159265358979323846; };
struct AdamOptions { double beta1 = 0.9, beta2 = 0.999, epsilon = 1e-8, learning_rate = 0.05; };
enum class StepSizeAdaptMethod : uint64_t { DualAverage = NM_STEP_DUAL_AVERAGE, Adam = NM_STEP_ADAM, Fixed = NM_STEP_FIXED };
struct StepSizeAdaptOptions {
    StepSizeAdaptMethod method = StepSizeAdaptMethod::DualAverage;
    double fixed_step_size = 0.0;             // the payload of StepSizeAdaptMethod::Fixed(f64)
    DualAverageOptions dual_average;
    AdamOptions adam;
};
struct StepSizeSettings {
    double target_accept = 0.8, initial_step = 0.1;
    std::optional<double> jitter = 0.1;
    StepSizeAdaptOptions adapt_options;
};
struct DiagAdaptExpSettings { bool store_mass_matrix = false, use_grad_based_estimate = true; };
struct EuclideanAdaptOptions {
    StepSizeSettings step_size_settings;
    DiagAdaptExpSettings mass_matrix_options;
    double early_window = 0.3, step_size_window = 0.15;
    uint64_t mass_matrix_switch_freq = 80, early_mass_matrix_switch_freq = 10, mass_matrix_update_freq = 1;
    double mass_matrix_window_growth = 1.5;
};
enum class KineticEnergyKind : uint64_t {      // src/dynamics/transformed_hamiltonian.rs:27-50
    Euclidean = NM_TRAJ_EUCLIDEAN, ExactNormal = NM_TRAJ_EXACT_NORMAL, Microcanonical = NM_TRAJ_MICROCANONICAL };
struct DiagNutsSettings {
    uint64_t num_tune = 400, num_draws = 1000, maxdepth = 10, mindepth = 0;
    bool store_gradient = false, store_unconstrained = false, store_transformed = false;
    double max_energy_error = 1000.0;
    bool store_divergences = false;
    EuclideanAdaptOptions adapt_options;
    bool check_turning = true;
    std::optional<double> target_integration_time;
    KineticEnergyKind trajectory_kind = KineticEnergyKind::Euclidean;
    uint64_t num_chains = 6, seed = 0, extra_doublings = 0;

    nm_settings to_c() const {
        nm_settings s;
        nm_settings_default(&s);
        s.num_tune = num_tune; s.num_draws = num_draws; s.maxdepth = maxdepth; s.mindepth = mindepth;
        s.max_energy_error = max_energy_error; s.check_turning = check_turning; s.extra_doublings = extra_doublings;
        s.seed = seed; s.num_chains = num_chains; s.trajectory_kind = (uint64_t)trajectory_kind;
        s.store_gradient = store_gradient; s.store_unconstrained = store_unconstrained;
        s.store_transformed = store_transformed; s.store_divergences = store_divergences;
        s.has_target_integration_time = target_integration_time.has_value();
        s.target_integration_time = target_integration_time.value_or(0.0);
        const EuclideanAdaptOptions& a = adapt_options;
        s.early_window = a.early_window; s.step_size_window = a.step_size_window;
        s.mass_matrix_switch_freq = a.mass_matrix_switch_freq;
        s.early_mass_matrix_switch_freq = a.early_mass_matrix_switch_freq;
        s.mass_matrix_update_freq = a.mass_matrix_update_freq; s.mass_matrix_window_growth = a.mass_matrix_window_growth;
        s.store_mass_matrix = a.mass_matrix_options.store_mass_matrix;
        s.use_grad_based_estimate = a.mass_matrix_options.use_grad_based_estimate;
        const StepSizeSettings& st = a.step_size_settings;
        s.target_accept = st.target_accept; s.initial_step = st.initial_step;
        s.has_jitter = st.jitter.has_value(); s.jitter = st.jitter.value_or(0.0);
        s.step_size_method = (uint64_t)st.adapt_options.method; s.fixed_step_size = st.adapt_options.fixed_step_size;
        s.da_k = st.adapt_options.dual_average.k; s.da_t0 = st.adapt_options.dual_average.t0;
        s.da_gamma = st.adapt_options.dual_average.gamma; s.da_max_step_size = st.adapt_options.dual_average.max_step_size;
        s.adam_beta1 = st.adapt_options.adam.beta1; s.adam_beta2 = st.adapt_options.adam.beta2;
        s.adam_epsilon = st.adapt_options.adam.epsilon; s.adam_learning_rate = st.adapt_options.adam.learning_rate;
        return s;
    }
};

// `DiagMclmcSettings` = MclmcSettings<EuclideanAdaptOptions<DiagAdaptExpSettings>> (src/sampler.rs:266-374; experimental upstream)
enum class MclmcTrajectoryKind : uint64_t {    // src/mclmc.rs:44-70
    Microcanonical = NM_MCLMC_MICROCANONICAL, Euclidean = NM_MCLMC_EUCLIDEAN,
    EuclideanEarlyThenMicrocanonical = NM_MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL };
struct DiagMclmcSettings {
    double step_size = 0.5, momentum_decoherence_length = 3.0;
    uint64_t num_tune = 400, num_draws = 1000, num_chains = 6, seed = 0;
    double max_energy_error = 1000.0;
    bool store_unconstrained = false, store_gradient = false, store_transformed = false, store_divergences = false;
    EuclideanAdaptOptions adapt_options;
    double subsample_frequency = 1.0;
    bool dynamic_step_size = true;
    MclmcTrajectoryKind trajectory_kind = MclmcTrajectoryKind::EuclideanEarlyThenMicrocanonical;
    double trajectory_switch_fraction = 0.3;

    // the NutsSettings-shaped part (seed, counts, adaptation, store_* flags), as new_chain builds its GlobalStrategy from it
    DiagNutsSettings base() const {
        DiagNutsSettings n;
        n.num_tune = num_tune; n.num_draws = num_draws; n.num_chains = num_chains; n.seed = seed;
        n.max_energy_error = max_energy_error; n.store_unconstrained = store_unconstrained; n.store_gradient = store_gradient;
        n.store_transformed = store_transformed; n.store_divergences = store_divergences; n.adapt_options = adapt_options;
        return n;
    }
    nm_settings to_c() const {
        nm_settings s = base().to_c();
        s.sampler = NM_SAMPLER_MCLMC;
        s.mclmc_step_size = step_size; s.momentum_decoherence_length = momentum_decoherence_length;
        s.subsample_frequency = subsample_frequency; s.dynamic_step_size = dynamic_step_size;
        s.mclmc_trajectory_kind = (uint64_t)trajectory_kind; s.trajectory_switch_fraction = trajectory_switch_fraction;
        s.step_size_method = NM_STEP_FIXED; s.fixed_step_size = step_size;      // Fixed(self.step_size) (sampler.rs:421-423)
        return s;
    }
};

// `LowRankNutsSettings` = NutsSettings<EuclideanAdaptOptions<LowRankSettings>> (src/sampler.rs:245, Default :636-642;
// LowRankSettings src/transform/low_rank.rs:188-203).  The diagonal-only option struct of `adapt_options` is ignored.
struct LowRankSettings { bool store_mass_matrix = false; double gamma = 1e-5, eigval_cutoff = 2.0; };
struct LowRankNutsSettings {
    DiagNutsSettings nuts;                      // every other NutsSettings field
    LowRankSettings mass_matrix_options;
    LowRankNutsSettings() { nuts.num_tune = 800; nuts.adapt_options.mass_matrix_update_freq = 20; }
    const DiagNutsSettings& base() const { return nuts; }
    nm_settings to_c() const {
        nm_settings s = nuts.to_c();
        s.adaptation = NM_ADAPT_LOW_RANK; s.store_mass_matrix = mass_matrix_options.store_mass_matrix;
        s.lr_gamma = mass_matrix_options.gamma; s.lr_eigval_cutoff = mass_matrix_options.eigval_cutoff;
        return s;
    }
};
// `LowRankMclmcSettings` = MclmcSettings<EuclideanAdaptOptions<LowRankSettings>> (src/sampler.rs:325-328, Default :376-384)
struct LowRankMclmcSettings {
    DiagMclmcSettings mclmc;
    LowRankSettings mass_matrix_options;
    LowRankMclmcSettings() { mclmc.num_tune = 800; mclmc.adapt_options.early_mass_matrix_switch_freq = 20; }
    DiagNutsSettings base() const { return mclmc.base(); }
    nm_settings to_c() const {
        nm_settings s = mclmc.to_c();
        s.adaptation = NM_ADAPT_LOW_RANK; s.store_mass_matrix = mass_matrix_options.store_mass_matrix;
        s.lr_gamma = mass_matrix_options.gamma; s.lr_eigval_cutoff = mass_matrix_options.eigval_cutoff;
        return s;
    }
};

// A registered device density (the device-side stand-in for a `CpuLogpFunc`, src/math/cpu_math.rs:885-891)
struct LogpSpec {
    uint64_t kind = NM_LOGP_IID_NORMAL, dim = 0;
    std::vector<double> params;
    std::string module_path;
    nm_host_logp_fn host_fn = nullptr;   // NM_LOGP_HOST_CALLBACK: a `CpuLogpFunc::logp` on the host (src/math/cpu_math.rs:885-891)
    void* host_ctx = nullptr;
    uint64_t host_threads = 0;
    static LogpSpec make(uint64_t kind, uint64_t dim, std::vector<double> params, std::string path) {
        LogpSpec l;
        l.kind = kind; l.dim = dim; l.params = std::move(params); l.module_path = std::move(path);
        return l;
    }
    static LogpSpec iid_normal(uint64_t dim, double mu = 3.0) { return make(NM_LOGP_IID_NORMAL, dim, {mu}, ""); }
    static LogpSpec diag_normal(std::vector<double> precision_diag) {
        const uint64_t d = precision_diag.size();
        return make(NM_LOGP_DIAG_NORMAL, d, std::move(precision_diag), "");
    }
    static LogpSpec funnel(uint64_t dim = 101) { return make(NM_LOGP_FUNNEL, dim, {}, ""); }
    static LogpSpec eight_schools() {
        return make(NM_LOGP_EIGHT_SCHOOLS, 10, {28., 8., -3., 7., -1., 1., 18., 12., 15., 10., 16., 11., 9., 11., 10., 18.}, "");
    }
    static LogpSpec mvn_precision(uint64_t dim, std::vector<double> precision_row_major) {
        return make(NM_LOGP_MVN_PREC, dim, std::move(precision_row_major), "");
    }
    static LogpSpec module(uint64_t dim, std::string path, std::vector<double> params = {}) {
        return make(NM_LOGP_MODULE, dim, std::move(params), std::move(path));
    }
    // the slow, fully general route: fn(ctx, chain, dim, position, gradient, &logp) -> 0 ok | 1 recoverable | 2 fatal
    static LogpSpec host_callback(uint64_t dim, nm_host_logp_fn fn, void* ctx = nullptr, uint64_t threads = 0) {
        LogpSpec l = make(NM_LOGP_HOST_CALLBACK, dim, {}, "");
        l.host_fn = fn; l.host_ctx = ctx; l.host_threads = threads;
        return l;
    }
    nm_logp_spec to_c() const {
        return {kind, dim, (uint64_t)params.size(), params.empty() ? nullptr : params.data(),
                module_path.empty() ? nullptr : module_path.c_str(), host_fn, host_ctx, host_threads};
    }
};

struct Progress {                  // src/sampler.rs:165-174
    uint64_t draw, chain;
    bool diverging, tuning;
    double step_size;
    uint64_t num_steps;
};

// n_chains NUTS chains on one GPU: the batched `settings.new_chain(chain, math, rng)` of the reference
class ChainBatch {
public:
    ChainBatch(const DiagNutsSettings& settings, const LogpSpec& logp, uint64_t n_chains, uint64_t chain_id_offset = 0,
               int64_t device = -1)
        : settings_(settings), n_(n_chains), dim_(logp.dim), offset_(chain_id_offset) {
        nm_settings s = settings.to_c();
        nm_logp_spec l = logp.to_c();
        nm_engine_config cfg;
        nm_engine_config_default(&cfg);
        cfg.device = device; cfg.chain_id_offset = chain_id_offset;
        check(nm_engine_create(&s, &l, n_chains, &cfg, &h_));
    }
    // `<Settings>::new_chain` for every chain with the other settings types (DiagMclmcSettings, LowRankNutsSettings,
    // LowRankMclmcSettings): anything with to_c() and base()
    template <class S, class = decltype(std::declval<const S&>().base())>
    ChainBatch(const S& settings, const LogpSpec& logp, uint64_t n_chains, uint64_t chain_id_offset = 0,
               int64_t device = -1)
        : settings_(settings.base()), n_(n_chains), dim_(logp.dim), offset_(chain_id_offset) {
        nm_settings s = settings.to_c();
        nm_logp_spec l = logp.to_c();
        nm_engine_config cfg;
        nm_engine_config_default(&cfg);
        cfg.device = device; cfg.chain_id_offset = chain_id_offset;
        check(nm_engine_create(&s, &l, n_chains, &cfg, &h_));
    }
    // the same with every engine-side knob (tiling, grid, lane_groups, ...): start from nm_engine_config_default
    ChainBatch(const DiagNutsSettings& settings, const LogpSpec& logp, uint64_t n_chains, const nm_engine_config& cfg)
        : settings_(settings), n_(n_chains), dim_(logp.dim), offset_(cfg.chain_id_offset) {
        nm_settings s = settings.to_c();
        nm_logp_spec l = logp.to_c();
        check(nm_engine_create(&s, &l, n_chains, &cfg, &h_));
    }
    ~ChainBatch() { if (h_) nm_engine_destroy(h_); }
    // draw launches served by the several-chains-per-wavefront kernels (many chains with dim <= 64)
    uint64_t group_launches() const { return nm_engine_group_launches(h_); }
    ChainBatch(const ChainBatch&) = delete;
    ChainBatch& operator=(const ChainBatch&) = delete;

    uint64_t dim() const { return dim_; }
    uint64_t num_chains() const { return n_; }
    nm_engine* handle() { return h_; }

    // x0 ~ U(-1, 1) from each chain's outer generator: `CpuMath::init_position` in Sampler order
    std::vector<double> init_positions_uniform() const {
        std::vector<double> x0(n_ * dim_);
        check(nm_init_positions_uniform(settings_.seed, offset_, n_, dim_, x0.data()));
        return x0;
    }
    // `Chain::set_position` for every chain; x0 is [n_chains][dim].  Throws on BadInitGrad / LogpFailure like the reference.
    void set_position(const std::vector<double>& x0) {
        if (x0.size() != n_ * dim_) throw NutsError(NM_ERR_INVALID_ARG, "x0 must hold n_chains * dim values");
        check(nm_engine_set_positions(h_, x0.data(), nullptr));
    }
    // per-chain `Chain::set_position`: chains with mask[c] == 0 keep their whole state
    void set_position(const std::vector<double>& x0, const std::vector<uint8_t>& mask) {
        if (x0.size() != n_ * dim_ || mask.size() != n_) throw NutsError(NM_ERR_INVALID_ARG, "x0 / mask size");
        check(nm_engine_set_positions_masked(h_, x0.data(), mask.data(), nullptr));
    }
    // the init loop of the reference's ChainProcess (src/sampler.rs:1126-1147): failed chains retry with their next
    // init_position, at most max_tries times; x0 == nullptr starts from the uniform init_position
    std::vector<uint64_t> init_with_retries(const std::vector<double>* x0 = nullptr, uint64_t max_tries = 500) {
        std::vector<uint64_t> tries(n_);
        check(nm_engine_init_positions_retry(h_, x0 ? x0->data() : nullptr, max_tries, nullptr, tries.data()));
        return tries;
    }
    // One `Chain::draw` per chain: (positions [n_chains][dim], one Progress per chain)
    std::pair<std::vector<double>, std::vector<Progress>> draw() {
        std::vector<double> pos(n_ * dim_);
        std::vector<nm_draw_stats> st(n_);
        check(nm_engine_draw_to_host(h_, 1, pos.data(), st.data()));
        std::vector<Progress> prog;
        prog.reserve(n_);
        for (const nm_draw_stats& s : st)
            prog.push_back({s.draw, s.chain, s.diverging != 0, s.tuning != 0, s.step_size, s.n_steps});
        return {std::move(pos), std::move(prog)};
    }
    // n_draws draws of every chain into host arrays ([n_draws][n_chains][dim], [n_draws][n_chains]); either may be null
    void draw_many(uint64_t n_draws, double* positions, nm_draw_stats* stats) {
        check(nm_engine_draw_to_host(h_, n_draws, positions, stats));
    }
    // `Chain::expanded_draw` x n_draws: host pointers of `out` select the vector statistics (nm_draw_outputs)
    void expanded_draw_many(uint64_t n_draws, const nm_draw_outputs& host_out) {
        check(nm_engine_draw_ex_to_host(h_, n_draws, &host_out));
    }
    std::vector<double> positions() { std::vector<double> v(n_ * dim_); check(nm_engine_get_positions(h_, v.data())); return v; }
    std::vector<double> step_sizes() { std::vector<double> v(n_); check(nm_engine_get_step_sizes(h_, v.data())); return v; }
    std::pair<std::vector<double>, std::vector<double>> mass_matrix() {
        std::vector<double> sd(n_ * dim_), mu(n_ * dim_);
        check(nm_engine_get_mass_matrix(h_, sd.data(), mu.data()));
        return {std::move(sd), std::move(mu)};
    }

private:
    DiagNutsSettings settings_;
    uint64_t n_, dim_, offset_;
    nm_engine* h_ = nullptr;
};

struct ChainProgress {             // src/sampler.rs:1009-1051
    uint64_t finished_draws = 0, total_draws = 0, divergences = 0;
    bool tuning = true, started = false;
    uint64_t latest_num_steps = 0, total_num_steps = 0;
    double step_size = 0.0;
    std::chrono::duration<double> runtime{0};
    std::vector<uint64_t> divergent_draws;
    void update(const nm_draw_stats& s, std::chrono::duration<double> draw_duration) {
        if (s.diverging && !s.tuning) { divergences += 1; divergent_draws.push_back(finished_draws); }
        finished_draws += 1;
        tuning = s.tuning != 0;
        latest_num_steps = s.n_steps; total_num_steps += s.n_steps;
        step_size = s.step_size;
        runtime += draw_duration;
    }
};

struct Trace {                     // what the controller has collected so far
    uint64_t n_draws = 0, n_chains = 0, dim = 0;
    std::vector<double> positions;           // [n_draws][n_chains][dim]
    std::vector<nm_draw_stats> stats;        // [n_draws][n_chains]
};

// The reference's `Sampler` control plane over one ChainBatch: a controller thread advances all chains `chunk_draws`
// draws per kernel launch; commands act between launches (src/sampler.rs:1229-1552).
class Sampler {
public:
    enum class WaitKind { Trace, Timeout, Err };
    struct WaitResult { WaitKind kind; Trace trace; std::string error; };

    Sampler(const DiagNutsSettings& settings, const LogpSpec& logp, std::optional<std::vector<double>> x0 = std::nullopt,
            uint64_t chunk_draws = 16, int64_t device = -1)
        : total_(settings.num_tune + settings.num_draws), n_(settings.num_chains), dim_(logp.dim),
          chunk_(chunk_draws ? chunk_draws : 1), progress_(settings.num_chains) {
        for (ChainProgress& p : progress_) p.total_draws = total_;
        thread_ = std::thread([this, settings, logp, x0, device]() { run(settings, logp, x0, device); });
    }
    ~Sampler() { if (thread_.joinable()) { request_abort(); thread_.join(); } }

    void pause() { { std::lock_guard<std::mutex> g(m_); paused_ = true; } cv_.notify_all(); }
    void resume() { { std::lock_guard<std::mutex> g(m_); paused_ = false; } cv_.notify_all(); }
    void flush() {}
    std::vector<ChainProgress> progress() { std::lock_guard<std::mutex> g(m_); return progress_; }
    // (error or empty, trace so far) without stopping the sampler
    std::pair<std::string, Trace> inspect() { std::lock_guard<std::mutex> g(m_); return {error_, finished_part(false)}; }
    // stop after the launch in flight; (error or empty, trace so far)
    std::pair<std::string, Trace> abort() {
        request_abort();
        if (thread_.joinable()) thread_.join();
        std::lock_guard<std::mutex> g(m_);
        return {error_, finished_part(true)};
    }
    WaitResult wait_timeout(std::chrono::duration<double> timeout) {
        std::unique_lock<std::mutex> g(m_);
        if (!cv_.wait_for(g, timeout, [this] { return done_; })) return {WaitKind::Timeout, {}, ""};
        g.unlock();
        if (thread_.joinable()) thread_.join();
        g.lock();
        if (!error_.empty()) return {WaitKind::Err, finished_part(true), error_};
        return {WaitKind::Trace, finished_part(true), ""};
    }

private:
    void request_abort() { { std::lock_guard<std::mutex> g(m_); abort_ = true; } cv_.notify_all(); }
    void run(DiagNutsSettings settings, LogpSpec logp, std::optional<std::vector<double>> x0, int64_t device) {
        try {
            ChainBatch batch(settings, logp, n_, 0, device);
            batch.init_with_retries(x0 ? &*x0 : nullptr);
            { std::lock_guard<std::mutex> g(m_); for (ChainProgress& p : progress_) p.started = true;
              trace_.n_chains = n_; trace_.dim = dim_;
              // one trace for the whole run: the engine writes every chunk straight into its place (capacity reserved here, so the
              // storage never moves; rows beyond n_draws are the chunk in flight)
              trace_.positions.reserve(total_ * n_ * dim_); trace_.stats.reserve(total_ * n_); }
            uint64_t finished = 0;
            while (finished < total_) {
                {
                    std::unique_lock<std::mutex> g(m_);
                    cv_.wait(g, [this] { return !paused_ || abort_; });
                    if (abort_) break;
                }
                const uint64_t n = std::min<uint64_t>(chunk_, total_ - finished);
                double* pos; nm_draw_stats* st;
                {   std::lock_guard<std::mutex> g(m_);
                    trace_.positions.resize((finished + n) * n_ * dim_); trace_.stats.resize((finished + n) * n_);
                    pos = trace_.positions.data() + finished * n_ * dim_; st = trace_.stats.data() + finished * n_; }
                const auto t0 = std::chrono::steady_clock::now();
                batch.draw_many(n, pos, st);
                const std::chrono::duration<double> per_draw = (std::chrono::steady_clock::now() - t0) / (double)n;
                std::lock_guard<std::mutex> g(m_);
                trace_.n_draws += n;
                for (uint64_t t = 0; t < n; ++t)
                    for (uint64_t c = 0; c < n_; ++c) progress_[c].update(st[t * n_ + c], per_draw);
                finished += n;
            }
        } catch (const std::exception& e) {
            std::lock_guard<std::mutex> g(m_);
            error_ = e.what();
        }
        { std::lock_guard<std::mutex> g(m_); done_ = true; }
        cv_.notify_all();
    }

    // the finished draws (m_ held).  `consume` (wait_timeout / abort, which consume the sampler in the reference:
    // src/sampler.rs:1487-1552) hands the trace itself over once the controller thread has ended; inspect() never does — it
    // clones the finished rows (src/sampler.rs:1469-1485), however often and whenever it is called.
    Trace finished_part(bool consume) {
        if (consume && done_) {
            trace_.positions.resize(trace_.n_draws * n_ * dim_); trace_.stats.resize(trace_.n_draws * n_);
            Trace out = std::move(trace_);
            trace_ = Trace{};
            trace_.n_chains = out.n_chains; trace_.dim = out.dim;
            return out;
        }
        Trace t;
        t.n_draws = trace_.n_draws; t.n_chains = trace_.n_chains; t.dim = trace_.dim;
        t.positions.assign(trace_.positions.begin(), trace_.positions.begin() + (std::ptrdiff_t)(t.n_draws * n_ * dim_));
        t.stats.assign(trace_.stats.begin(), trace_.stats.begin() + (std::ptrdiff_t)(t.n_draws * n_));
        return t;
    }

    uint64_t total_, n_, dim_, chunk_;
    std::mutex m_;
    std::condition_variable cv_;
    bool paused_ = false, abort_ = false, done_ = false;
    std::string error_;
    std::vector<ChainProgress> progress_;
    Trace trace_;
    std::thread thread_;
};

}  // namespace nuts_amd
