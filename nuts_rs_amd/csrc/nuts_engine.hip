// nuts_engine.hip — host side of libnuts_amd.so: the C ABI of include/nuts_amd.h over the gfx950 kernels
// of nuts_kernels.hpp.  No CPU fallback: every entry point that computes needs a HIP device.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <dlfcn.h>
#include <new>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "nuts_launch.hpp"

using namespace nm;

// the matrix-core kernel for shared matrices lives in its own (tile-mode) translation unit: kern_tile_mvn_prec.hip.
// POD mirror of nm::tile::TileMats (nuts_tile.hpp) — that header switches the whole TU to tile mode, so it is not included here.
#ifndef NM_TILE_CHAINS
#define NM_TILE_CHAINS 16    // (nuts_tile.hpp: chains per block of the matrix-core kernels; tuning builds pass both to all units)
#endif
#ifndef NM_TILE_OCC
#define NM_TILE_OCC 1
#endif
namespace nm { namespace tile { struct TileMats { const double *ut, *u, *p; int dim, rank, dim_kp, rank_kp, dim_st, rank_st; }; } }
namespace nm { hipError_t launch_tile_mvn_prec(int dpl, int query, const KParams& P, const tile::TileMats& M, unsigned grid, hipStream_t stream, int* occ); }
namespace nm { hipError_t launch_tile_mvn_diag(int dpl, const KParams& P, const tile::TileMats& M, unsigned grid, hipStream_t stream); }

#include "zig_tables.hpp"
// rand_distr's ZIG_NORM_X then ZIG_NORM_F (257 + 257 fixed constants; tools/gen_ziggurat_tables.py)
static const double kZigX[257] = NM_ZIG_NORM_X, kZigF[257] = NM_ZIG_NORM_F;
static const struct ZigInit { double t[2 * 257]; ZigInit() { memcpy(t, kZigX, sizeof kZigX); memcpy(t + 257, kZigF, sizeof kZigF); } } kZigInit;
static const double* const kZigTablesPtr = kZigInit.t;
#define kZigTables kZigInit.t

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static nm_status fail(nm_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return st;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(NM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

extern "C" const char* nm_last_error(void) { return g_last_error.c_str(); }
static nm_status ensure_device(int64_t device);
// HBM streaming probes (probe_bw.hip)
extern "C" int nm_probe_bandwidth_impl(uint64_t kind, uint64_t bytes_per_array, uint64_t iters, double* ms_per_iter,
                                       uint64_t* bytes_read, uint64_t* bytes_written, const char** err);
extern "C" nm_status nm_probe_bandwidth(uint64_t kind, uint64_t bytes_per_array, uint64_t iters, double* ms_per_iter,
                                        uint64_t* bytes_read, uint64_t* bytes_written) {
    if (kind > NM_PROBE_COPY_NT) return fail(NM_ERR_INVALID_ARG, "unknown probe kind %llu", (unsigned long long)kind);
    nm_status st = ensure_device(-1);
    if (st != NM_OK) return st;
    const char* err = nullptr;
    const int rc = nm_probe_bandwidth_impl(kind, bytes_per_array, iters, ms_per_iter, bytes_read, bytes_written, &err);
    if (rc != 0) return fail(rc == 1 ? NM_ERR_INVALID_ARG : NM_ERR_HIP, "bandwidth probe: %s", err ? err : "?");
    return NM_OK;
}
extern "C" uint64_t nm_abi_version(void) { return NM_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------
// host-side random stream (chain keys, init positions): rand's ChaCha8Rng / seed_from_u64 semantics,
// same published algorithms as the device generator in dev_math.hpp.
// ---------------------------------------------------------------------------------------------
namespace {
inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
void host_chacha8_block(const uint32_t key[8], uint64_t counter, uint64_t stream, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32),
                      (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t x[16];
    memcpy(x, s, sizeof x);
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl32(x[d], 16);
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl32(x[b], 12);
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl32(x[d], 8);
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl32(x[b], 7);
    };
    for (int r = 0; r < 4; ++r) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}
struct HostRng {
    uint32_t key[8];
    uint64_t stream = 0, pos = 0, buf_block = ~0ull;
    uint32_t buf[16];
    static HostRng seed_from_u64(uint64_t state) {            // PCG32 expansion (rand_core)
        HostRng r;
        const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
        for (int c = 0; c < 8; ++c) {
            state = state * MUL + INC;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            r.key[c] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
        return r;
    }
    uint32_t next_u32() {
        uint64_t b = pos >> 4;
        if (b != buf_block) { host_chacha8_block(key, b, stream, buf); buf_block = b; }
        return buf[pos++ & 15];
    }
    uint64_t next_u64() { uint64_t lo = next_u32(); uint64_t hi = next_u32(); return (hi << 32) | lo; }
};
// outer generator of a chain (reference src/sampler.rs:1105-1106)
HostRng outer_rng(uint64_t seed, uint64_t chain_id) {
    HostRng r = HostRng::seed_from_u64(seed);
    r.stream = chain_id + 1;
    return r;
}
}  // namespace

extern "C" nm_status nm_chain_rng_key(uint64_t seed, uint64_t chain_id, uint8_t key_out[32]) {
    if (!key_out) return fail(NM_ERR_INVALID_ARG, "key_out is null");
    HostRng o = outer_rng(seed, chain_id);
    for (int i = 0; i < 8; ++i) {                              // ChaCha8Rng::try_from_rng (src/sampler.rs:761)
        uint32_t w = o.next_u32();
        for (int b = 0; b < 4; ++b) key_out[4 * i + b] = (uint8_t)(w >> (8 * b));
    }
    return NM_OK;
}

extern "C" nm_status nm_init_positions_uniform_at(uint64_t seed, uint64_t chain_id_offset, uint64_t n_chains, uint64_t dim,
                                                  uint64_t attempt, double* h_x0);
extern "C" nm_status nm_init_positions_uniform(uint64_t seed, uint64_t chain_id_offset, uint64_t n_chains,
                                               uint64_t dim, double* h_x0) {
    return nm_init_positions_uniform_at(seed, chain_id_offset, n_chains, dim, 0, h_x0);
}
extern "C" nm_status nm_init_positions_uniform_at(uint64_t seed, uint64_t chain_id_offset, uint64_t n_chains, uint64_t dim,
                                                  uint64_t attempt, double* h_x0) {
    if (!h_x0) return fail(NM_ERR_INVALID_ARG, "h_x0 is null");
    for (uint64_t c = 0; c < n_chains; ++c) {
        HostRng o = outer_rng(seed, chain_id_offset + c);
        for (int i = 0; i < 8; ++i) (void)o.next_u32();
        o.pos += 2 * attempt * dim;                            // the earlier attempts' draws (one u64 per coordinate)
        for (uint64_t d = 0; d < dim; ++d) {                   // CpuMath::init_position (cpu_math.rs:184-187)
            double val = (double)(o.next_u64() >> 11) * (1.0 / 9007199254740992.0);
            h_x0[c * dim + d] = val * 2.0 - 1.0;
        }
    }
    return NM_OK;
}

// ---------------------------------------------------------------------------------------------
// settings
// ---------------------------------------------------------------------------------------------
extern "C" void nm_settings_default(nm_settings* s) {
    memset(s, 0, sizeof *s);
    s->num_tune = 400; s->num_draws = 1000; s->maxdepth = 10; s->mindepth = 0;
    s->max_energy_error = 1000.0; s->check_turning = 1; s->extra_doublings = 0; s->seed = 0; s->num_chains = 6;
    s->early_window = 0.3; s->step_size_window = 0.15;
    s->mass_matrix_switch_freq = 80; s->early_mass_matrix_switch_freq = 10; s->mass_matrix_update_freq = 1;
    s->mass_matrix_window_growth = 1.5;
    s->store_mass_matrix = 0; s->use_grad_based_estimate = 1;
    s->target_accept = 0.8; s->initial_step = 0.1; s->has_jitter = 1; s->jitter = 0.1;
    s->step_size_method = NM_STEP_DUAL_AVERAGE; s->fixed_step_size = 0.0;
    s->da_k = 0.75; s->da_t0 = 10.; s->da_gamma = 0.05; s->da_max_step_size = 3.14159265358979323846;
    s->adam_beta1 = 0.9; s->adam_beta2 = 0.999; s->adam_epsilon = 1e-8; s->adam_learning_rate = 0.05;
    s->adaptation = NM_ADAPT_DIAG; s->lr_gamma = 1e-5; s->lr_eigval_cutoff = 2.0; s->freeze_transform = 0;
    s->trajectory_kind = NM_TRAJ_EUCLIDEAN;                                  // sampler.rs:528
    // default_mclmc_settings (sampler.rs:342-366); inert while sampler == NM_SAMPLER_NUTS
    s->sampler = NM_SAMPLER_NUTS; s->mclmc_step_size = 0.5; s->momentum_decoherence_length = 3.0; s->subsample_frequency = 1.0;
    s->dynamic_step_size = 1; s->mclmc_trajectory_kind = NM_MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL; s->trajectory_switch_fraction = 0.3;
}
extern "C" void nm_settings_default_mclmc(nm_settings* s) {        // DiagMclmcSettings::default (src/sampler.rs:368-374)
    nm_settings_default(s);
    s->sampler = NM_SAMPLER_MCLMC; s->step_size_method = NM_STEP_FIXED; s->fixed_step_size = s->mclmc_step_size;
}
extern "C" void nm_settings_default_low_rank(nm_settings* s) {     // LowRankNutsSettings::default (src/sampler.rs:636-642)
    nm_settings_default(s);
    s->num_tune = 800; s->mass_matrix_update_freq = 20; s->adaptation = NM_ADAPT_LOW_RANK;
}
extern "C" void nm_engine_config_default(nm_engine_config* c) {
    memset(c, 0, sizeof *c);
    c->device = -1;
}

// ---------------------------------------------------------------------------------------------
// kernel dispatch
// ---------------------------------------------------------------------------------------------

typedef int (*module_launch_fn)(int kind, const void* kparams, unsigned grid_blocks, void* stream, int* occ);
typedef void (*module_info_fn)(uint64_t out[6]);

static hipError_t launch(uint64_t logp_kind, int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ = nullptr,
                         module_launch_fn module = nullptr, int variant = 0) {     // variant: 0 plain, 1 LrWrap (low-rank transformation), 2 KinWrap (trajectory kinds), 3 cluster (dim > 4096), 4 cluster + KinWrap
    const bool lr = variant == 1;
    if (variant == 3 && logp_kind == NM_LOGP_MODULE) return module ? (hipError_t)module((int)kind, &P, grid, stream, occ) : hipErrorInvalidValue;   // a module built in cluster mode
    if (variant == 3) return launch_cluster(logp_kind, kind, P, grid, stream, occ);   // chains wider than one block (kern_cluster.hip)
    if (variant == 4) return launch_cluster_kin(logp_kind, kind, P, grid, stream, occ);
    if (logp_kind == NM_LOGP_MODULE) return module && variant == 0 ? (hipError_t)module((int)kind, &P, grid, stream, occ) : hipErrorInvalidValue;
    if (lr) {      // the kernels that carry the low-rank transformation (LrWrap<Density>, kern_lr_*.hip)
        switch (logp_kind) {
        case NM_LOGP_IID_NORMAL: return launch_iid_normal_lr(dpl, w, kind, P, grid, stream, occ);
        case NM_LOGP_DIAG_NORMAL: return launch_diag_normal_lr(dpl, w, kind, P, grid, stream, occ);
        case NM_LOGP_FUNNEL: return launch_funnel_lr(dpl, w, kind, P, grid, stream, occ);
        case NM_LOGP_EIGHT_SCHOOLS: return launch_eight_schools_lr(dpl, w, kind, P, grid, stream, occ);
        case NM_LOGP_MVN_PREC: return launch_mvn_prec_lr(dpl, w, kind, P, grid, stream, occ);
        case NM_LOGP_HOST_CALLBACK: return launch_host_cb_lr(dpl, w, kind, P, grid, stream, occ);
        }
        return hipErrorInvalidValue;
    }
    if (variant == 2) {      // the kernels that carry the non-Euclidean KineticEnergyKinds (KinWrap<Density>, kern_kin_*.hip)
        switch (logp_kind) {
        case NM_LOGP_IID_NORMAL: return launch_iid_normal_kin(dpl, w, kind, P, grid, stream, occ);
        case NM_LOGP_DIAG_NORMAL: return launch_diag_normal_kin(dpl, w, kind, P, grid, stream, occ);
        case NM_LOGP_FUNNEL: return launch_funnel_kin(dpl, w, kind, P, grid, stream, occ);
        case NM_LOGP_EIGHT_SCHOOLS: return launch_eight_schools_kin(dpl, w, kind, P, grid, stream, occ);
        case NM_LOGP_MVN_PREC: return launch_mvn_prec_kin(dpl, w, kind, P, grid, stream, occ);
        case NM_LOGP_HOST_CALLBACK: return launch_host_cb_kin(dpl, w, kind, P, grid, stream, occ);
        }
        return hipErrorInvalidValue;
    }
    switch (logp_kind) {
    case NM_LOGP_IID_NORMAL: return launch_iid_normal(dpl, w, kind, P, grid, stream, occ);
    case NM_LOGP_DIAG_NORMAL: return launch_diag_normal(dpl, w, kind, P, grid, stream, occ);
    case NM_LOGP_FUNNEL: return launch_funnel(dpl, w, kind, P, grid, stream, occ);
    case NM_LOGP_EIGHT_SCHOOLS: return launch_eight_schools(dpl, w, kind, P, grid, stream, occ);
    case NM_LOGP_MVN_PREC: return launch_mvn_prec(dpl, w, kind, P, grid, stream, occ);
    case NM_LOGP_HOST_CALLBACK: return launch_host_cb(dpl, w, kind, P, grid, stream, occ);
    }
    return hipErrorInvalidValue;
}
// choose (DPL, W) for a dim; requested_dpl / requested_w = 0 means automatic (fewest waves, then smallest tile)
static bool pick_tiling(uint64_t dim, uint64_t requested_dpl, uint64_t requested_w, int* dpl_out, int* w_out) {
    static const int combos[8][2] = {{2, 1}, {4, 1}, {8, 1}, {16, 1}, {8, 2}, {16, 2}, {4, 4}, {16, 4}};   // {DPL, W}
    for (auto& c : combos) {
        if (requested_dpl && (uint64_t)c[0] != requested_dpl) continue;
        if (requested_w && (uint64_t)c[1] != requested_w) continue;
        if ((uint64_t)c[0] * 64 * c[1] >= dim) { *dpl_out = c[0]; *w_out = c[1]; return true; }
    }
    return false;
}
extern "C" nm_status nm_pick_tiling(uint64_t dim, uint64_t requested_dpl, uint64_t requested_w, uint64_t* dpl_out, uint64_t* w_out) {
    int d = 0, w = 0;
    if (!dpl_out || !w_out) return fail(NM_ERR_INVALID_ARG, "null output");
    if (!pick_tiling(dim, requested_dpl, requested_w, &d, &w)) return fail(NM_ERR_UNSUPPORTED, "no tiling for dim %llu", (unsigned long long)dim);
    *dpl_out = (uint64_t)d; *w_out = (uint64_t)w;
    return NM_OK;
}
static int pick_dpl(uint64_t dim, uint64_t requested) {
    int dpl = 0, w = 0;
    return pick_tiling(dim, requested, 1, &dpl, &w) ? dpl : 0;
}
static nm_status check_logp(const nm_logp_spec* l) {
    if (!l) return fail(NM_ERR_INVALID_ARG, "logp spec is null");
    if (l->dim == 0) return fail(NM_ERR_INVALID_ARG, "dim must be > 0");
    switch (l->kind) {
    case NM_LOGP_IID_NORMAL:
        if (l->n_params != 1 || !l->h_params) return fail(NM_ERR_INVALID_ARG, "NM_LOGP_IID_NORMAL takes 1 parameter (mu)");
        return NM_OK;
    case NM_LOGP_DIAG_NORMAL:
        if (l->n_params != l->dim || !l->h_params) return fail(NM_ERR_INVALID_ARG, "NM_LOGP_DIAG_NORMAL takes dim parameters");
        return NM_OK;
    case NM_LOGP_FUNNEL:
        if (l->dim < 2) return fail(NM_ERR_INVALID_ARG, "NM_LOGP_FUNNEL needs dim >= 2 (v plus at least one x)");
        return NM_OK;
    case NM_LOGP_EIGHT_SCHOOLS:
        if (l->dim != 10 || l->n_params != 16 || !l->h_params) return fail(NM_ERR_INVALID_ARG, "NM_LOGP_EIGHT_SCHOOLS: dim 10, params = y[8], sigma[8]");
        return NM_OK;
    case NM_LOGP_MVN_PREC:
        if (l->n_params != l->dim * l->dim || !l->h_params) return fail(NM_ERR_INVALID_ARG, "NM_LOGP_MVN_PREC takes dim*dim parameters (the precision matrix, row-major)");
        if (l->dim > 2048) return fail(NM_ERR_UNSUPPORTED, "NM_LOGP_MVN_PREC: dim %llu > 2048 (the position is kept in LDS)", (unsigned long long)l->dim);
        for (uint64_t i = 0; i < l->dim; ++i)
            for (uint64_t j = 0; j < i; ++j)
                if (l->h_params[i * l->dim + j] != l->h_params[j * l->dim + i]) return fail(NM_ERR_INVALID_ARG, "NM_LOGP_MVN_PREC: the precision matrix must be symmetric (entry %llu,%llu)", (unsigned long long)i, (unsigned long long)j);
        return NM_OK;
    }
    if (l->kind == NM_LOGP_HOST_CALLBACK) {
        if (!l->host_fn) return fail(NM_ERR_INVALID_ARG, "NM_LOGP_HOST_CALLBACK needs host_fn");
        return NM_OK;
    }
    if (l->kind == NM_LOGP_MODULE) {
        if (!l->module_path) return fail(NM_ERR_INVALID_ARG, "NM_LOGP_MODULE needs module_path");
        if (l->n_params && !l->h_params) return fail(NM_ERR_INVALID_ARG, "NM_LOGP_MODULE: n_params without h_params");
        return NM_OK;
    }
    return fail(NM_ERR_INVALID_ARG, "unknown logp kind %llu", (unsigned long long)l->kind);
}

static nm_status ensure_device(int64_t device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(NM_ERR_NO_DEVICE, "no HIP device available (%s); libnuts_amd has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device >= 0) {
        if (device >= n) return fail(NM_ERR_INVALID_ARG, "device %lld out of range (%d devices)", (long long)device, n);
        HIP_TRY(hipSetDevice((int)device));
    }
    return NM_OK;
}

// a matrix in the matrix-core kernels' A-operand order (nuts_tile.hpp TileMats): [stripes][kpairs][64 lanes][2]
template <class F>
static std::vector<double> pack_mfma_operand(uint64_t R, uint64_t K, F&& elem) {
    const uint64_t st = (R + 15) / 16, kp = (K + 7) / 8;
    std::vector<double> out(st * kp * 128, 0.0);
    for (uint64_t s_ = 0; s_ < st; ++s_)
        for (uint64_t q = 0; q < kp; ++q)
            for (uint64_t l = 0; l < 64; ++l)
                for (uint64_t j = 0; j < 2; ++j) {
                    const uint64_t row = 16 * s_ + (l & 15), k = 8 * q + 4 * j + (l >> 4);
                    out[((s_ * kp + q) * 64 + l) * 2 + j] = (row < R && k < K) ? elem(row, k) : 0.0;
                }
    return out;
}

// ---------------------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------------------
struct nm_engine {
    nm_settings s;
    nm_engine_config cfg;
    uint64_t logp_kind = 0, dim = 0, n_chains = 0;
    int dpl = 0, wpc = 1;         // doubles per lane, waves per chain
    int device = 0;
    bool positioned = false;
    KParams P;
    double* d_pvec = nullptr;     // [n_chains][NUM_PSLOT][dpad]
    double* d_svec = nullptr;     // [n_waves][nsslot][dpad]
    unsigned n_waves = 0;         // resident waves of the draw kernel = its grid
    unsigned group_grid = 0;      // > 0: the 8-lanes-per-chain kernel (nuts_group.hpp) serves the draw launches
    ChainScalars* d_sc = nullptr;
    unsigned long long* d_prof = nullptr;   // 32 cycle counters for NM_PROF builds
    double* d_zig = nullptr;      // x[257] then f[257]
    double* d_params = nullptr;
    double* d_x0 = nullptr;
    uint8_t* d_init_mask = nullptr;
    void* staging[11] = {};                 // device staging of the *_to_host calls, one per output array, grow-only
    size_t staging_bytes[11] = {};
    hipStream_t copy_stream = nullptr;      // device -> host copies of the *_to_host calls, overlapping the next chunk's kernel
    hipEvent_t ev_chunk[2] = {nullptr, nullptr};
    void* module_handle = nullptr;          // NM_LOGP_MODULE: dlopen handle and its launch entry
    module_launch_fn module_launch = nullptr;
    int module_group_lanes = 0;             // lanes per chain of the module's group form (0: it has none)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double kernel_ms = 0.0;
    uint64_t kernel_launches = 0;
    uint64_t steps_base = 0, draws_total = 0;
    uint64_t group_launches = 0;
    uint64_t draws_launched = 0;            // draw index of the healthy chain that is furthest behind (set_positions + launches)
    bool pending_timing = false;
    // low-rank transformation (settings.adaptation == NM_ADAPT_LOW_RANK)
    bool lr = false;
    int variant = 0;            // kernel family: 0 plain, 1 LrWrap, 2 KinWrap (see launch())
    // chains wider than one block (dim > 4096): cl_k blocks per chain, each with its own persistent vectors and copy of the scalars
    uint64_t cl_k = 1;
    unsigned long long *d_cl_box = nullptr, *d_cl_cnt = nullptr;
    uint64_t n_clusters = 0;
    uint64_t lr_rmax = 0, lr_cap = 0;
    double *d_lrvec = nullptr, *d_lrval = nullptr, *d_lrwin = nullptr;
    nm_lowrank_estimator_fn lr_estimator = nm_lowrank_compute_update;
    void* lr_estimator_ctx = nullptr;
    uint64_t lr_threads = 0;
    uint64_t lr_updates = 0, lr_rounds = 0;  // estimator calls / pause-resume rounds so far
    // staging of an estimator round (grow-only): the pending chains' windows come down in one batch of asynchronous copies into
    // pinned memory, their updates go up as ONE packed buffer that a kernel scatters into the chains' slots
    double *h_lr_win = nullptr, *h_lr_upd = nullptr, *d_lr_upd = nullptr;
    uint64_t* d_lr_meta = nullptr;
    size_t h_lr_win_bytes = 0, h_lr_upd_bytes = 0, d_lr_upd_bytes = 0, d_lr_meta_bytes = 0;
    double lr_host_seconds = 0.0;
    double lr_download_seconds = 0.0, lr_estimator_seconds = 0.0, lr_upload_seconds = 0.0;   // its parts
    // shared transformation + full-precision normal: 16 chains per block, products on the matrix cores (nuts_tile.hpp)
    bool tile_active = false;
    bool tile_diag_active = false;          // DiagNutsSettings on the full-precision normal: P x on the matrix cores, per-chain mass matrices
    tile::TileMats tile_mats = {};
    double *d_tile_ut = nullptr, *d_tile_u = nullptr, *d_tile_p = nullptr;
    unsigned tile_grid = 0;
    uint64_t tile_launches = 0;
    std::vector<double> h_params;           // the density's parameters (the tile kernel packs P from them)
    // NM_LOGP_HOST_CALLBACK: mailboxes in pinned host memory and the threads that answer them
    nm_host_logp_fn cb_fn = nullptr;
    void* cb_ctx = nullptr;
    unsigned char* cb_mail = nullptr;       // host pointer (hipHostMalloc, mapped, coherent)
    unsigned char* cb_mail_dev = nullptr;   // the same memory as the device sees it
    uint64_t cb_stride = 0;
    std::vector<std::thread> cb_threads;
    std::atomic<int> cb_stop{0}, cb_active{0};
    std::atomic<uint64_t> cb_calls{0};
};


// ---------------------------------------------------------------------------------------------
// Stream discipline (DESIGN "Stream discipline"): device memory the library owns is only ever touched on the stream of the
// handle that owns it (nm_engine::stream / the caller's `stream` argument of the batch helpers), never on the null stream.
// The handles' streams are hipStreamNonBlocking, so a null-stream hipMemset / kernel is NOT ordered against them (round 2's
// red test: a zero fill that landed after the kernel it should have preceded).  Host <-> device copies go through copy_on /
// copy2d_on: enqueued on the owning stream, and the host waits for THAT stream — ordered after everything queued before it,
// complete (so the host buffer may be reused or freed) when the call returns.  tests/test_stream_discipline.py greps the
// sources for the forbidden forms.
// ---------------------------------------------------------------------------------------------
static hipError_t copy_on(hipStream_t s, void* dst, const void* src, size_t n, hipMemcpyKind kind) {
    if (n == 0) return hipSuccess;
    const hipError_t er = hipMemcpyAsync(dst, src, n, kind, s);
    return er != hipSuccess ? er : hipStreamSynchronize(s);
}
static hipError_t copy2d_on(hipStream_t s, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind) {
    if (width == 0 || height == 0) return hipSuccess;
    const hipError_t er = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, s);
    return er != hipSuccess ? er : hipStreamSynchronize(s);
}

static void engine_free(nm_engine* e) {
    if (!e) return;
    e->cb_stop.store(1);
    for (auto& t : e->cb_threads) if (t.joinable()) t.join();
    if (e->cb_mail) (void)hipHostFree(e->cb_mail);
    if (e->module_handle) dlclose(e->module_handle);
    for (void* q : e->staging) if (q) (void)hipFree(q);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    for (hipEvent_t q : e->ev_chunk) if (q) (void)hipEventDestroy(q);
    if (e->d_pvec) (void)hipFree(e->d_pvec);
    if (e->d_svec) (void)hipFree(e->d_svec);
    if (e->d_sc) (void)hipFree(e->d_sc);
    if (e->d_cl_box) (void)hipFree(e->d_cl_box);
    if (e->h_lr_win) (void)hipHostFree(e->h_lr_win);
    if (e->h_lr_upd) (void)hipHostFree(e->h_lr_upd);
    if (e->d_lr_upd) (void)hipFree(e->d_lr_upd);
    if (e->d_lr_meta) (void)hipFree(e->d_lr_meta);
    if (e->d_cl_cnt) (void)hipFree(e->d_cl_cnt);
    if (e->d_prof) (void)hipFree(e->d_prof);
    if (e->d_zig) (void)hipFree(e->d_zig);
    if (e->d_params) (void)hipFree(e->d_params);
    if (e->d_x0) (void)hipFree(e->d_x0);
    if (e->d_init_mask) (void)hipFree(e->d_init_mask);
    if (e->d_lrvec) (void)hipFree(e->d_lrvec);
    if (e->d_lrval) (void)hipFree(e->d_lrval);
    if (e->d_lrwin) (void)hipFree(e->d_lrwin);
    if (e->d_tile_ut) (void)hipFree(e->d_tile_ut);
    if (e->d_tile_u) (void)hipFree(e->d_tile_u);
    if (e->d_tile_p) (void)hipFree(e->d_tile_p);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}
// the chains' scalars (cluster mode: the copy of every chain's first member)
static hipError_t read_scalars(nm_engine* e, ChainScalars* out) {
    if (e->cl_k == 1) return copy_on(e->stream, out, e->d_sc, e->n_chains * sizeof(ChainScalars), hipMemcpyDeviceToHost);
    return copy2d_on(e->stream, out, sizeof(ChainScalars), e->d_sc, e->cl_k * sizeof(ChainScalars), sizeof(ChainScalars), e->n_chains, hipMemcpyDeviceToHost);
}

extern "C" void nm_engine_destroy(nm_engine* e) { engine_free(e); }

extern "C" nm_status nm_engine_create(const nm_settings* settings, const nm_logp_spec* logp, uint64_t n_chains,
                                      const nm_engine_config* cfg_in, nm_engine** out) {
    if (!settings || !out) return fail(NM_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    nm_status st = check_logp(logp);
    if (st != NM_OK) return st;
    if (n_chains == 0) return fail(NM_ERR_INVALID_ARG, "n_chains must be > 0");
    nm_settings s = *settings;
    if (s.maxdepth > (uint64_t)MAX_MAXDEPTH) return fail(NM_ERR_UNSUPPORTED, "maxdepth %llu > %d", (unsigned long long)s.maxdepth, MAX_MAXDEPTH);
    if (s.step_size_method > NM_STEP_FIXED) return fail(NM_ERR_INVALID_ARG, "step_size_method %llu is not one of NM_STEP_*", (unsigned long long)s.step_size_method);
    // GlobalStrategy::new asserts (adapt_strategy.rs:83-84)
    const double num_tune_f = (double)s.num_tune;
    const uint64_t step_size_window = (uint64_t)(s.step_size_window * num_tune_f);
    const uint64_t early_end = (uint64_t)(s.early_window * num_tune_f);
    if (!(early_end < s.num_tune)) return fail(NM_ERR_INVALID_ARG, "early_end < num_tune violated (reference asserts, adapt_strategy.rs:83)");
    if (!(s.mass_matrix_window_growth >= 1.0)) return fail(NM_ERR_INVALID_ARG, "mass_matrix_window_growth must be >= 1");
    if (s.has_jitter && !(1.0 - s.jitter < 1.0 + s.jitter)) return fail(NM_ERR_INVALID_ARG, "invalid jitter");
    if (s.adaptation > NM_ADAPT_LOW_RANK) return fail(NM_ERR_INVALID_ARG, "adaptation %llu is not one of NM_ADAPT_*", (unsigned long long)s.adaptation);
    const bool lr = s.adaptation == NM_ADAPT_LOW_RANK;
    if (lr && logp->kind == NM_LOGP_MODULE) return fail(NM_ERR_UNSUPPORTED, "NM_ADAPT_LOW_RANK with a density module: modules carry the diagonal kernels only");
    if (lr && !(s.lr_gamma > 0.0) ) return fail(NM_ERR_INVALID_ARG, "lr_gamma must be > 0");
    if (s.trajectory_kind > NM_TRAJ_MICROCANONICAL) return fail(NM_ERR_INVALID_ARG, "trajectory_kind %llu is not one of NM_TRAJ_*", (unsigned long long)s.trajectory_kind);
    if (s.sampler > NM_SAMPLER_MCLMC) return fail(NM_ERR_INVALID_ARG, "sampler %llu is not one of NM_SAMPLER_*", (unsigned long long)s.sampler);
    const bool mclmc = s.sampler == NM_SAMPLER_MCLMC;
    if (mclmc) {
        if (s.mclmc_trajectory_kind > NM_MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL) return fail(NM_ERR_INVALID_ARG, "mclmc_trajectory_kind %llu is not one of NM_MCLMC_*", (unsigned long long)s.mclmc_trajectory_kind);
        if (!(s.mclmc_step_size > 0.0) || !(s.momentum_decoherence_length > 0.0)) return fail(NM_ERR_INVALID_ARG, "mclmc_step_size and momentum_decoherence_length must be > 0");
        if (logp->dim < 2) return fail(NM_ERR_INVALID_ARG, "ESH dynamics requires at least 2 dimensions (reference src/math/cpu_math.rs:514)");
        // DiagMclmcSettings::new_chain: adapt_options.step_size_settings.adapt_options.method = Fixed(self.step_size) (sampler.rs:421-423)
        s.step_size_method = NM_STEP_FIXED; s.fixed_step_size = s.mclmc_step_size;
        s.trajectory_kind = NM_TRAJ_EUCLIDEAN;      // (a NutsSettings field; MclmcChain's kind lives in the chain state)
    }
    const bool kin = mclmc || s.trajectory_kind != NM_TRAJ_EUCLIDEAN;
    if (kin && logp->kind == NM_LOGP_MODULE) return fail(NM_ERR_UNSUPPORTED, "the non-Euclidean trajectory kinds / NM_SAMPLER_MCLMC with a density module: modules carry the Euclidean NUTS kernels only");
    if (s.trajectory_kind == NM_TRAJ_MICROCANONICAL && logp->dim < 2) return fail(NM_ERR_INVALID_ARG, "ESH dynamics requires at least 2 dimensions (reference src/math/cpu_math.rs:514)");
    nm_engine_config cfg;
    if (cfg_in) cfg = *cfg_in; else nm_engine_config_default(&cfg);
    int dpl = 0, wv = 0;
    uint64_t cl_k = 1;
    constexpr uint64_t CL_SLICE = 4096, CL_MAX_K = CL_MAX_MEMBERS;
    if (logp->dim > CL_SLICE) {     // wider than one block: ceil(dim / 4096) blocks per chain (kern_cluster.hip)
        cl_k = (logp->dim + CL_SLICE - 1) / CL_SLICE;
        if (cl_k > CL_MAX_K) return fail(NM_ERR_UNSUPPORTED, "dim %llu > %llu", (unsigned long long)logp->dim, (unsigned long long)(CL_SLICE * CL_MAX_K));
        if (logp->kind != NM_LOGP_IID_NORMAL && logp->kind != NM_LOGP_DIAG_NORMAL && logp->kind != NM_LOGP_HOST_CALLBACK && logp->kind != NM_LOGP_MODULE)
            return fail(NM_ERR_UNSUPPORTED, "dim %llu > 4096: chains wider than one block exist for the element-wise densities (iid / diagonal normal), for density modules built for wide chains and for NM_LOGP_HOST_CALLBACK", (unsigned long long)logp->dim);
        if (logp->kind == NM_LOGP_MODULE && kin) return fail(NM_ERR_UNSUPPORTED, "density modules carry the Euclidean NUTS kernels only");
        if (lr) return fail(NM_ERR_UNSUPPORTED, "dim %llu > 4096: the diagonal adaptation only", (unsigned long long)logp->dim);
        if ((cfg.dims_per_lane && cfg.dims_per_lane != 16) || (cfg.waves_per_chain && cfg.waves_per_chain != 4))
            return fail(NM_ERR_UNSUPPORTED, "dim %llu > 4096 runs on the (16 doubles, 4 waves) tiling", (unsigned long long)logp->dim);
        dpl = 16; wv = 4;
    } else
    if (!pick_tiling(logp->dim, logp->kind == NM_LOGP_EIGHT_SCHOOLS ? 2 : cfg.dims_per_lane,
                     logp->kind == NM_LOGP_EIGHT_SCHOOLS ? 1 : cfg.waves_per_chain, &dpl, &wv))
        return fail(NM_ERR_UNSUPPORTED, "no tiling for dim %llu with dims_per_lane %llu / waves_per_chain %llu (max dim 4096 = 16 doubles x 64 lanes x 4 waves)",
                    (unsigned long long)logp->dim, (unsigned long long)cfg.dims_per_lane, (unsigned long long)cfg.waves_per_chain);
    st = ensure_device(cfg.device);
    if (st != NM_OK) return st;

    nm_engine* e = new (std::nothrow) nm_engine();
    if (!e) return fail(NM_ERR_HIP, "out of host memory");
    e->s = s; e->cfg = cfg; e->logp_kind = logp->kind; e->dim = logp->dim; e->n_chains = n_chains; e->dpl = dpl; e->wpc = wv;
    e->lr = lr;
    e->variant = cl_k > 1 ? (kin ? 4 : 3) : lr ? 1 : (kin ? 2 : 0);
    e->cl_k = cl_k;
    (void)hipGetDevice(&e->device);
    const uint64_t dpad = 64ull * (uint64_t)dpl * (uint64_t)wv;
    // (tilings with batched merges keep the z of the last NM_RING leaves behind the tree's other scratch slots: resolve_chunk)
    const uint64_t nsslot = (uint64_t)num_sslots((int)s.maxdepth) + ((dpl <= 4 && wv == 1 && cl_k == 1) ? (uint64_t)NM_RING : 0);
#define E_TRY(expr)                                                                                 \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) { engine_free(e); return fail(NM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } \
    } while (0)
    if (logp->kind == NM_LOGP_MODULE) {
        e->module_handle = dlopen(logp->module_path, RTLD_NOW | RTLD_LOCAL);
        if (!e->module_handle) { const char* why = dlerror(); std::string msg = why ? why : "?"; engine_free(e); return fail(NM_ERR_INVALID_ARG, "cannot load density module %s: %s", logp->module_path, msg.c_str()); }
        module_info_fn info = (module_info_fn)dlsym(e->module_handle, "nm_module_info");
        e->module_launch = (module_launch_fn)dlsym(e->module_handle, "nm_module_launch");
        if (!info || !e->module_launch) { engine_free(e); return fail(NM_ERR_INVALID_ARG, "%s is not a density module (nm_module_info / nm_module_launch missing)", logp->module_path); }
        uint64_t mi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        info(mi);
        e->module_group_lanes = (int)mi[4];
        if ((mi[5] != 0) != (cl_k > 1)) { engine_free(e); return fail(NM_ERR_INVALID_ARG, cl_k > 1 ? "dim %llu > 4096 needs a density module built for wide chains (-DNM_CLUSTER_MODE=1, init_slice)" : "this density module was built for wide chains (dim > 4096), dim is %llu", (unsigned long long)logp->dim); }
        if (mi[0] != sizeof(KParams) || mi[1] != NM_ABI_VERSION) { engine_free(e); return fail(NM_ERR_INVALID_ARG, "density module built against another engine (kernel parameters %llu bytes / ABI %llu, engine %zu / %d)", (unsigned long long)mi[0], (unsigned long long)mi[1], sizeof(KParams), NM_ABI_VERSION); }
        if ((int)mi[2] != dpl || (int)mi[3] != wv) { engine_free(e); return fail(NM_ERR_INVALID_ARG, "density module was built for tiling (%llu doubles per lane, %llu waves) but dim %llu uses (%d, %d): rebuild it with nm_pick_tiling's answer", (unsigned long long)mi[2], (unsigned long long)mi[3], (unsigned long long)logp->dim, dpl, wv); }
    }
    E_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    E_TRY(hipEventCreate(&e->ev0));
    E_TRY(hipEventCreate(&e->ev1));
    {   // one wave per resident slot of the chip (waves stride over the chains); tree scratch belongs to the wave
        int occ = 0, cus = 0;
        KParams dummy;
        memset(&dummy, 0, sizeof dummy);
        E_TRY(launch(logp->kind, dpl, wv, K_QUERY, dummy, 0, nullptr, &occ, e->module_launch, e->variant));
        E_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
        uint64_t resident = (uint64_t)(occ > 0 ? occ : 1) * (uint64_t)(cus > 0 ? cus : 1);
        const uint64_t wave_slots = resident;                      // chains the wave-per-chain kernel runs at once
        if (cfg.grid_blocks) resident = cfg.grid_blocks;            // tuning override: blocks in the grid
        e->n_waves = (unsigned)(n_chains < resident ? n_chains : resident);
        if (cl_k > 1) {     // whole clusters only, 8 x cl_k blocks at a time (block -> (cluster, member) map of the kernels); never
            const uint64_t unit = 8 * cl_k;                                  // more blocks than the chip holds at once (the members spin)
            uint64_t units = resident / unit;
            if (units == 0) { engine_free(e); return fail(NM_ERR_UNSUPPORTED, "dim %llu needs %llu co-resident blocks, the device holds %llu", (unsigned long long)logp->dim, (unsigned long long)unit, (unsigned long long)resident); }
            const uint64_t need = (n_chains + 7) / 8;
            if (units > need) units = need;
            e->n_waves = (unsigned)(units * unit);
            e->n_clusters = units * 8;
        }
        // small chains, more of them than the chip has wavefront slots: several chains per wave (nuts_group.hpp)
        const int gs = grp::group_size(logp->dim);
        const bool group_density = (logp->kind != NM_LOGP_MODULE && logp->kind != NM_LOGP_HOST_CALLBACK) || (logp->kind == NM_LOGP_MODULE && gs && e->module_group_lanes == gs);   // every built-in density has a group form
        // (the full-precision normal beyond 32 dims is faster on the matrix cores than two chains per wavefront: 3.7e8 against 3.0e8
        // leapfrogs/s at dim 64, tools/probes/mvn_small_dims.py)
        const bool mvn_tiles = logp->kind == NM_LOGP_MVN_PREC && logp->dim > 32 && cfg.chain_tiles != 1 && n_chains >= 256 && cfg.lane_groups != 2;
        if (cl_k == 1 && !lr && !kin && !mvn_tiles && cfg.lane_groups != 1 && group_density && gs && (gs == 8 || logp->kind != NM_LOGP_EIGHT_SCHOOLS) && dpl == 2 && wv == 1 &&
            s.maxdepth <= (uint64_t)grp::GMAXDEPTH && (n_chains > wave_slots || cfg.lane_groups == 2)) {   // measured crossover (K4): 2048 chains
            int gocc = 0;
            dummy.dim = logp->dim;       // the group size follows the dim
            E_TRY(launch(logp->kind, dpl, wv, K_GROUP_QUERY, dummy, 0, nullptr, &gocc, e->module_launch));
            uint64_t gres = (uint64_t)(gocc > 0 ? gocc : 1) * (uint64_t)(cus > 0 ? cus : 1);
            if (cfg.grid_blocks) gres = cfg.grid_blocks;
            const uint64_t need = (n_chains + (64 / gs) - 1) / (64 / gs);
            e->group_grid = (unsigned)(need < gres ? need : gres);
        }
    }
    const size_t pvec_bytes = (size_t)n_chains * cl_k * NUM_PSLOT * dpad * sizeof(double);   // cluster mode: one set per member
    {   // the matrix-core kernel's grid: 16-chain tiles on resident blocks (one 16-wave block fills a CU's wave slots)
        int cus = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device);
        const uint64_t n_tiles = (n_chains + NM_TILE_CHAINS - 1) / NM_TILE_CHAINS;
        e->tile_grid = (unsigned)std::min<uint64_t>(n_tiles, (uint64_t)(cus > 0 ? cus : 1) * NM_TILE_OCC);
    }
    size_t scratch_slots = e->n_waves > e->group_grid ? e->n_waves : e->group_grid;
    if (logp->kind == NM_LOGP_MVN_PREC && cfg.chain_tiles != 1) scratch_slots = std::max<size_t>(scratch_slots, (size_t)e->tile_grid * NM_TILE_CHAINS);
    const size_t svec_bytes = scratch_slots * nsslot * dpad * sizeof(double);
    E_TRY(hipMalloc(&e->d_pvec, pvec_bytes));
    E_TRY(hipMemsetAsync(e->d_pvec, 0, pvec_bytes, e->stream));
    E_TRY(hipMalloc(&e->d_svec, svec_bytes));
    E_TRY(hipMemsetAsync(e->d_svec, 0, svec_bytes, e->stream));
    E_TRY(hipMalloc(&e->d_sc, n_chains * cl_k * sizeof(ChainScalars)));
    if (cl_k > 1) {
        E_TRY(hipMalloc(&e->d_cl_box, e->n_clusters * CL_BOX_WORDS * cl_k * RED_MAX_VALUES * sizeof(unsigned long long)));
        E_TRY(hipMalloc(&e->d_cl_cnt, e->n_clusters * sizeof(unsigned long long)));
        E_TRY(hipMemsetAsync(e->d_cl_box, 0, e->n_clusters * CL_BOX_WORDS * cl_k * RED_MAX_VALUES * sizeof(unsigned long long), e->stream));
    }
    E_TRY(hipMalloc(&e->d_prof, 32 * sizeof(unsigned long long)));
    E_TRY(hipMemsetAsync(e->d_prof, 0, 32 * sizeof(unsigned long long), e->stream));
    E_TRY(hipMalloc(&e->d_zig, 2 * 257 * sizeof(double)));
    E_TRY(hipMalloc(&e->d_params, (logp->n_params ? logp->n_params : 1) * sizeof(double)));
    E_TRY(hipMalloc(&e->d_x0, n_chains * logp->dim * sizeof(double)));
    if (logp->n_params) E_TRY(copy_on(e->stream, e->d_params, logp->h_params, logp->n_params * sizeof(double), hipMemcpyHostToDevice));
    if (lr && logp->kind == NM_LOGP_MVN_PREC) e->h_params.assign(logp->h_params, logp->h_params + logp->n_params);
    // DiagNutsSettings on the full-precision normal: the density's P is shared by all chains whatever their mass matrices are,
    // so P x can run on the matrix cores 16 chains at a time (nuts_tile_diag_kernel) — same bits as the one-chain GEMV.
    // chain_tiles: 0 = when there are enough chains to fill the blocks, 1 = never, 2 = whenever it applies
    if (!lr && !kin && cl_k == 1 && logp->kind == NM_LOGP_MVN_PREC && cfg.chain_tiles != 1 && wv == 1 && (dpl == 2 || dpl == 4) &&
        logp->dim <= 256 && e->group_grid == 0 && (cfg.chain_tiles == 2 || n_chains >= 256)) {      // (any dim: the packed P is zero-padded)
        const uint64_t dim = logp->dim;
        const double* P_ = logp->h_params;
        const std::vector<double> pp = pack_mfma_operand(dim, dim, [&](uint64_t d, uint64_t j) { return P_[j * dim + d]; });
        E_TRY(hipMalloc(&e->d_tile_p, pp.size() * 8));
        E_TRY(copy_on(e->stream, e->d_tile_p, pp.data(), pp.size() * 8, hipMemcpyHostToDevice));
        e->tile_mats = {nullptr, nullptr, e->d_tile_p, (int)dim, 0, (int)((dim + 7) / 8), 0, (int)((dim + 15) / 16), 0};
        e->tile_diag_active = true;
    }
    if (lr) {   // eigenvector slots, eigenvalue arrays and the window of draws / gradients of every chain
        // the estimator's rank is <= min(dim, 2 n_draws); a transformation given from outside may have any rank <= dim
        const uint64_t most = s.freeze_transform ? logp->dim : std::min<uint64_t>(logp->dim, 2 * (s.num_tune + 1));
        e->lr_rmax = cfg.lowrank_max_rank ? std::min<uint64_t>(cfg.lowrank_max_rank, logp->dim) : most;
        if (e->lr_rmax == 0) e->lr_rmax = 1;
        e->lr_cap = s.num_tune + 2;
        const size_t vb = (size_t)n_chains * (1 + e->lr_rmax) * dpad * sizeof(double);
        const size_t lb = (size_t)n_chains * 2 * e->lr_rmax * sizeof(double);
        const size_t wb = (size_t)n_chains * e->lr_cap * 2 * logp->dim * sizeof(double);
        E_TRY(hipMalloc(&e->d_lrvec, vb));
        E_TRY(hipMemsetAsync(e->d_lrvec, 0, vb, e->stream));
        E_TRY(hipMalloc(&e->d_lrval, lb));
        E_TRY(hipMemsetAsync(e->d_lrval, 0, lb, e->stream));
        E_TRY(hipMalloc(&e->d_lrwin, wb));
    }
    // ziggurat tables of rand_distr's StandardNormal: the fixed ZIG_NORM_X / ZIG_NORM_F constants (zig_tables.hpp), x then f
    E_TRY(copy_on(e->stream, e->d_zig, kZigTables, sizeof kZigTables, hipMemcpyHostToDevice));
    // per-chain scalars: NutsChain::new / GlobalStrategy::new state (the DualAverage is reset on the device)
    {
        std::vector<ChainScalars> sc(n_chains * cl_k);       // cluster mode: every member keeps an identical copy
        for (uint64_t c = 0; c < n_chains * cl_k; ++c) {
            ChainScalars& q = sc[c];
            memset(&q, 0, sizeof q);
            uint8_t key[32];
            nm_chain_rng_key(s.seed, cfg.chain_id_offset + c / cl_k, key);
            for (int i = 0; i < 8; ++i)
                q.key[i] = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) | ((uint32_t)key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
            q.transform_id = -1; q.mm_id = -1; q.stats_last_id = -1;
            q.tuning = 1; q.has_initial_mass_matrix = 1;
            q.current_window_size = s.mass_matrix_switch_freq;
            q.status = NM_CHAIN_OK;
        }
        E_TRY(copy_on(e->stream, e->d_sc, sc.data(), n_chains * cl_k * sizeof(ChainScalars), hipMemcpyHostToDevice));
    }
    KParams& P = e->P;
    memset(&P, 0, sizeof P);
    P.s = s;
    P.n_chains = n_chains; P.dim = logp->dim; P.dpad = dpad; P.chain_id_offset = cfg.chain_id_offset; P.nsslot = nsslot;
    P.pvec = e->d_pvec; P.svec = e->d_svec; P.sc = e->d_sc; P.prof = e->d_prof; P.zig_x = e->d_zig; P.zig_f = e->d_zig + 257; P.logp_params = e->d_params;
    P.early_end = early_end;
    P.cl_k = cl_k; P.cl_slice = CL_SLICE; P.cl_box = e->d_cl_box; P.cl_cnt = e->d_cl_cnt;
    { const char* g = getenv("NM_CLUSTER_GENERAL"); P.cl_general = g && g[0] == '1'; }      // the fallback protocol on request (it is otherwise taken only where a chain's blocks do not share an XCD)
    {   // MclmcChain's switch_draw = (trajectory_switch_fraction * num_tune) as u64 (sampler.rs:441; `as` saturates, NaN -> 0)
        const double q = s.trajectory_switch_fraction * num_tune_f;
        P.mclmc_switch_draw = q >= 18446744073709551616.0 ? ~0ull : (q > 0.0 ? (uint64_t)q : 0ull);
    }
    P.final_step_size_window = s.num_tune >= step_size_window ? s.num_tune - step_size_window : 0;   // saturating_sub
    P.ln_max_step = dlog(s.da_max_step_size);
    if (s.has_jitter) {   // Uniform::new(1 - j, 1 + j)
        const double low = 1.0 - s.jitter, high = 1.0 + s.jitter, max_rand = 1.0 - 2.220446049250313e-16;
        double scale = high - low;
        while (scale * max_rand + low >= high) scale = u2d(d2u(scale) - 1);
        P.jitter_low = low; P.jitter_scale = scale;
    }
    P.x0 = e->d_x0;
    P.lrvec = e->d_lrvec; P.lrval = e->d_lrval; P.lrwin = e->d_lrwin; P.lr_rmax = e->lr_rmax; P.lr_cap = e->lr_cap;
    if (logp->kind == NM_LOGP_HOST_CALLBACK) {
        e->cb_fn = logp->host_fn; e->cb_ctx = logp->host_ctx;
        e->cb_stride = ((sizeof(CbMail) + 2 * logp->dim * sizeof(double)) + 127) / 128 * 128;
        E_TRY(hipHostMalloc((void**)&e->cb_mail, n_chains * e->cb_stride, hipHostMallocMapped | hipHostMallocCoherent));
        memset(e->cb_mail, 0, n_chains * e->cb_stride);
        E_TRY(hipHostGetDevicePointer((void**)&e->cb_mail_dev, e->cb_mail, 0));
        P.cb_mail = e->cb_mail_dev; P.cb_stride = e->cb_stride;
        unsigned nt = (unsigned)(logp->host_threads ? logp->host_threads : std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u));
        if (nt > n_chains) nt = (unsigned)n_chains;
        for (unsigned t = 0; t < nt; ++t)
            e->cb_threads.emplace_back([e, t, nt]() {
                // thread t answers the mailboxes of chains t, t + nt, ...: a chain is never evaluated by two threads at once
                std::vector<double> x(e->dim), g(e->dim);
                unsigned idle = 0;
                while (!e->cb_stop.load(std::memory_order_relaxed)) {
                    if (!e->cb_active.load(std::memory_order_acquire)) { std::this_thread::sleep_for(std::chrono::microseconds(200)); continue; }
                    bool any = false;
                    for (uint64_t c = t; c < e->n_chains; c += nt) {
                        CbMail* m = reinterpret_cast<CbMail*>(e->cb_mail + c * e->cb_stride);
                        const uint64_t req = __atomic_load_n(&m->req, __ATOMIC_ACQUIRE);
                        if (req == __atomic_load_n(&m->resp, __ATOMIC_RELAXED)) continue;
                        any = true;
                        double* mx = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(m) + sizeof(CbMail));
                        double* mg = mx + e->dim;
                        memcpy(x.data(), mx, e->dim * sizeof(double));
                        double lp = 0.0;
                        int st = e->cb_fn(e->cb_ctx, e->cfg.chain_id_offset + c, e->dim, x.data(), g.data(), &lp);
                        if (st < 0 || st > 2) st = 2;
                        memcpy(mg, g.data(), e->dim * sizeof(double));
                        m->logp = lp; m->status = st;
                        __atomic_store_n(&m->resp, req, __ATOMIC_RELEASE);
                        e->cb_calls.fetch_add(1, std::memory_order_relaxed);
                    }
                    if (any) idle = 0;
                    else if (++idle > 64) std::this_thread::yield();
                }
            });
    }
    E_TRY(hipStreamSynchronize(e->stream));
#undef E_TRY
    *out = e;
    return NM_OK;
}

static nm_status collect_timing(nm_engine* e) {
    if (e->pending_timing) {
        HIP_TRY(hipEventSynchronize(e->ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
        e->kernel_ms += (double)ms;
        e->pending_timing = false;
    }
    return NM_OK;
}

extern "C" nm_status nm_engine_set_positions(nm_engine* e, const double* h_x0, uint64_t* h_chain_status) {
    return nm_engine_set_positions_masked(e, h_x0, nullptr, h_chain_status);
}
extern "C" nm_status nm_engine_init_positions_retry(nm_engine* e, const double* h_x0, uint64_t max_tries, uint64_t* h_chain_status, uint64_t* h_tries) {
    if (!e) return fail(NM_ERR_INVALID_ARG, "null engine");
    if (max_tries == 0) max_tries = 500;
    const uint64_t n = e->n_chains, dim = e->dim;
    std::vector<double> x0(n * dim);
    std::vector<uint64_t> status(n, NM_CHAIN_BAD_INIT), tries(n, 0);
    std::vector<uint8_t> mask(n, 1);
    nm_status st = NM_OK;
    for (uint64_t attempt = 0; attempt < max_tries; ++attempt) {
        if (attempt == 0 && h_x0) memcpy(x0.data(), h_x0, n * dim * sizeof(double));
        else nm_init_positions_uniform_at(e->s.seed, e->cfg.chain_id_offset, n, dim, h_x0 ? attempt - 1 : attempt, x0.data());
        std::vector<uint64_t> now(n);
        st = nm_engine_set_positions_masked(e, x0.data(), mask.data(), now.data());
        if (st != NM_OK && st != NM_ERR_BAD_INIT) break;                 // a fatal logp error ends the loop like the reference's `?`
        uint64_t left = 0;
        for (uint64_t c = 0; c < n; ++c) {
            if (!mask[c]) continue;
            status[c] = now[c]; tries[c] = attempt + 1;
            mask[c] = now[c] == NM_CHAIN_BAD_INIT ? 1 : 0;
            left += mask[c];
        }
        if (!left) { st = NM_OK; break; }
    }
    if (h_chain_status) memcpy(h_chain_status, status.data(), n * sizeof(uint64_t));
    if (h_tries) memcpy(h_tries, tries.data(), n * sizeof(uint64_t));
    return st;
}
extern "C" nm_status nm_engine_set_positions_masked(nm_engine* e, const double* h_x0, const uint8_t* h_mask, uint64_t* h_chain_status) {
    if (!e || !h_x0) return fail(NM_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(e->d_x0, h_x0, e->n_chains * e->dim * sizeof(double), hipMemcpyHostToDevice, e->stream));
    KParams P = e->P;
    if (h_mask) {
        if (!e->d_init_mask) HIP_TRY(hipMalloc(&e->d_init_mask, e->n_chains));
        HIP_TRY(hipMemcpyAsync(e->d_init_mask, h_mask, e->n_chains, hipMemcpyHostToDevice, e->stream));
        P.init_mask = e->d_init_mask;
    }
    e->cb_active.store(1, std::memory_order_release);
    if (e->cl_k > 1) {      // the clusters' arrival counters, and their mailboxes (the tags of the last launch must not match this one's)
        HIP_TRY(hipMemsetAsync(e->d_cl_cnt, 0, e->n_clusters * sizeof(unsigned long long), e->stream));
        HIP_TRY(hipMemsetAsync(e->d_cl_box, 0, e->n_clusters * CL_BOX_WORDS * e->cl_k * RED_MAX_VALUES * sizeof(unsigned long long), e->stream));
    }
    HIP_TRY(launch(e->logp_kind, e->dpl, e->wpc, K_INIT, P, e->n_waves, e->stream, nullptr, e->module_launch, e->variant));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->cb_active.store(0, std::memory_order_release);
    std::vector<ChainScalars> sc(e->n_chains);
    HIP_TRY(read_scalars(e, sc.data()));
    uint64_t bad = 0, fatal = 0, min_draws = ~0ull;
    for (uint64_t c = 0; c < e->n_chains; ++c) {
        if (h_chain_status) h_chain_status[c] = sc[c].status;
        if (sc[c].status == NM_CHAIN_BAD_INIT) bad++;
        else if (sc[c].status != NM_CHAIN_OK) fatal++;
        else if (sc[c].draw_count < min_draws) min_draws = sc[c].draw_count;
    }
    // the draw index of the chain that is furthest behind decides whether a launch still needs the warm-up kernel (a chain
    // whose first set_position failed may start later than the others)
    e->draws_launched = min_draws == ~0ull ? 0 : min_draws;
    e->positioned = true;
    if (fatal) return fail(NM_ERR_LOGP_FAILURE, "%llu chain(s): logp failure during set_position", (unsigned long long)fatal);
    if (bad) return fail(NM_ERR_BAD_INIT, "%llu chain(s): Could not initialize state because of bad initial gradient", (unsigned long long)bad);
    return NM_OK;
}

// ---------------------------------------------------------------------------------------------
// low-rank transformation: uploads and the estimator rounds between launches
// ---------------------------------------------------------------------------------------------
// One chain's LowRankMassMatrix::update input -> device (sigma, 1/sigma, mean into the chain's P slots; mu_lr and the
// eigenvectors into lrvec; lambda^(+-1/2) into lrval) and the scalars the kernel commits it with.  Returns false (and
// leaves the device untouched) when the reference's update() would return early on non-finite input.
static bool lr_stage_update(nm_engine* e, uint64_t c, ChainScalars& q, uint64_t n_eig, const double* stds, const double* mean,
                            const double* vals, const double* vecs /*[n_eig][dim]*/, const double* mu_lr, hipError_t* err) {
    const uint64_t dim = e->dim, dpad = e->P.dpad;
    auto finite = [](const double* a, uint64_t n) { for (uint64_t i = 0; i < n; ++i) if (!std::isfinite(a[i])) return false; return true; };
    *err = hipSuccess;
    q.lr_upd_ok = 0; q.lr_upd_rank = 0; q.lr_upd_logdet = 0.0;
    if (!finite(stds, dim) || !finite(mean, dim) || !finite(vals, n_eig) || !finite(vecs, n_eig * dim)) return false;
    std::vector<double> isig(dim), vs(n_eig ? n_eig : 1), vi(n_eig ? n_eig : 1);
    for (uint64_t i = 0; i < dim; ++i) isig[i] = 1.0 / stds[i];                       // array_recip (diagonal.rs:158)
    double ld = -0.0;                                                                  // InnerMatrix::new (low_rank.rs:55-92)
    for (uint64_t k = 0; k < n_eig; ++k) { ld += -0.5 * dlog(vals[k]); vs[k] = std::sqrt(vals[k]); vi[k] = 1.0 / vs[k]; }
    double* pv = e->d_pvec + (size_t)c * NUM_PSLOT * dpad;
    double* lv = e->d_lrvec + (size_t)c * (1 + e->lr_rmax) * dpad;
    double* lw = e->d_lrval + (size_t)c * 2 * e->lr_rmax;
    hipError_t er = copy_on(e->stream, pv + (size_t)P_SIG * dpad, stds, dim * 8, hipMemcpyHostToDevice);
    if (er == hipSuccess) er = copy_on(e->stream, pv + (size_t)P_ISIG * dpad, isig.data(), dim * 8, hipMemcpyHostToDevice);
    if (er == hipSuccess) er = copy_on(e->stream, pv + (size_t)P_MU * dpad, mean, dim * 8, hipMemcpyHostToDevice);
    if (er == hipSuccess) er = copy_on(e->stream, lv, mu_lr, dim * 8, hipMemcpyHostToDevice);
    if (er == hipSuccess && n_eig) er = copy2d_on(e->stream, lv + dpad, dpad * 8, vecs, dim * 8, dim * 8, n_eig, hipMemcpyHostToDevice);
    if (er == hipSuccess && n_eig) er = copy_on(e->stream, lw, vs.data(), n_eig * 8, hipMemcpyHostToDevice);
    if (er == hipSuccess && n_eig) er = copy_on(e->stream, lw + e->lr_rmax, vi.data(), n_eig * 8, hipMemcpyHostToDevice);
    *err = er;
    if (er != hipSuccess) return false;
    q.lr_upd_ok = 1; q.lr_upd_rank = n_eig; q.lr_upd_logdet = ld;
    return true;
}

// One transformation for ALL chains: staged once on the device, then copied into every chain's slots by a kernel (per chain:
// sigma, 1/sigma, mean -> P_SIG, P_ISIG, P_MU; mu_lr and the eigenvectors -> lrvec; lambda^(+-1/2) -> lrval; the scalars of the
// pending LowRankMassMatrix::update -> ChainScalars).  stage: [4 + n_eig][dpad] rows (sigma, 1/sigma, mean, mu_lr, vecs...).
__global__ __launch_bounds__(256) void lr_broadcast_kernel(const KParams P, const double* stage, const double* vals2, uint64_t n_eig,
                                                          double logdet, int ok) {
    const uint64_t dp = P.dpad;
    for (uint64_t c = blockIdx.x; c < P.n_chains; c += gridDim.x) {
        ChainScalars& q = P.sc[c];
        if (ok) {
            double* pv = P.pvec + (size_t)c * NUM_PSLOT * dp;
            double* lv = P.lrvec + (size_t)c * (1 + P.lr_rmax) * dp;
            double* lw = P.lrval + (size_t)c * 2 * P.lr_rmax;
            for (uint64_t i = threadIdx.x; i < dp; i += blockDim.x) {
                pv[(size_t)P_SIG * dp + i] = stage[i];
                pv[(size_t)P_ISIG * dp + i] = stage[dp + i];
                pv[(size_t)P_MU * dp + i] = stage[2 * dp + i];
                lv[i] = stage[3 * dp + i];
            }
            for (uint64_t i = threadIdx.x; i < n_eig * dp; i += blockDim.x) lv[dp + i] = stage[4 * dp + i];
            for (uint64_t i = threadIdx.x; i < n_eig; i += blockDim.x) { lw[i] = vals2[i]; lw[P.lr_rmax + i] = vals2[n_eig + i]; }
        }
        if (threadIdx.x == 0) {
            q.lr_upd_ok = ok ? 1 : 0; q.lr_upd_rank = ok ? n_eig : 0; q.lr_upd_logdet = ok ? logdet : 0.0;
            q.lr_pending = LR_SET_TRANSFORM;
        }
    }
}

// host threads worth starting: the logical CPUs, capped by the cgroup CPU quota (a container may show 256 CPUs and be allowed 16
// CPUs' worth of time: more runnable threads than that are only throttled)
static unsigned usable_host_threads() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        double per = 0.0;
        if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0.0) {
            const double cpus = atof(q) / per;
            if (cpus >= 1.0 && cpus < (double)n) n = (unsigned)(cpus + 0.5);
        }
        fclose(f);
    }
    return n;
}

// An estimator round's updates, packed on the host: per pending chain `rows` = 4 + rmax vectors of dpad doubles (sigma, 1/sigma,
// mean, mu_lr, eigenvectors) and 2 x rmax eigenvalue scales; meta[i] = {chain, n_eig, ok}.  One block per pending chain.
__global__ __launch_bounds__(256) void lr_scatter_kernel(const KParams P, const uint64_t* meta, const double* stage, const double* vals2,
                                                        uint64_t rmax, uint64_t n) {
    const uint64_t dp = P.dpad, rows = 4 + rmax;
    for (uint64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint64_t c = meta[3 * i], n_eig = meta[3 * i + 1];
        if (!meta[3 * i + 2]) continue;
        const double* st = stage + (size_t)i * rows * dp;
        double* pv = P.pvec + (size_t)c * NUM_PSLOT * dp;
        double* lv = P.lrvec + (size_t)c * (1 + P.lr_rmax) * dp;
        double* lw = P.lrval + (size_t)c * 2 * P.lr_rmax;
        for (uint64_t j = threadIdx.x; j < dp; j += blockDim.x) {
            pv[(size_t)P_SIG * dp + j] = st[j];
            pv[(size_t)P_ISIG * dp + j] = st[dp + j];
            pv[(size_t)P_MU * dp + j] = st[2 * dp + j];
            lv[j] = st[3 * dp + j];
        }
        for (uint64_t j = threadIdx.x; j < n_eig * dp; j += blockDim.x) lv[dp + j] = st[4 * dp + j];
        for (uint64_t j = threadIdx.x; j < n_eig; j += blockDim.x) { lw[j] = vals2[i * 2 * rmax + j]; lw[P.lr_rmax + j] = vals2[i * 2 * rmax + rmax + j]; }
    }
}
// lr_stage_update's checks and derived values, written into a packed row block instead of the device
static bool lr_pack_update(nm_engine* e, ChainScalars& q, uint64_t n_eig, const double* stds, const double* mean, const double* vals,
                           const double* vecs, const double* mu_lr, double* st /*[4 + n_eig][dpad]*/, double* v2 /*[2][rmax]*/, uint64_t rmax) {
    const uint64_t dim = e->dim, dp = e->P.dpad;
    auto finite = [](const double* a, uint64_t n) { for (uint64_t i = 0; i < n; ++i) if (!std::isfinite(a[i])) return false; return true; };
    q.lr_upd_ok = 0; q.lr_upd_rank = 0; q.lr_upd_logdet = 0.0;
    if (!finite(stds, dim) || !finite(mean, dim) || !finite(vals, n_eig) || !finite(vecs, n_eig * dim)) return false;
    memset(st, 0, (4 + n_eig) * dp * sizeof(double));
    for (uint64_t i = 0; i < dim; ++i) { st[i] = stds[i]; st[dp + i] = 1.0 / stds[i]; st[2 * dp + i] = mean[i]; st[3 * dp + i] = mu_lr[i]; }
    for (uint64_t k = 0; k < n_eig; ++k) memcpy(st + (4 + k) * dp, vecs + k * dim, dim * sizeof(double));
    double ld = -0.0;                                                                  // InnerMatrix::new (low_rank.rs:55-92)
    for (uint64_t k = 0; k < n_eig; ++k) { ld += -0.5 * dlog(vals[k]); v2[k] = std::sqrt(vals[k]); v2[rmax + k] = 1.0 / v2[k]; }
    q.lr_upd_ok = 1; q.lr_upd_rank = n_eig; q.lr_upd_logdet = ld;
    return true;
}
template <class T>
static hipError_t grow(T** p, size_t* have, size_t want, bool pinned_host) {
    if (*have >= want) return hipSuccess;
    if (*p) { if (pinned_host) (void)hipHostFree(*p); else (void)hipFree(*p); *p = nullptr; *have = 0; }
    const size_t sz = want + want / 4;
    hipError_t er = pinned_host ? hipHostMalloc((void**)p, sz, hipHostMallocDefault) : hipMalloc((void**)p, sz);
    if (er == hipSuccess) *have = sz;
    return er;
}

extern "C" nm_status nm_engine_set_lowrank_estimator(nm_engine* e, nm_lowrank_estimator_fn fn, void* ctx, uint64_t n_threads) {
    if (!e) return fail(NM_ERR_INVALID_ARG, "null engine");
    if (!e->lr) return fail(NM_ERR_STATE, "the engine was not created with adaptation = NM_ADAPT_LOW_RANK");
    e->lr_estimator = fn ? fn : nm_lowrank_compute_update;
    e->lr_estimator_ctx = fn ? ctx : nullptr;
    e->lr_threads = n_threads;
    return NM_OK;
}
extern "C" uint64_t nm_engine_lowrank_max_rank(const nm_engine* e) { return e ? e->lr_rmax : 0; }
// Development aid (not part of include/nuts_amd.h): seconds spent in the estimator rounds so far: total, window download,
// estimator threads, upload + scatter; rounds; estimator calls
extern "C" void nm_debug_lowrank_timing(const nm_engine* e, double out[6]) {
    out[0] = e->lr_host_seconds; out[1] = e->lr_download_seconds; out[2] = e->lr_estimator_seconds; out[3] = e->lr_upload_seconds;
    out[4] = (double)e->lr_rounds; out[5] = (double)e->lr_updates;
}

extern "C" nm_status nm_engine_set_transform(nm_engine* e, uint64_t per_chain, uint64_t n_eig, const double* h_stds, const double* h_mean,
                                             const double* h_vals, const double* h_vecs, const double* h_mu_lr) {
    if (!e || !h_stds || !h_mean || !h_mu_lr || (n_eig && (!h_vals || !h_vecs))) return fail(NM_ERR_INVALID_ARG, "null argument");
    if (!e->lr) return fail(NM_ERR_STATE, "nm_engine_set_transform needs settings.adaptation == NM_ADAPT_LOW_RANK");
    if (!e->positioned) return fail(NM_ERR_STATE, "nm_engine_set_transform before nm_engine_set_positions");
    if (n_eig > e->lr_rmax) return fail(NM_ERR_UNSUPPORTED, "%llu eigenvectors > lowrank_max_rank %llu", (unsigned long long)n_eig, (unsigned long long)e->lr_rmax);
    HIP_TRY(hipSetDevice(e->device));
    nm_status st = nm_engine_synchronize(e);
    if (st != NM_OK) return st;
    std::vector<ChainScalars> sc(e->n_chains);
    HIP_TRY(read_scalars(e, sc.data()));
    const uint64_t dim = e->dim;
    for (uint64_t c = 0; c < e->n_chains; ++c)
        if (sc[c].lr_pending != LR_IDLE && sc[c].lr_pending != LR_SET_TRANSFORM) return fail(NM_ERR_STATE, "chain %llu is waiting for its estimator", (unsigned long long)c);
    bool shared_ok = false;
    if (!per_chain) {      // one upload + a broadcast kernel instead of several copies per chain
        const uint64_t dp = e->P.dpad;
        auto finite = [](const double* a, uint64_t n) { for (uint64_t i = 0; i < n; ++i) if (!std::isfinite(a[i])) return false; return true; };
        shared_ok = finite(h_stds, dim) && finite(h_mean, dim) && (!n_eig || (finite(h_vals, n_eig) && finite(h_vecs, n_eig * dim)));
        std::vector<double> stage((4 + n_eig) * dp, 0.0), vals2(2 * (n_eig ? n_eig : 1), 0.0);
        double ld = -0.0;
        if (shared_ok) {
            for (uint64_t i = 0; i < dim; ++i) { stage[i] = h_stds[i]; stage[dp + i] = 1.0 / h_stds[i]; stage[2 * dp + i] = h_mean[i]; stage[3 * dp + i] = h_mu_lr[i]; }
            for (uint64_t k = 0; k < n_eig; ++k) {
                memcpy(&stage[(4 + k) * dp], h_vecs + k * dim, dim * 8);
                ld += -0.5 * dlog(h_vals[k]);
                vals2[k] = std::sqrt(h_vals[k]); vals2[n_eig + k] = 1.0 / vals2[k];
            }
        }
        double *d_stage = nullptr, *d_vals2 = nullptr;
        HIP_TRY(hipMalloc(&d_stage, stage.size() * 8));
        HIP_TRY(hipMalloc(&d_vals2, vals2.size() * 8));
        hipError_t er = hipMemcpyAsync(d_stage, stage.data(), stage.size() * 8, hipMemcpyHostToDevice, e->stream);
        if (er == hipSuccess) er = hipMemcpyAsync(d_vals2, vals2.data(), vals2.size() * 8, hipMemcpyHostToDevice, e->stream);
        if (er == hipSuccess) {
            hipLaunchKernelGGL(lr_broadcast_kernel, dim3((unsigned)std::min<uint64_t>(e->n_chains, 4096)), dim3(256), 0, e->stream, e->P, d_stage, d_vals2,
                               n_eig, ld, shared_ok ? 1 : 0);
            er = hipGetLastError();
        }
        if (er == hipSuccess) er = hipStreamSynchronize(e->stream);
        (void)hipFree(d_stage); (void)hipFree(d_vals2);
        if (er != hipSuccess) return fail(NM_ERR_HIP, "broadcast of the transformation: %s", hipGetErrorString(er));
        HIP_TRY(read_scalars(e, sc.data()));
    }
    for (uint64_t c = 0; per_chain && c < e->n_chains; ++c) {
        const uint64_t k = per_chain ? c : 0;
        hipError_t er;
        (void)lr_stage_update(e, c, sc[c], n_eig, h_stds + k * dim, h_mean + k * dim, h_vals ? h_vals + k * n_eig : nullptr,
                              h_vecs ? h_vecs + k * n_eig * dim : nullptr, h_mu_lr + k * dim, &er);
        if (er != hipSuccess) return fail(NM_ERR_HIP, "upload of the transformation: %s", hipGetErrorString(er));
        sc[c].lr_pending = LR_SET_TRANSFORM;            // committed by the next launch (LowRankMassMatrix::update)
    }
    if (per_chain) HIP_TRY(copy_on(e->stream, e->d_sc, sc.data(), e->n_chains * sizeof(ChainScalars), hipMemcpyHostToDevice));
    // One transformation for all chains, frozen, on the full-precision normal: the draws can run 16 chains per block with
    // U', U and P on the matrix cores (nuts_tile.hpp).  The matrices are packed in MFMA operand order once, here.
    e->tile_active = false;
    if (!per_chain && e->s.freeze_transform && e->s.trajectory_kind == NM_TRAJ_EUCLIDEAN && e->s.sampler == NM_SAMPLER_NUTS && e->logp_kind == NM_LOGP_MVN_PREC && e->cfg.chain_tiles != 1 && e->wpc == 1 &&
        (e->dpl == 2 || e->dpl == 4) && dim <= 256 && n_eig >= 8 && n_eig <= 256 && dim % 8 == 0 && n_eig % 8 == 0 && !e->h_params.empty() &&
        sc[0].lr_upd_ok) {
        auto pack = [](uint64_t R, uint64_t K, auto&& elem) {      // [stripes][kpairs][64 lanes][2]
            const uint64_t st = (R + 15) / 16, kp = (K + 7) / 8;
            std::vector<double> out(st * kp * 128, 0.0);
            for (uint64_t s_ = 0; s_ < st; ++s_)
                for (uint64_t q = 0; q < kp; ++q)
                    for (uint64_t l = 0; l < 64; ++l)
                        for (uint64_t j = 0; j < 2; ++j) {
                            const uint64_t row = 16 * s_ + (l & 15), k = 8 * q + 4 * j + (l >> 4);
                            out[((s_ * kp + q) * 64 + l) * 2 + j] = (row < R && k < K) ? elem(row, k) : 0.0;
                        }
            return out;
        };
        const double* P_ = e->h_params.data();
        const std::vector<double> ut = pack(n_eig, dim, [&](uint64_t k, uint64_t d) { return h_vecs[k * dim + d]; });
        const std::vector<double> u = pack(dim, n_eig, [&](uint64_t d, uint64_t k) { return h_vecs[k * dim + d]; });
        const std::vector<double> pp = pack(dim, dim, [&](uint64_t d, uint64_t j) { return P_[j * dim + d]; });
        auto up = [&](double** dst, const std::vector<double>& src) -> hipError_t {
            if (*dst) (void)hipFree(*dst);
            *dst = nullptr;
            hipError_t er = hipMalloc(dst, src.size() * 8);
            return er != hipSuccess ? er : copy_on(e->stream, *dst, src.data(), src.size() * 8, hipMemcpyHostToDevice);
        };
        HIP_TRY(up(&e->d_tile_ut, ut)); HIP_TRY(up(&e->d_tile_u, u)); HIP_TRY(up(&e->d_tile_p, pp));
        e->tile_mats = {e->d_tile_ut, e->d_tile_u, e->d_tile_p, (int)dim, (int)n_eig, (int)((dim + 7) / 8), (int)((n_eig + 7) / 8),
                        (int)((dim + 15) / 16), (int)((n_eig + 15) / 16)};
        e->tile_active = true;
    }
    return NM_OK;
}
extern "C" uint64_t nm_engine_tile_launches(const nm_engine* e) { return e ? e->tile_launches : 0; }
extern "C" uint64_t nm_engine_host_logp_calls(const nm_engine* e) { return e ? e->cb_calls.load() : 0; }

extern "C" nm_status nm_engine_get_lowrank(nm_engine* e, uint64_t* h_n_eig, double* h_vals_sqrt, double* h_vecs, double* h_mu_lr) {
    if (!e) return fail(NM_ERR_INVALID_ARG, "null engine");
    if (!e->lr) return fail(NM_ERR_STATE, "the engine was not created with adaptation = NM_ADAPT_LOW_RANK");
    nm_status st = nm_engine_synchronize(e);
    if (st != NM_OK) return st;
    std::vector<ChainScalars> sc(e->n_chains);
    HIP_TRY(read_scalars(e, sc.data()));
    const uint64_t dim = e->dim, dpad = e->P.dpad, R = e->lr_rmax;
    for (uint64_t c = 0; c < e->n_chains; ++c) {
        if (h_n_eig) h_n_eig[c] = sc[c].lr_has_inner ? sc[c].lr_rank : 0;
        const double* lv = e->d_lrvec + (size_t)c * (1 + R) * dpad;
        if (h_mu_lr) HIP_TRY(copy_on(e->stream, h_mu_lr + c * dim, lv, dim * 8, hipMemcpyDeviceToHost));
        if (h_vecs) HIP_TRY(copy2d_on(e->stream, h_vecs + c * R * dim, dim * 8, lv + dpad, dpad * 8, dim * 8, R, hipMemcpyDeviceToHost));
        if (h_vals_sqrt) HIP_TRY(copy_on(e->stream, h_vals_sqrt + c * R, e->d_lrval + (size_t)c * 2 * R, R * 8, hipMemcpyDeviceToHost));
    }
    return NM_OK;
}

// The draw loop of an NM_ADAPT_LOW_RANK engine: launch; chains whose schedule asks for a new matrix pause inside
// GlobalStrategy::adapt; their windows go through the estimator on host threads; the answers are uploaded; relaunch.
static nm_status lr_draw(nm_engine* e, uint64_t n_draws, const KParams& P_in) {
    KParams P = P_in;
    P.row_base = e->draws_launched;
    P.draw_end = e->draws_launched + n_draws;
    const uint64_t dim = e->dim, nc = e->n_chains;
    std::vector<ChainScalars> sc(nc);
    for (;;) {
        HIP_TRY(hipEventRecord(e->ev0, e->stream));
        if (e->tile_active) {
            HIP_TRY(launch_tile_mvn_prec(e->dpl, 0, P, e->tile_mats, e->tile_grid, e->stream, nullptr));
            e->tile_launches += 1;
        } else
            HIP_TRY(launch(e->logp_kind, e->dpl, e->wpc, K_DRAW, P, e->n_waves, e->stream, nullptr, e->module_launch, true));
        HIP_TRY(hipEventRecord(e->ev1, e->stream));
        e->pending_timing = true;
        e->kernel_launches += 1;
        HIP_TRY(hipStreamSynchronize(e->stream));
        nm_status st = collect_timing(e);
        if (st != NM_OK) return st;
        HIP_TRY(copy_on(e->stream, sc.data(), e->d_sc, nc * sizeof(ChainScalars), hipMemcpyDeviceToHost));
        std::vector<uint64_t> pend;
        for (uint64_t c = 0; c < nc; ++c)
            if (sc[c].status == NM_CHAIN_OK && sc[c].lr_pending == LR_WAIT_HOST) pend.push_back(c);
        if (pend.empty()) break;
        const auto t0 = std::chrono::steady_clock::now();
        unsigned nt = (unsigned)(e->lr_threads ? e->lr_threads : usable_host_threads());
        if (nt == 0) nt = 1;
        if (nt > pend.size()) nt = (unsigned)pend.size();
        // 1. every pending chain's window -> pinned host memory: a batch of asynchronous copies, one wait
        const size_t np = pend.size();
        std::vector<size_t> woff(np + 1, 0);
        uint64_t rmax = 1;
        for (size_t i = 0; i < np; ++i) {
            const uint64_t n = sc[pend[i]].lr_len;
            woff[i + 1] = woff[i] + (size_t)n * 2 * dim;
            rmax = std::max<uint64_t>(rmax, std::min<uint64_t>(dim, 2 * n));
        }
        rmax = std::min<uint64_t>(rmax, e->lr_rmax);
        const uint64_t dp = e->P.dpad, rows = 4 + rmax;
        HIP_TRY(grow(&e->h_lr_win, &e->h_lr_win_bytes, woff[np] * sizeof(double), true));
        HIP_TRY(grow(&e->h_lr_upd, &e->h_lr_upd_bytes, np * (rows * dp + 2 * rmax) * sizeof(double), true));
        HIP_TRY(grow(&e->d_lr_upd, &e->d_lr_upd_bytes, np * (rows * dp + 2 * rmax) * sizeof(double), false));
        HIP_TRY(grow(&e->d_lr_meta, &e->d_lr_meta_bytes, np * 3 * sizeof(uint64_t), false));
        for (size_t i = 0; i < np; ++i) {
            const ChainScalars& q = sc[pend[i]];
            HIP_TRY(hipMemcpyAsync(e->h_lr_win + woff[i], e->d_lrwin + ((size_t)pend[i] * e->lr_cap + q.lr_start) * 2 * dim,
                                   (size_t)q.lr_len * 2 * dim * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        }
        HIP_TRY(hipStreamSynchronize(e->stream));
        const auto t1 = std::chrono::steady_clock::now();
        // 2. the estimator, one chain per task, on the host threads; results packed for one upload
        std::vector<uint64_t> meta(np * 3, 0);
        double* const upd_vec = e->h_lr_upd;
        double* const upd_val = e->h_lr_upd + np * rows * dp;
        std::atomic<size_t> next{0};
        std::atomic<int> too_wide{0};
        auto work = [&]() {
            std::vector<double> draws, grads, stds(dim), mean(dim), mu(dim), vals, vecs;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= np) break;
                const uint64_t c = pend[i];
                ChainScalars& q = sc[c];
                const uint64_t n = q.lr_len;
                const double* win = e->h_lr_win + woff[i];
                draws.resize(n * dim); grads.resize(n * dim);
                for (uint64_t r = 0; r < n; ++r) {
                    memcpy(&draws[r * dim], &win[(2 * r) * dim], dim * 8);
                    memcpy(&grads[r * dim], &win[(2 * r + 1) * dim], dim * 8);
                }
                const uint64_t most = std::min<uint64_t>(dim, 2 * n);
                vals.assign(most, 0.0); vecs.assign(most * dim, 0.0);
                uint64_t n_eig = 0;
                const int rc = e->lr_estimator(e->lr_estimator_ctx, dim, n, draws.data(), grads.data(), e->s.lr_gamma,
                                               e->s.lr_eigval_cutoff, stds.data(), mean.data(), &n_eig, vals.data(), vecs.data(), mu.data());
                q.lr_upd_ok = 0; q.lr_upd_rank = 0; q.lr_upd_logdet = 0.0;
                meta[3 * i] = c;
                if (rc == 0) {
                    if (n_eig > rmax) { too_wide++; continue; }
                    if (lr_pack_update(e, q, n_eig, stds.data(), mean.data(), vals.data(), vecs.data(), mu.data(),
                                       upd_vec + i * rows * dp, upd_val + i * 2 * rmax, rmax)) { meta[3 * i + 1] = n_eig; meta[3 * i + 2] = 1; }
                }
                q.lr_pending = LR_ANSWERED;
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 0; t + 1 < nt; ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
        const auto t2 = std::chrono::steady_clock::now();
        e->lr_updates += np;
        e->lr_rounds += 1;
        if (too_wide.load()) return fail(NM_ERR_UNSUPPORTED, "the estimator returned more eigenvectors than lowrank_max_rank %llu", (unsigned long long)e->lr_rmax);
        // 3. one upload, one scatter into the chains' slots
        HIP_TRY(hipMemcpyAsync(e->d_lr_upd, e->h_lr_upd, np * (rows * dp + 2 * rmax) * sizeof(double), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->d_lr_meta, meta.data(), np * 3 * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(lr_scatter_kernel, dim3((unsigned)std::min<size_t>(np, 2048)), dim3(256), 0, e->stream, P, (const uint64_t*)e->d_lr_meta,
                           (const double*)e->d_lr_upd, (const double*)(e->d_lr_upd + np * rows * dp), rmax, (uint64_t)np);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(e->stream));
        const auto t3 = std::chrono::steady_clock::now();
        e->lr_host_seconds += std::chrono::duration<double>(t3 - t0).count();
        e->lr_download_seconds += std::chrono::duration<double>(t1 - t0).count();
        e->lr_estimator_seconds += std::chrono::duration<double>(t2 - t1).count();
        e->lr_upload_seconds += std::chrono::duration<double>(t3 - t2).count();
        HIP_TRY(copy_on(e->stream, e->d_sc, sc.data(), nc * sizeof(ChainScalars), hipMemcpyHostToDevice));
    }
    e->draws_total += n_draws;
    e->draws_launched += n_draws;
    return NM_OK;
}

extern "C" nm_status nm_engine_draw_ex_async(nm_engine* e, uint64_t n_draws, const nm_draw_outputs* out) {
    if (!e) return fail(NM_ERR_INVALID_ARG, "null engine");
    if (!out) return fail(NM_ERR_INVALID_ARG, "null nm_draw_outputs");
    if (!e->positioned) return fail(NM_ERR_STATE, "nm_engine_draw before nm_engine_set_positions");
    if (n_draws == 0) return NM_OK;
    HIP_TRY(hipSetDevice(e->device));
    nm_status st = collect_timing(e);
    if (st != NM_OK) return st;
    KParams P = e->P;
    P.n_draws = n_draws; P.out_positions = out->d_positions; P.out_stats = out->d_stats;
    P.out_gradient = out->d_gradient;
    P.out_tpos = out->d_transformed_position; P.out_tgrad = out->d_transformed_gradient;
    P.out_mm_inv = out->d_mass_matrix_inv; P.out_mm_mu = out->d_transformation_mu;
    P.out_div_start = out->d_divergence_start; P.out_div_start_grad = out->d_divergence_start_gradient;
    P.out_div_end = out->d_divergence_end;
    P.out_mm_eigvals = out->d_mass_matrix_eigvals;
    e->cb_active.store(1, std::memory_order_release);   // NM_LOGP_HOST_CALLBACK: the service threads answer until the next synchronize
    if (e->lr) return lr_draw(e, n_draws, P);           // synchronous: the estimator rounds need the host between launches
    if (e->cl_k > 1) {
        HIP_TRY(hipMemsetAsync(e->d_cl_cnt, 0, e->n_clusters * sizeof(unsigned long long), e->stream));
        HIP_TRY(hipMemsetAsync(e->d_cl_box, 0, e->n_clusters * CL_BOX_WORDS * e->cl_k * RED_MAX_VALUES * sizeof(unsigned long long), e->stream));
    }
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    // small chains, many of them: the several-chains-per-wavefront kernels compute the same draws and statistics
    if (e->group_grid) {
        // a launch that starts inside the warm-up takes the kernel with the adaptation compiled in
        HIP_TRY(launch(e->logp_kind, e->dpl, e->wpc, e->draws_launched < e->s.num_tune ? K_GROUP_TUNE : K_GROUP_DRAW, P, e->group_grid, e->stream, nullptr, e->module_launch));
        e->group_launches += 1;
    } else if (e->tile_diag_active) {
        HIP_TRY(launch_tile_mvn_diag(e->dpl, P, e->tile_mats, e->tile_grid, e->stream));
        e->tile_launches += 1;
    } else
        HIP_TRY(launch(e->logp_kind, e->dpl, e->wpc, K_DRAW, P, e->n_waves, e->stream, nullptr, e->module_launch, e->variant));
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    e->pending_timing = true;
    e->kernel_launches += 1;
    e->draws_total += n_draws;
    e->draws_launched += n_draws;
    return NM_OK;
}
extern "C" nm_status nm_engine_synchronize(nm_engine* e) {
    if (!e) return fail(NM_ERR_INVALID_ARG, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->cb_active.store(0, std::memory_order_release);
    return collect_timing(e);
}
extern "C" nm_status nm_engine_draw_async(nm_engine* e, uint64_t n_draws, double* d_positions, nm_draw_stats* d_stats) {
    nm_draw_outputs out = {};
    out.d_positions = d_positions; out.d_stats = d_stats;
    return nm_engine_draw_ex_async(e, n_draws, &out);
}
extern "C" nm_status nm_engine_draw_ex(nm_engine* e, uint64_t n_draws, const nm_draw_outputs* out) {
    nm_status st = nm_engine_draw_ex_async(e, n_draws, out);
    if (st != NM_OK) return st;
    return nm_engine_synchronize(e);
}
extern "C" nm_status nm_engine_draw(nm_engine* e, uint64_t n_draws, double* d_positions, nm_draw_stats* d_stats) {
    nm_status st = nm_engine_draw_async(e, n_draws, d_positions, d_stats);
    if (st != NM_OK) return st;
    return nm_engine_synchronize(e);
}

// Draws delivered to HOST memory: the launch is cut into chunks of draws that go through two sets of device staging buffers —
// while chunk i travels to the host on the copy stream, the kernel of chunk i + 1 already runs (the draws of a chain do not
// depend on where a launch is cut: same results as one launch).  Device staging stays bounded (2 x <= 256 MiB per output array)
// whatever n_draws is.  The copy itself runs at the PCIe rate when the destination is pinned / registered memory
// (nm_host_register) or has been touched before; first-touch pageable memory is bound by the page faults (~17 GB/s).
extern "C" nm_status nm_engine_draw_ex_to_host(nm_engine* e, uint64_t n_draws, const nm_draw_outputs* h_out) {
    if (!e) return fail(NM_ERR_INVALID_ARG, "null engine");
    if (!h_out) return fail(NM_ERR_INVALID_ARG, "null nm_draw_outputs");
    if (n_draws == 0) return NM_OK;
    HIP_TRY(hipSetDevice(e->device));
    const size_t vec_row = (size_t)e->n_chains * e->dim * sizeof(double);          // bytes of one draw of all chains
    const size_t st_row = (size_t)e->n_chains * sizeof(nm_draw_stats);
    // (host destination, bytes per draw, is an event array)
    struct Item { char* host; size_t row; bool event; };
    const Item items[11] = {
        {(char*)h_out->d_positions, vec_row, false},
        {(char*)h_out->d_stats, st_row, false},
        {(char*)h_out->d_gradient, vec_row, false},
        {(char*)h_out->d_transformed_position, vec_row, false},
        {(char*)h_out->d_transformed_gradient, vec_row, false},
        {(char*)h_out->d_mass_matrix_inv, vec_row, true},
        {(char*)h_out->d_transformation_mu, vec_row, true},
        {(char*)h_out->d_divergence_start, vec_row, true},
        {(char*)h_out->d_divergence_start_gradient, vec_row, true},
        {(char*)h_out->d_divergence_end, vec_row, true},
        {(char*)h_out->d_mass_matrix_eigvals, vec_row, true},
    };
    constexpr size_t STAGE_CAP = 256ull << 20;
    const uint64_t chunk = std::min<uint64_t>(n_draws, std::max<uint64_t>(1, STAGE_CAP / std::max(vec_row, st_row)));
    const uint64_t n_chunks = (n_draws + chunk - 1) / chunk;
    const int n_sets = n_chunks > 1 ? 2 : 1;
    if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    for (int q = 0; q < 2; ++q) if (!e->ev_chunk[q]) HIP_TRY(hipEventCreateWithFlags(&e->ev_chunk[q], hipEventDisableTiming));
    for (int k = 0; k < 11; ++k) {
        if (!items[k].host) continue;
        const size_t need = (size_t)n_sets * chunk * items[k].row;
        if (e->staging_bytes[k] < need) {              // a controller calls this once per chunk of draws: keep the buffers
            if (e->staging[k]) (void)hipFree(e->staging[k]);
            e->staging[k] = nullptr; e->staging_bytes[k] = 0;
            const hipError_t er = hipMalloc(&e->staging[k], need);
            if (er != hipSuccess) return fail(NM_ERR_HIP, "device staging buffer of %zu bytes: %s", need, hipGetErrorString(er));
            e->staging_bytes[k] = need;
        }
    }
    auto stage = [&](int k, uint64_t i) { return (char*)e->staging[k] + (size_t)(n_sets == 2 ? (i & 1) : 0) * chunk * items[k].row; };
    auto launch_chunk = [&](uint64_t i) -> nm_status {
        const uint64_t c = std::min<uint64_t>(chunk, n_draws - i * chunk);
        void* dev[11] = {};
        for (int k = 0; k < 11; ++k) {
            if (!items[k].host) continue;
            dev[k] = stage(k, i);
            // rows a chain never writes (it stopped mid-launch, or the event did not happen) must not show an earlier chunk's
            // data: vectors read as NaN (all-ones), statistics as zeros
            HIP_TRY(hipMemsetAsync(dev[k], k == 1 ? 0x00 : 0xFF, (size_t)c * items[k].row, e->stream));
        }
        nm_draw_outputs d = {};
        d.d_positions = (double*)dev[0]; d.d_stats = (nm_draw_stats*)dev[1]; d.d_gradient = (double*)dev[2];
        d.d_transformed_position = (double*)dev[3]; d.d_transformed_gradient = (double*)dev[4];
        d.d_mass_matrix_inv = (double*)dev[5]; d.d_transformation_mu = (double*)dev[6];
        d.d_divergence_start = (double*)dev[7]; d.d_divergence_start_gradient = (double*)dev[8]; d.d_divergence_end = (double*)dev[9];
        d.d_mass_matrix_eigvals = (double*)dev[10];
        const nm_status st = nm_engine_draw_ex_async(e, c, &d);     // (synchronous with the low-rank adaptation: estimator rounds)
        if (st != NM_OK) return st;
        HIP_TRY(hipEventRecord(e->ev_chunk[i & 1], e->stream));
        return NM_OK;
    };
    // A freshly allocated destination costs a page fault per 4 KiB while the copy waits (17 GB/s instead of the link's 50+):
    // a few host threads touch the pages of chunk i + 1 while chunk i is on the wire (every byte of the destination is
    // overwritten by the copies afterwards, so writing a zero first changes nothing).
    const unsigned n_touch = std::min(4u, usable_host_threads());
    std::vector<std::thread> touchers;
    auto start_prefault = [&](uint64_t i) {
        const uint64_t c = std::min<uint64_t>(chunk, n_draws - i * chunk);
        for (unsigned t = 0; t < n_touch; ++t)
            touchers.emplace_back([&, i, c, t]() {
                for (int k = 0; k < 11; ++k) {
                    if (!items[k].host) continue;
                    char* const base = items[k].host + (size_t)i * chunk * items[k].row;
                    const size_t bytes = (size_t)c * items[k].row;
                    const size_t lo = bytes / n_touch * t, hi = t + 1 == n_touch ? bytes : bytes / n_touch * (t + 1);
                    for (size_t o = lo; o < hi; o += 4096) *(volatile char*)(base + o) = 0;
                    if (hi > lo) *(volatile char*)(base + hi - 1) = 0;
                }
            });
    };
    auto join_prefault = [&]() { for (auto& t : touchers) t.join(); touchers.clear(); };
    start_prefault(0);
    nm_status st = launch_chunk(0);
    join_prefault();
    for (uint64_t i = 0; i < n_chunks && st == NM_OK; ++i) {
        if (i + 1 < n_chunks) {
            start_prefault(i + 1);
            st = launch_chunk(i + 1);                             // runs while chunk i is copied
        }
        if (st != NM_OK) break;
        const uint64_t c = std::min<uint64_t>(chunk, n_draws - i * chunk);
        hipError_t er = hipStreamWaitEvent(e->copy_stream, e->ev_chunk[i & 1], 0);
        for (int k = 0; k < 11 && er == hipSuccess; ++k)
            if (items[k].host)
                er = hipMemcpyAsync(items[k].host + (size_t)i * chunk * items[k].row, stage(k, i), (size_t)c * items[k].row, hipMemcpyDeviceToHost, e->copy_stream);
        if (er == hipSuccess) er = hipStreamSynchronize(e->copy_stream);      // (its staging set is free again before chunk i + 2 is launched)
        join_prefault();
        if (er != hipSuccess) { (void)nm_engine_synchronize(e); return fail(NM_ERR_HIP, "copy of a result chunk to the host failed: %s", hipGetErrorString(er)); }
    }
    join_prefault();
    const nm_status st_sync = nm_engine_synchronize(e);
    if (st != NM_OK) return st;
    if (st_sync != NM_OK) return st_sync;
    // surface chain failures the way Chain::draw's Result does
    std::vector<ChainScalars> sc(e->n_chains);
    HIP_TRY(read_scalars(e, sc.data()));
    uint64_t failed = 0;
    for (auto& q : sc) if (q.status != NM_CHAIN_OK) failed++;
    if (failed) return fail(NM_ERR_LOGP_FAILURE, "%llu chain(s) stopped with an error status", (unsigned long long)failed);
    return NM_OK;
}
// Pin a host array for the *_to_host calls (hipHostRegister): the copies then run at the PCIe rate, straight into it.
extern "C" nm_status nm_host_register(void* h_ptr, uint64_t bytes) {
    if (!h_ptr || !bytes) return fail(NM_ERR_INVALID_ARG, "null / empty host range");
    nm_status st = ensure_device(-1);
    if (st != NM_OK) return st;
    HIP_TRY(hipHostRegister(h_ptr, bytes, hipHostRegisterDefault));
    return NM_OK;
}
extern "C" nm_status nm_host_unregister(void* h_ptr) {
    if (!h_ptr) return fail(NM_ERR_INVALID_ARG, "null host pointer");
    HIP_TRY(hipHostUnregister(h_ptr));
    return NM_OK;
}
extern "C" nm_status nm_engine_draw_to_host(nm_engine* e, uint64_t n_draws, double* h_positions, nm_draw_stats* h_stats) {
    nm_draw_outputs h = {};
    h.d_positions = h_positions; h.d_stats = h_stats;
    return nm_engine_draw_ex_to_host(e, n_draws, &h);
}

// Development aid (not part of include/nuts_amd.h): the XCD (HW_REG_XCC_ID) each block of a small grid ran on — the placement
// the cluster kernels' block -> (cluster, member) map is built on
__global__ void xcc_probe_kernel(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)xcc_id(); }
extern "C" nm_status nm_debug_xcc_ids(unsigned* h_out, unsigned n_blocks) {
    nm_status st = ensure_device(-1);
    if (st != NM_OK) return st;
    unsigned* d = nullptr;
    HIP_TRY(hipMalloc(&d, n_blocks * sizeof(unsigned)));
    hipStream_t s = nullptr;
    hipError_t er = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (er == hipSuccess) { hipLaunchKernelGGL(xcc_probe_kernel, dim3(n_blocks), dim3(64), 0, s, d); er = hipGetLastError(); }
    if (er == hipSuccess) er = copy_on(s, h_out, d, n_blocks * sizeof(unsigned), hipMemcpyDeviceToHost);
    if (s) (void)hipStreamDestroy(s);
    (void)hipFree(d);
    if (er != hipSuccess) return fail(NM_ERR_HIP, "xcc probe: %s", hipGetErrorString(er));
    return NM_OK;
}

// Development aid (not part of include/nuts_amd.h): cycle counters of NM_PROF builds; reading clears them.
extern "C" nm_status nm_debug_read_prof(nm_engine* e, unsigned long long out[32]) {
    if (!e || !out) return fail(NM_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(copy_on(e->stream, out, e->d_prof, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemsetAsync(e->d_prof, 0, 32 * sizeof(unsigned long long), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return NM_OK;
}

static nm_status read_slot(nm_engine* e, int slot, double* h_out) {
    if (!e || !h_out) return fail(NM_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const KParams& P = e->P;
    // strided copy: row c = vec[c][slot][0..dim)
    if (e->cl_k > 1) {      // a chain's vector is the concatenation of its members' slices
        for (uint64_t m = 0; m < e->cl_k; ++m) {
            const uint64_t off = m * P.cl_slice, len = std::min<uint64_t>(P.cl_slice, e->dim - off);
            HIP_TRY(copy2d_on(e->stream, h_out + off, e->dim * sizeof(double), e->d_pvec + (m * NUM_PSLOT + (size_t)slot) * P.dpad,
                                (size_t)e->cl_k * NUM_PSLOT * P.dpad * sizeof(double), len * sizeof(double), e->n_chains, hipMemcpyDeviceToHost));
        }
        return NM_OK;
    }
    HIP_TRY(copy2d_on(e->stream, h_out, e->dim * sizeof(double), e->d_pvec + (size_t)slot * P.dpad,
                        (size_t)NUM_PSLOT * P.dpad * sizeof(double), e->dim * sizeof(double), e->n_chains, hipMemcpyDeviceToHost));
    return NM_OK;
}
extern "C" uint64_t nm_engine_blocks_per_chain(const nm_engine* e) { return e ? e->cl_k : 0; }
// Development aid (not part of include/nuts_amd.h): any persistent slot of every chain, [n_chains][dim]
extern "C" nm_status nm_debug_read_slot(nm_engine* e, int slot, double* h_out) { return slot >= 0 && slot < NUM_PSLOT ? read_slot(e, slot, h_out) : NM_ERR_INVALID_ARG; }
extern "C" nm_status nm_engine_get_positions(nm_engine* e, double* h_x) { return read_slot(e, P_X, h_x); }
extern "C" nm_status nm_engine_get_gradients(nm_engine* e, double* h_gx) { return read_slot(e, P_GX, h_gx); }
extern "C" nm_status nm_engine_get_mass_matrix(nm_engine* e, double* h_stds, double* h_mean) {
    nm_status st = NM_OK;
    if (h_stds) st = read_slot(e, P_SIG, h_stds);
    if (st == NM_OK && h_mean) st = read_slot(e, P_MU, h_mean);
    return st;
}
extern "C" nm_status nm_engine_get_step_sizes(nm_engine* e, double* h_step_size) {
    if (!e || !h_step_size) return fail(NM_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<ChainScalars> sc(e->n_chains);
    HIP_TRY(read_scalars(e, sc.data()));
    for (uint64_t c = 0; c < e->n_chains; ++c) h_step_size[c] = sc[c].step_size;
    return NM_OK;
}
extern "C" nm_status nm_engine_get_counters(nm_engine* e, uint64_t* total_leapfrogs, uint64_t* total_draws,
                                            double* kernel_ms, uint64_t* kernel_launches) {
    if (!e) return fail(NM_ERR_INVALID_ARG, "null engine");
    nm_status st = nm_engine_synchronize(e);
    if (st != NM_OK) return st;
    if (total_leapfrogs) {
        std::vector<ChainScalars> sc(e->n_chains);
        HIP_TRY(read_scalars(e, sc.data()));
        uint64_t tot = 0;
        for (auto& q : sc) tot += q.total_steps;
        *total_leapfrogs = tot - e->steps_base;
    }
    if (total_draws) *total_draws = e->draws_total;
    if (kernel_ms) *kernel_ms = e->kernel_ms;
    if (kernel_launches) *kernel_launches = e->kernel_launches;
    return NM_OK;
}
extern "C" nm_status nm_engine_reset_counters(nm_engine* e) {
    if (!e) return fail(NM_ERR_INVALID_ARG, "null engine");
    uint64_t tot = 0;
    nm_status st = nm_engine_get_counters(e, &tot, nullptr, nullptr, nullptr);
    if (st != NM_OK) return st;
    e->steps_base += tot; e->draws_total = 0; e->kernel_ms = 0.0; e->kernel_launches = 0;
    return NM_OK;
}
extern "C" uint64_t nm_engine_dim(const nm_engine* e) { return e ? e->dim : 0; }
extern "C" uint64_t nm_engine_num_chains(const nm_engine* e) { return e ? e->n_chains : 0; }
extern "C" uint64_t nm_engine_threads_per_chain(const nm_engine* e) { return e ? 64ull * (uint64_t)e->wpc : 0; }
extern "C" uint64_t nm_engine_dims_per_lane(const nm_engine* e) { return e ? (uint64_t)e->dpl : 0; }
extern "C" uint64_t nm_engine_group_launches(const nm_engine* e) { return e ? e->group_launches : 0; }
extern "C" void* nm_engine_stream(nm_engine* e) { return e ? (void*)e->stream : nullptr; }

// ---------------------------------------------------------------------------------------------
// batched Math primitives (unit-parity surface): rows are [n][dim], unpadded
// ---------------------------------------------------------------------------------------------
namespace nm {
struct LfArgs {
    KParams P;
    const double *z, *v, *gz, *sigma, *mu, *eps, *logdet, *e0;
    double *z_out, *v_out, *gz_out, *x_out, *gx_out, *logp_out, *ke_out, *err_out;
};
template <int DPL>
NM_DEV void load_row(Tile<DPL>& t, const double* row, int dim) {
#pragma unroll
    for (int k = 0; k < DPL; ++k) { int d = elem_index<1>(k); t.a[k] = d < dim ? row[d] : 0.0; }
}
template <int DPL>
NM_DEV void store_row(const Tile<DPL>& t, double* row, int dim) {
#pragma unroll
    for (int k = 0; k < DPL; ++k) { int d = elem_index<1>(k); if (d < dim) row[d] = t.a[k]; }
}
template <int DPL, class Dens>
__global__ __launch_bounds__(64) void leapfrog_batch_kernel(const LfArgs A) {
    dm_init_lds();
    const uint64_t i = blockIdx.x;
    const int dim = (int)A.P.dim;
    __shared__ double lsig[64 * DPL], lmu[64 * DPL], lred[2 * RED_MAX_VALUES];
    __shared__ ChainScalars lsc;
    __shared__ double ldens[Dens::kNeedsLdsVector ? 64 * DPL : 2];
    ChainCtx<DPL, 1, Dens> C(A.P, lsc);
    C.dim = dim;
    C.red.init(lred);
    C.dens.init(A.P.logp_params, dim, C.red);
    C.dens.set_lds(ldens);
    C.lsig = lsig; C.lmu = lmu;
    {
        Tile<DPL> t;
        load_row(t, A.sigma + i * dim, dim); C.store(t, C.lsig);
        load_row(t, A.mu + i * dim, dim); C.store(t, C.lmu);
    }
    Pt<DPL> s0, s;
    load_row(s0.z, A.z + i * dim, dim);
    load_row(s0.v, A.v + i * dim, dim);
    load_row(s0.g, A.gz + i * dim, dim);
    Tile<DPL> x, gx;
    leapfrog(C, s0, s, A.eps[i], &x, &gx);
    store_row(s.z, A.z_out + i * dim, dim); store_row(s.v, A.v_out + i * dim, dim);
    store_row(s.g, A.gz_out + i * dim, dim); store_row(x, A.x_out + i * dim, dim);
    store_row(gx, A.gx_out + i * dim, dim);
    if (lane_id() == 0) {
        A.logp_out[i] = s.logp;
        A.ke_out[i] = s.ke;
        A.err_out[i] = (s.ke - (s.logp + A.logdet[i])) - A.e0[i];
    }
}
// the three maps of the low-rank transformation for chain i of a batch (nm_lowrank_transform_batch): the block first
// packs its chain's unpadded inputs into the engine's layouts (pvec slots, lrvec rows, lrval), then applies the map
struct LrtArgs {
    KParams P;
    uint64_t which, n_eig;
    const double *stds, *mean, *vals, *vecs, *mu_lr, *in;
    double* out;
};
template <int DPL>
__global__ __launch_bounds__(64) void lowrank_transform_batch_kernel(const LrtArgs A) {
    dm_init_lds();
    typedef LrWrap<IidNormal> D;
    const uint64_t i = blockIdx.x;
    const int dim = (int)A.P.dim;
    __shared__ double lsig[64 * DPL], lmu[64 * DPL], lred[2 * RED_MAX_VALUES];
    __shared__ ChainScalars lsc;
    ChainCtx<DPL, 1, D> C(A.P, lsc);
    C.dim = dim;
    C.red.init(lred);
    C.lsig = lsig; C.lmu = lmu;
    C.slot_bytes = (int)(A.P.dpad * 8);
    C.voff = tid() * 16;
    C.pv = A.P.pvec + (size_t)i * NUM_PSLOT * A.P.dpad;
    C.rp = make_rsrc(C.pv, (uint64_t)NUM_PSLOT * A.P.dpad * 8);
    double* lv = A.P.lrvec + (size_t)i * (1 + A.P.lr_rmax) * A.P.dpad;
    double* lw = A.P.lrval + (size_t)i * 2 * A.P.lr_rmax;
    C.rl = make_rsrc(lv, (uint64_t)(1 + A.P.lr_rmax) * A.P.dpad * 8);
    C.lvals = lw;
    Tile<DPL> t, u;
    load_row(t, A.stds + i * dim, dim); C.store(t, C.lsig);
#pragma unroll
    for (int k = 0; k < DPL; ++k) u.a[k] = elem_index<1>(k) < dim ? 1.0 / t.a[k] : 0.0;
    C.storeP(u, P_ISIG);
    load_row(t, A.mean + i * dim, dim); C.store(t, C.lmu);
    load_row(t, A.mu_lr + i * dim, dim); C.store(t, lv);
    for (uint64_t k = 0; k < A.n_eig; ++k) {
        load_row(t, A.vecs + (i * A.n_eig + k) * dim, dim);
        C.store(t, lv + (1 + k) * A.P.dpad);
    }
    for (uint64_t k = tid(); k < A.n_eig; k += 64) {
        const double sq = __builtin_sqrt(A.vals[i * A.n_eig + k]);
        lw[k] = sq; lw[A.P.lr_rmax + k] = 1.0 / sq;
    }
    if (tid() == 0) { lsc.lr_has_inner = 1; lsc.lr_rank = A.n_eig; }
    __threadfence_block();
    __syncthreads();
    load_row(t, A.in + i * dim, dim);
    if (A.which == 0) transform_to_z(C, t, u);
    else if (A.which == 1) transform_to_x(C, t, u);
    else transform_to_gz(C, t, u);
    store_row(u, A.out + i * dim, dim);
}

template <int DPL>
__global__ __launch_bounds__(64) void turning_batch_kernel(uint64_t dim, const double* zs, const double* vs,
                                                           const double* ze, const double* ve, double* out) {
    const uint64_t i = blockIdx.x;
    Tile<DPL> a, b, c, d;
    load_row(a, zs + i * dim, (int)dim); load_row(b, vs + i * dim, (int)dim);
    load_row(c, ze + i * dim, (int)dim); load_row(d, ve + i * dim, (int)dim);
    double t1 = 0., t2 = 0.;
#pragma unroll
    for (int k = 0; k < DPL; ++k) turn_acc(a.a[k], b.a[k], c.a[k], d.a[k], t1, t2);
    wave_sum2(t1, t2);
    if (lane_id() == 0) { out[2 * i] = t1; out[2 * i + 1] = t2; }
}
__global__ void scalar_math_kernel(uint64_t op, uint64_t n, const double* a, const double* b, double* out) {
    dm_init_lds();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b ? b[i] : 0.0, r;
    switch (op) {
    case 0: r = dexp(x); break;
    case 1: r = dlog(x); break;
    case 2: r = dlog1p(x); break;
    case 3: r = logaddexp_lane(x, y); break;
    case 4: r = __builtin_sqrt(x); break;
    case 5: r = x / y; break;
    case 7: r = dexpm1(x); break;
    case 8: r = dsincos(x).x; break;
    case 9: r = dsincos(x).y; break;
    default: r = __builtin_nan("");
    }
    out[i] = r;
}
__global__ __launch_bounds__(64) void normal_batch_kernel(uint64_t count, const uint32_t* keys, const double* zig_x,
                                                          const double* zig_f, double* out, uint64_t* words) {
    dm_init_lds();
    __shared__ uint32_t cache[RNG_CACHE_WORDS];
    __shared__ double stage[1024];   // LDS staging in this test kernel (the engine stages through its HBM scratch)
    const uint64_t i = blockIdx.x;
    DevRng rng;
    rng.init(keys + 8 * i, 0, cache);
    ZigTables T = {zig_x, zig_f};
    uint64_t done = 0;
    while (done < count) {
        int chunk = (count - done) < 1024 ? (int)(count - done) : 1024;
        fill_standard_normals(rng, stage, chunk, T);
        for (int j = lane_id(); j < chunk; j += 64) out[i * count + done + j] = stage[j];
        __syncthreads();
        done += chunk;
    }
    if (words && lane_id() == 0) words[i] = rng.pos;
}
}  // namespace nm

template <class Dens>
static hipError_t launch_lf_d(int dpl, const LfArgs& A, uint64_t n, hipStream_t st) {
    dim3 g((unsigned)n), b(64);
    switch (dpl) {
    case 2: hipLaunchKernelGGL((leapfrog_batch_kernel<2, Dens>), g, b, 0, st, A); break;
    case 4: hipLaunchKernelGGL((leapfrog_batch_kernel<4, Dens>), g, b, 0, st, A); break;
    case 8: hipLaunchKernelGGL((leapfrog_batch_kernel<8, Dens>), g, b, 0, st, A); break;
    case 16: hipLaunchKernelGGL((leapfrog_batch_kernel<16, Dens>), g, b, 0, st, A); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

extern "C" nm_status nm_leapfrog_batch(const nm_logp_spec* logp, uint64_t n, uint64_t dims_per_lane,
                                       const double* d_z, const double* d_v, const double* d_gz,
                                       const double* d_sigma, const double* d_mu,
                                       const double* d_eps, const double* d_logdet, const double* d_initial_energy,
                                       double* d_z_out, double* d_v_out, double* d_gz_out,
                                       double* d_x_out, double* d_gx_out,
                                       double* d_logp_out, double* d_kinetic_out, double* d_energy_error_out,
                                       void* stream) {
    nm_status st = check_logp(logp);
    if (st == NM_OK && (logp->kind == NM_LOGP_MODULE || logp->kind == NM_LOGP_HOST_CALLBACK)) return fail(NM_ERR_UNSUPPORTED, "nm_leapfrog_batch covers the built-in densities only");
    if (st != NM_OK) return st;
    st = ensure_device(-1);
    if (st != NM_OK) return st;
    const int dpl = pick_dpl(logp->dim, dims_per_lane);
    if (!dpl) return fail(NM_ERR_UNSUPPORTED, "unsupported dim / dims_per_lane");
    if (n == 0) return NM_OK;
    double* d_params = nullptr;
    HIP_TRY(hipMalloc(&d_params, (logp->n_params ? logp->n_params : 1) * sizeof(double)));
    if (logp->n_params) HIP_TRY(copy_on((hipStream_t)stream, d_params, logp->h_params, logp->n_params * sizeof(double), hipMemcpyHostToDevice));
    LfArgs A;
    memset(&A, 0, sizeof A);
    A.P.dim = logp->dim; A.P.dpad = 64ull * dpl; A.P.logp_params = d_params; A.P.n_chains = n;
    A.z = d_z; A.v = d_v; A.gz = d_gz; A.sigma = d_sigma; A.mu = d_mu; A.eps = d_eps; A.logdet = d_logdet; A.e0 = d_initial_energy;
    A.z_out = d_z_out; A.v_out = d_v_out; A.gz_out = d_gz_out; A.x_out = d_x_out; A.gx_out = d_gx_out;
    A.logp_out = d_logp_out; A.ke_out = d_kinetic_out; A.err_out = d_energy_error_out;
    hipError_t er;
    switch (logp->kind) {
    case NM_LOGP_IID_NORMAL: er = launch_lf_d<IidNormal>(dpl, A, n, (hipStream_t)stream); break;
    case NM_LOGP_DIAG_NORMAL: er = launch_lf_d<DiagNormal>(dpl, A, n, (hipStream_t)stream); break;
    case NM_LOGP_FUNNEL: er = launch_lf_d<Funnel>(dpl, A, n, (hipStream_t)stream); break;
    case NM_LOGP_MVN_PREC: er = launch_lf_d<MvnPrec>(dpl, A, n, (hipStream_t)stream); break;
    default: er = launch_lf_d<EightSchools>(2, A, n, (hipStream_t)stream); break;
    }
    if (er == hipSuccess) er = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d_params);
    if (er != hipSuccess) return fail(NM_ERR_HIP, "leapfrog_batch: %s", hipGetErrorString(er));
    return NM_OK;
}

extern "C" nm_status nm_lowrank_transform_batch(uint64_t which, uint64_t n, uint64_t dim, uint64_t n_eig, uint64_t dims_per_lane,
                                                const double* d_stds, const double* d_mean, const double* d_vals, const double* d_vecs,
                                                const double* d_mu_lr, const double* d_in, double* d_out, void* stream) {
    nm_status st = ensure_device(-1);
    if (st != NM_OK) return st;
    if (which > 2) return fail(NM_ERR_INVALID_ARG, "which must be 0 (x -> z), 1 (z -> x) or 2 (g_x -> g_z)");
    const int dpl = pick_dpl(dim, dims_per_lane);
    if (!dpl) return fail(NM_ERR_UNSUPPORTED, "unsupported dim / dims_per_lane (one wavefront per chain: dim <= 1024)");
    if (n == 0) return NM_OK;
    LrtArgs A;
    memset(&A, 0, sizeof A);
    const uint64_t dpad = 64ull * dpl, R = n_eig ? n_eig : 1;
    double *pv = nullptr, *lv = nullptr, *lw = nullptr;
    HIP_TRY(hipMalloc(&pv, n * NUM_PSLOT * dpad * 8));
    HIP_TRY(hipMalloc(&lv, n * (1 + R) * dpad * 8));
    HIP_TRY(hipMalloc(&lw, n * 2 * R * 8));
    A.P.dim = dim; A.P.dpad = dpad; A.P.n_chains = n; A.P.pvec = pv; A.P.lrvec = lv; A.P.lrval = lw; A.P.lr_rmax = R;
    A.which = which; A.n_eig = n_eig; A.stds = d_stds; A.mean = d_mean; A.vals = d_vals; A.vecs = d_vecs; A.mu_lr = d_mu_lr;
    A.in = d_in; A.out = d_out;
    dim3 g((unsigned)n), b(64);
    hipStream_t s = (hipStream_t)stream;
    hipError_t er = hipMemsetAsync(pv, 0, n * NUM_PSLOT * dpad * 8, s);
    if (er == hipSuccess) er = hipMemsetAsync(lv, 0, n * (1 + R) * dpad * 8, s);
    if (er == hipSuccess) {
        switch (dpl) {
        case 2: hipLaunchKernelGGL((lowrank_transform_batch_kernel<2>), g, b, 0, s, A); break;
        case 4: hipLaunchKernelGGL((lowrank_transform_batch_kernel<4>), g, b, 0, s, A); break;
        case 8: hipLaunchKernelGGL((lowrank_transform_batch_kernel<8>), g, b, 0, s, A); break;
        case 16: hipLaunchKernelGGL((lowrank_transform_batch_kernel<16>), g, b, 0, s, A); break;
        }
        er = hipGetLastError();
    }
    if (er == hipSuccess) er = hipStreamSynchronize(s);
    (void)hipFree(pv); (void)hipFree(lv); (void)hipFree(lw);
    if (er != hipSuccess) return fail(NM_ERR_HIP, "lowrank_transform_batch: %s", hipGetErrorString(er));
    return NM_OK;
}

extern "C" nm_status nm_turning_batch(uint64_t n, uint64_t dim, uint64_t dims_per_lane,
                                      const double* d_z_start, const double* d_v_start,
                                      const double* d_z_end, const double* d_v_end, double* d_out_t, void* stream) {
    nm_status st = ensure_device(-1);
    if (st != NM_OK) return st;
    const int dpl = pick_dpl(dim, dims_per_lane);
    if (!dpl) return fail(NM_ERR_UNSUPPORTED, "unsupported dim / dims_per_lane");
    if (n == 0) return NM_OK;
    dim3 g((unsigned)n), b(64);
    hipStream_t s = (hipStream_t)stream;
    switch (dpl) {
    case 2: hipLaunchKernelGGL((turning_batch_kernel<2>), g, b, 0, s, dim, d_z_start, d_v_start, d_z_end, d_v_end, d_out_t); break;
    case 4: hipLaunchKernelGGL((turning_batch_kernel<4>), g, b, 0, s, dim, d_z_start, d_v_start, d_z_end, d_v_end, d_out_t); break;
    case 8: hipLaunchKernelGGL((turning_batch_kernel<8>), g, b, 0, s, dim, d_z_start, d_v_start, d_z_end, d_v_end, d_out_t); break;
    case 16: hipLaunchKernelGGL((turning_batch_kernel<16>), g, b, 0, s, dim, d_z_start, d_v_start, d_z_end, d_v_end, d_out_t); break;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return NM_OK;
}

extern "C" nm_status nm_scalar_math_batch(uint64_t op, uint64_t n, const double* d_a, const double* d_b, double* d_out, void* stream) {
    nm_status st = ensure_device(-1);
    if (st != NM_OK) return st;
    if (op > 9 || op == 6) return fail(NM_ERR_INVALID_ARG, "unknown op");
    if (n == 0) return NM_OK;
    hipLaunchKernelGGL(scalar_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, n, d_a, d_b, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return NM_OK;
}

extern "C" nm_status nm_standard_normal_batch(uint64_t n, uint64_t count, const uint8_t* h_keys, double* d_out,
                                              uint64_t* h_words_consumed, void* stream) {
    nm_status st = ensure_device(-1);
    if (st != NM_OK) return st;
    if (!h_keys || !d_out) return fail(NM_ERR_INVALID_ARG, "null argument");
    if (n == 0 || count == 0) return NM_OK;
    std::vector<uint32_t> keys(8 * n);
    for (uint64_t i = 0; i < 8 * n; ++i)
        keys[i] = (uint32_t)h_keys[4 * i] | ((uint32_t)h_keys[4 * i + 1] << 8) | ((uint32_t)h_keys[4 * i + 2] << 16) | ((uint32_t)h_keys[4 * i + 3] << 24);
    std::vector<double> t(kZigTables, kZigTables + 2 * 257);
    uint32_t* d_keys = nullptr; double* d_zig = nullptr; uint64_t* d_words = nullptr;
    HIP_TRY(hipMalloc(&d_keys, keys.size() * 4));
    HIP_TRY(hipMalloc(&d_zig, t.size() * 8));
    HIP_TRY(hipMalloc(&d_words, n * 8));
    HIP_TRY(copy_on((hipStream_t)stream, d_keys, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(copy_on((hipStream_t)stream, d_zig, t.data(), t.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(normal_batch_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, count, d_keys, d_zig, d_zig + 257, d_out, d_words);
    hipError_t er = hipGetLastError();
    if (er == hipSuccess) er = hipStreamSynchronize((hipStream_t)stream);
    if (er == hipSuccess && h_words_consumed) er = copy_on((hipStream_t)stream, h_words_consumed, d_words, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_keys); (void)hipFree(d_zig); (void)hipFree(d_words);
    if (er != hipSuccess) return fail(NM_ERR_HIP, "standard_normal_batch: %s", hipGetErrorString(er));
    return NM_OK;
}
