// nuts_lane.hpp — ONE CHAIN PER LANE: the draw kernels for very small chains (dim <= 16), 64 chains per wavefront.
//
// north_star's own mapping ("one chain per wavefront lane"), where it pays: at dim 10 (BASELINE configs 1 and 4) the
// wave-per-chain kernel executes one chain's scalar work 64 lanes wide, and the 8-lanes-per-chain kernels of nuts_group.hpp still
// carry eight copies of it per chain (131 vector instructions per chain-leapfrog, 236-256 registers of per-lane duplicates).
// Here a lane owns a whole chain: its vectors are 2 NP doubles in registers (NP = pairs of elements, dim <= 2 NP <= 16), the tree's
// "scalars" are the lane's own registers, nothing is duplicated and no cross-lane operation exists.  The code below is the
// reference's transition written once, per lane (a port of nuts_group_impl.hpp, itself a port of nuts_transition); lanes diverge
// where their trees do, and the wavefront's doubling / leaf loops run until its last lane is done.
//
// Same results, bit for bit, as the other kernels (and so as the oracle):
//   * sums over dim: the wave kernel's order for a chain of <= 16 elements is "pair partials (elements 2l, 2l + 1, from +0.0),
//     then the xor-butterfly 1, 2, 4 over the 8 lanes that hold them" = a balanced tree over 8 pair partials; pairs beyond the
//     dim are +0.0 and adding +0.0 to a partial sum (never -0.0: it starts at +0.0) changes nothing, so they are skipped (pair_tree);
//   * the same exp / ln / ln_1p (the per-lane entry points), the same ChaCha8 stream (every lane generates ITS chain's blocks),
//     the sequential ziggurat of rand_distr, the same tree, merges, adaptation;
//   * the chain's state lives in the same pvec slots / ChainScalars records between launches (gathered into a lane-major
//     workspace when the kernel starts and scattered back when it ends), so launches of this kernel and of the others can
//     alternate on one engine.
// Memory: the tree's scratch and the chain's persistent vectors are LANE-MAJOR ([slot][element][lane]): every access of a
// wavefront is 64 consecutive doubles.  The pending-sub-tree table and the ChaCha word cache are in LDS ([..][lane]: bank =
// lane, conflict-free).
#pragma once
#include <type_traits>
#include "nuts_kernels.hpp"

namespace nm {
namespace lane {

constexpr int LMAXDEPTH = 10;
// (round 5: the 8-pair kernel for dim 11 .. 16 — 2 KB of scratch per lane, slower than the 8-lane kernels at every size measured — is removed)
__host__ __device__ inline int lane_pairs(uint64_t dim) { return dim <= 4 ? 2 : dim <= 8 ? 4 : dim <= 10 ? 5 : 0; }

struct LaneParams {
    double* lws;          // [grid][NUM_PSLOT][2 NP][64]  the chains' persistent vectors while the kernel runs
    double* lsv;          // [grid][nslots][2 NP][64]     tree scratch
    uint64_t nslots;
};

// ---- the engine's sum over a chain of <= 16 elements: balanced tree over the pair partials (see the header) ----
template <int NP>
NM_DEV double pair_tree(const double (&p)[NP]) {
    constexpr int NQ = (NP + 1) / 2;
    double q[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) q[i] = (2 * i + 1 < NP) ? p[2 * i] + p[2 * i + 1] : p[2 * i];
    constexpr int NR = (NQ + 1) / 2;
    double r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = (2 * i + 1 < NQ) ? q[2 * i] + q[2 * i + 1] : q[2 * i];
    if constexpr (NR == 2) return r[0] + r[1];
    else return r[0];
}

// per-lane special functions (same operations, same bits as nm::dexp & co.)
NM_DEV double lexp(double x) { return dexp_impl<false>(x); }
NM_DEV double llog(double x) { return dlog_impl<false>(x); }
NM_DEV double llog1p(double x) { return dlog1p_impl<false>(x); }
NM_DEV double llogaddexp(double a, double b) {           // reference src/math/util.rs:6-19
    if (a == b) return a + llog(2.0);
    const double diff = a - b;
    if (diff > 0.) return a + llog1p(lexp(-diff));
    if (diff < 0.) return b + llog1p(lexp(diff));
    return diff;
}

// ---- densities: lane forms of the built-in ones (same operations in the same order as the group / wave forms) ----
template <int NP>
struct LIidNormal {
    static constexpr int E = 2 * NP;
    double mu;
    NM_DEV void init(const double* params, int) { mu = params[0]; }
    NM_DEV double eval(const double (&x)[E], double (&gx)[E], int dim) const {
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int d = 2 * l + k;
                const bool valid = d < dim;
                const double diff = x[d] - mu;
                const double term = -0.5 * diff * diff;
                gx[d] = valid ? -diff : 0.0;
                acc = acc + (valid ? term : 0.0);
            }
            p[l] = acc;
        }
        return pair_tree<NP>(p);
    }
};
template <int NP>
struct LDiagNormal {
    static constexpr int E = 2 * NP;
    const double* prec;
    double norm;
    NM_DEV void init(const double* params, int dim) {
        prec = params;
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
            for (int j = 0; j < 2; ++j) {
                const int d = 2 * l + j;
                acc = acc + (d < dim ? llog(params[d < dim ? d : 0]) : 0.0);
            }
            p[l] = acc;
        }
        const double log_det_p = pair_tree<NP>(p);
        norm = -0.5 * ((double)dim * llog(6.283185307179586) - log_det_p);
    }
    NM_DEV double eval(const double (&x)[E], double (&gx)[E], int dim) const {
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int d = 2 * l + k;
                const bool valid = d < dim;
                const double pr = valid ? prec[d] : 0.0;
                const double px = pr * x[d];
                gx[d] = valid ? -px : 0.0;
                acc = acc + (valid ? x[d] * px : 0.0);
            }
            p[l] = acc;
        }
        const double quad = -0.5 * pair_tree<NP>(p);
        return quad + norm;
    }
};
template <int NP>
struct LEightSchools {          // dim 10 (NP = 5): mu, log tau, theta~_1..8
    static constexpr int E = 2 * NP;
    const double* par;
    NM_DEV void init(const double* params, int) { par = params; }
    NM_DEV double eval(const double (&x)[E], double (&gx)[E], int) const {
        const double mu = x[0], lt = x[1];
        const double tau = lexp(lt);
        const double t5 = (tau / 5.0) * (tau / 5.0);
        const double prior_tau = lt - llog1p(t5);
        double term[E], dr[E], drth[E];
#pragma unroll
        for (int d = 0; d < E; ++d) {
            const bool school = d >= 2 && d < 10;
            const int i = school ? d - 2 : 0;
            const double th = x[d];
            const double sg = par[8 + i];
            const double r = (par[i] - (mu + tau * th)) / sg;
            term[d] = -0.5 * th * th - 0.5 * r * r;
            dr[d] = r / sg;
            drth[d] = dr[d] * th;
            if (!school) { term[d] = 0.0; dr[d] = 0.0; drth[d] = 0.0; }
        }
        double gmu = 0.0, gtl = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (2 + i < E) { gmu = gmu + dr[2 + i]; gtl = gtl + drth[2 + i]; }
        }
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int d = 2 * l + k;
                double t = 0.0, g = 0.0;
                if (d == 0) { t = -mu * mu / 50.0; g = -mu / 25.0 + gmu; }
                else if (d == 1) { t = prior_tau; g = 1.0 - 2.0 * t5 / (1.0 + t5) + gtl * tau; }
                else if (d < 10) { t = term[d]; g = -x[d] + dr[d] * tau; }
                gx[d] = g;
                acc = acc + t;
            }
            p[l] = acc;
        }
        return pair_tree<NP>(p);
    }
};
template <int NP>
struct LFunnel {
    static constexpr int E = 2 * NP;
    NM_DEV void init(const double*, int) {}
    NM_DEV double eval(const double (&x)[E], double (&gx)[E], int dim) const {
        const double v = x[0];
        const double kk = (double)(dim - 1);
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int d = 2 * l + k;
                const bool in = d >= 1 && d < dim;
                acc = acc + (in ? x[d] * x[d] : 0.0);
            }
            p[l] = acc;
        }
        const double ss = pair_tree<NP>(p);
        const double ev = lexp(-v);
        const double g0 = -v / 9.0 - 0.5 * kk + 0.5 * ev * ss;
#pragma unroll
        for (int d = 0; d < E; ++d) gx[d] = d == 0 ? g0 : (d < dim ? -ev * x[d] : 0.0);
        return -v * v / 18.0 - 0.5 * kk * v - 0.5 * ev * ss;
    }
};
template <class Dens, int NP> struct LaneDensity { using type = void; };
// KinWrap<D> (nuts_kernels.hpp): the density's own lane form; the non-Euclidean trajectory kinds are switched on in the lane code by the marker
// LKin around it (round 4; kern_lane_kin.hip) — the Euclidean instantiations compile the code they always had
template <class D, int NP> struct LaneDensity<KinWrap<D>, NP> { using type = typename LaneDensity<D, NP>::type; };
template <class LD> struct LKin : LD {};
template <class LD> struct lkin_trait { static constexpr bool value = false; };
template <class LD> struct lkin_trait<LKin<LD>> { static constexpr bool value = true; };
template <int NP> struct LaneDensity<IidNormal, NP> { using type = LIidNormal<NP>; };
template <int NP> struct LaneDensity<DiagNormal, NP> { using type = LDiagNormal<NP>; };
template <int NP> struct LaneDensity<Funnel, NP> { using type = LFunnel<NP>; };
template <> struct LaneDensity<EightSchools, 5> { using type = LEightSchools<5>; };

// ---- the chain's generator: every lane produces the blocks of ITS stream into a ring of two 16-word blocks in LDS ([word][lane]) ----
// The lanes' stream positions drift apart (rejections, tree shapes), so "refill when MY block is used up" makes nearly every call
// of the wave pay a ChaCha block for somebody (1 - (15/16)^64 = 98 %; measured: the momentum refresh's 20 words cost 20 blocks).
// Instead a refill is a wave event: when ANY active lane has run dry, EVERY active lane whose older block is used up produces
// its next one.  After an event each participating lane holds more than 16 unread words, so events are at least 16 words apart.
struct LRng {
    uint32_t key[8];
    uint64_t pos;          // next word of the stream
    uint64_t filled;       // the blocks below this number have been produced; the ring holds blocks filled - 2 and filled - 1
    uint32_t* ring;        // LDS, this lane's column: stream word w at ring[64 (w & 31)]
    NM_DEV void init(const uint32_t* k, uint64_t p, uint32_t* col) {
#pragma unroll
        for (int i = 0; i < 8; ++i) key[i] = k[i];
        pos = p; filled = p >> 4; ring = col;
    }
    NM_DEV uint32_t next_u32() {
        if (__any((pos >> 4) >= filled)) {
            if (pos + 16 >= 16 * filled) {              // block filled - 2 is used up (or nothing is held yet): its half is free
                uint32_t out[16];
                chacha8_block(key, filled, 0ull, out);
                uint32_t* half = ring + 64 * 16 * (uint32_t)(filled & 1);
#pragma unroll
                for (int i = 0; i < 16; ++i) half[64 * i] = out[i];
                filled += 1;
                asm volatile("" ::: "memory");
            }
        }
        const uint32_t w = ring[64 * (uint32_t)(pos & 31)];
        pos += 1;
        return w;
    }
    NM_DEV uint64_t next_u64() { const uint64_t lo = next_u32(), hi = next_u32(); return (hi << 32) | lo; }   // consecutive stream words (BlockRng::next_u64)
    NM_DEV bool random_bool_std() { return (int32_t)next_u32() < 0; }
    NM_DEV double random_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    NM_DEV int random_bool(double p) {
        if (!(p >= 0.0 && p < 1.0)) return p == 1.0 ? 1 : -1;
        const uint64_t p_int = (uint64_t)(p * 18446744073709551616.0);
        return next_u64() < p_int ? 1 : 0;
    }
};
// rand_distr's StandardNormal, one sample (the sequential ziggurat: its first test is the fast path)
NM_DEV double l_normal(LRng& rng, ZigTables T) {
    uint64_t bits = rng.next_u64();
    for (;;) {
        const int i = (int)(bits & 0xff);
        const double u = u2d((bits >> 12) | 0x4000000000000000ull) - 3.0;
        const double x = u * T.x[i];
        if (__builtin_fabs(x) < T.x[i + 1]) return x;
        if (i == 0) {
            double xx = 1.0, yy = 0.0;
            while (-2.0 * yy < xx * xx) {
                const double a = u2d((rng.next_u64() >> 12) | 0x3ff0000000000000ull) - (1.0 - 2.220446049250313e-16 / 2.0);
                const double b = u2d((rng.next_u64() >> 12) | 0x3ff0000000000000ull) - (1.0 - 2.220446049250313e-16 / 2.0);
                xx = llog(a) / ZIG_R;
                yy = llog(b);
            }
            return u < 0.0 ? xx - ZIG_R : ZIG_R - xx;
        }
        if (T.f[i + 1] + (T.f[i] - T.f[i + 1]) * rng.random_f64() < lexp(-x * x / 2.0)) return x;
        bits = rng.next_u64();
    }
}

// AcceptanceRateCollector (src/stepsize/dual_avg.rs:112-166): the lane evaluates its exps where the reference does
struct LAccept {
    double initial_energy, sum, sum_sym, max_energy_error;
    uint64_t count;
    NM_DEV void register_init(double e0) { initial_energy = e0; sum = 0.; sum_sym = 0.; count = 0; max_energy_error = 0.; }
    NM_DEV void register_divergent() { sum = sum + 0.; sum_sym = sum_sym + 0.; count += 1; max_energy_error = -__builtin_inf(); }
    NM_DEV void register_ok(double end_energy) {
        const double diff = initial_energy - end_energy;
        const double e = lexp(fmin_rs(diff, 0.));
        const double es = 2. * e / (1. + lexp(diff));
        sum = sum + e;
        sum_sym = sum_sym + es;
        count += 1;
        if (__builtin_fabs(diff) > __builtin_fabs(max_energy_error)) max_energy_error = diff;
    }
    NM_DEV double mean() const { return sum / (double)count; }
    NM_DEV double mean_sym() const { return sum_sym / (double)count; }
};

// pending sub-trees, per level (>= 1), in LDS: [level - 1][word][lane]; words: log_size, cand_logp, cand_ke, (cand_idx, cand_slot)
struct LPend {
    uint64_t* base;        // this lane's column
    NM_DEV void put(int level, double log_size, const CandRef& c) {
        uint64_t* q = base + (size_t)(level - 1) * 4 * 64;
        q[0] = d2u(log_size); q[64] = d2u(c.logp); q[128] = d2u(c.ke);
        q[192] = ((uint64_t)(uint32_t)(int32_t)c.idx) | ((uint64_t)(uint32_t)c.slot << 32);
    }
    NM_DEV void get(int level, double& log_size, CandRef& c) const {
        const uint64_t* q = base + (size_t)(level - 1) * 4 * 64;
        log_size = u2d(q[0]); c.logp = u2d(q[64]); c.ke = u2d(q[128]);
        const uint64_t w = q[192];
        c.idx = (int64_t)(int32_t)(uint32_t)w; c.slot = (int)(int32_t)(uint32_t)(w >> 32);
    }
};

// Development (-DNM_LANE_PROF=1, tools/prof_lane.py): the wave-level timeline of block 0.  A mark charges the shader-clock cycles
// since the wave's previous mark (kept in LDS: ONE clock per wave, so serialised divergent paths are charged once each) to `slot`.
#ifndef NM_LANE_STORE_GUARD
#define NM_LANE_STORE_GUARD 1
#endif
#ifndef NM_LANE_PROF
#define NM_LANE_PROF 0
#endif
#if NM_LANE_PROF
#define NM_LP(C, slot)                                                                                                 \
    do {                                                                                                               \
        if (blockIdx.x == 0) {                                                                                         \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                              \
            const unsigned long long prev_ = *(C).lp_clock;                                                            \
            if ((int)threadIdx.x == __ffsll((unsigned long long)__ballot(1)) - 1) {                                    \
                (void)__hip_atomic_fetch_add(&(C).P.prof[slot], now_ - prev_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                (void)__hip_atomic_fetch_add(&(C).P.prof[16 + (slot)], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  \
            }                                                                                                          \
            *(C).lp_clock = now_;                                                                                      \
        }                                                                                                              \
    } while (0)
#else
#define NM_LP(C, slot) do {} while (0)
#endif

template <int NP>
struct LPt { double z[2 * NP], v[2 * NP], g[2 * NP]; double logp, ke; int64_t idx; };

template <int NP, class LD>
struct LCtx {
    static constexpr int E = 2 * NP;
    const KParams& P;
    LD dens;
    LRng rng;
    ZigTables zig;
    // lane-major persistent slots / tree scratch of this WAVE, addressed as buffer descriptor (SGPRs) + one 32-bit byte offset
    // ((slot E + e) 64 + lane) 8: with plain pointers the compiler keeps one 64-bit per-lane address per (slot, e) alive across the
    // tree loop, hundreds of registers that it then spills and reloads in front of every access (round-3 asm: `scratch_load`
    // before each `global_load`)
    rsrc_t rw, rsv;
    int l8;                // lane * 8
#if NM_LANE_PROF
    unsigned long long* lp_clock;
#endif
    NM_DEV static double bld(rsrc_t r, int off) {
        const v2u q = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        return __hiloint2double((int)q.y, (int)q.x);
    }
    NM_DEV static void bst(rsrc_t r, int off, double a) {
        v2u q; q.x = (unsigned)__double2loint(a); q.y = (unsigned)__double2hiint(a);
        __builtin_amdgcn_raw_buffer_store_b64(q, r, off, 0, 0);
#if NM_LANE_STORE_GUARD
        asm volatile("s_nop 1" ::"v"(q));
#endif
    }
    NM_DEV double ldWe(int slot, int e) const { return bld(rw, l8 + (slot * E + e) * 512); }
    NM_DEV void stWe(int slot, int e, double a) const { bst(rw, l8 + (slot * E + e) * 512, a); }
    NM_DEV double ldSe(int slot, int e) const { return bld(rsv, l8 + (slot * E + e) * 512); }
    NM_DEV void stSe(int slot, int e, double a) const { bst(rsv, l8 + (slot * E + e) * 512, a); }
    LPend pend;
    double* stage;         // LDS, this lane's column of E doubles ([e][lane]); rows beyond dim stay zero
    ChainScalars& sc;
    double sig[E], mu[E];
    // array_gaussian through LDS: ONE copy of the ziggurat (a loop over the elements) instead of 2 NP inlined ones.
    // (A variant that staged the samples as uint64 in the pend table's idle rows and masked the rows beyond dim failed the parity
    // sweep in the sampling kernel although the generated code read correctly — tools/probes/lane_case.py, case 12; not understood,
    // so the staging area stays a zero-initialised array of its own.)
    NM_DEV void draw_normals(double (&v)[E]) {
        for (int d = 0; d < dim; ++d) stage[64 * d] = 1.0 * l_normal(rng, zig);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = stage[64 * e];
    }
    int dim, md;
    __device__ LCtx(const KParams& p, ChainScalars& s) : P(p), sc(s) {}
    NM_DEV void ldW(double (&t)[E], int slot) const {
#pragma unroll
        for (int e = 0; e < E; ++e) t[e] = ldWe(slot, e);
    }
    NM_DEV void stW(const double (&t)[E], int slot) const {
#pragma unroll
        for (int e = 0; e < E; ++e) stWe(slot, e, t[e]);
    }
    NM_DEV void ldS(double (&t)[E], int slot) const {
#pragma unroll
        for (int e = 0; e < E; ++e) t[e] = ldSe(slot, e);
    }
    NM_DEV void stS(const double (&t)[E], int slot) const {
#pragma unroll
        for (int e = 0; e < E; ++e) stSe(slot, e, t[e]);
    }
    // main-tree edges: ids 0 (the initial point), 1, 2 in the scratch slots EDGE<id>_Z / _V / _G.  (The wavefront kernels read edge 0
    // from P_Z / STAGE_V / P_GZ; here the draw's start copies the initial point into EDGE0_* so that an edge is ONE address
    // computation whatever its id — ids differ between the lanes.)
    NM_DEV void ld_edge(LPt<NP>& p, int id) const { ldS(p.z, EDGE0_Z + 3 * id); ldS(p.v, EDGE0_V + 3 * id); ldS(p.g, EDGE0_G + 3 * id); }
};

// ---- the non-Euclidean KineticEnergyKinds in one lane (LKin instantiations only): normalize_tile / esh_update_core / leapfrog_kin of
// nuts_kernels.hpp with the chain's sums formed as the engine forms them — per pair of elements, then the balanced tree over the pairs ----
template <int NP>
NM_DEV double l_sum_sq(const double (&v)[2 * NP]) {
    double p[NP];
#pragma unroll
    for (int l = 0; l < NP; ++l) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) acc = acc + v[2 * l + k] * v[2 * l + k];
        p[l] = acc;
    }
    return pair_tree<NP>(p);
}
template <int NP>
NM_DEV void l_normalize(double (&v)[2 * NP]) {
    const double inv = 1.0 / __builtin_sqrt(l_sum_sq<NP>(v));
#pragma unroll
    for (int d = 0; d < 2 * NP; ++d) v[d] *= inv;
}
template <int NP>
NM_DEV double l_esh_update(const double (&g)[2 * NP], double (&pv)[2 * NP], double step_size, int dim) {
    const double grad_norm = __builtin_sqrt(l_sum_sq<NP>(g));
    const double inv_grad_norm = 1.0 / grad_norm;
    double p[NP];
#pragma unroll
    for (int l = 0; l < NP; ++l) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) acc = acc + pv[2 * l + k] * g[2 * l + k] * inv_grad_norm;
        p[l] = acc;
    }
    const double momentum_proj = pair_tree<NP>(p);
    const double dims_m1 = (double)(dim - 1);
    const double delta = step_size * grad_norm / dims_m1;
    const double zeta = lexp(-delta);
    const double coeff_g = (1.0 - zeta) * (1.0 + zeta + momentum_proj * (1.0 - zeta));
    const double coeff_p = 2.0 * zeta;
#pragma unroll
    for (int d = 0; d < 2 * NP; ++d) pv[d] = coeff_g * (g[d] * inv_grad_norm) + coeff_p * pv[d];
    l_normalize<NP>(pv);
    const double arg = momentum_proj + (1.0 - momentum_proj) * zeta * zeta;
    return (delta - 6.93147180559945286227e-01 + llog1p(arg)) * dims_m1;
}
template <int NP, class LD>
NM_DEV void l_leapfrog_kin(LCtx<NP, LD>& C, const LPt<NP>& s, LPt<NP>& o, double epsilon) {
    constexpr int E = 2 * NP;
    const bool micro = C.sc.kin == NM_TRAJ_MICROCANONICAL;
    const double half = epsilon / 2.;
    const double sqrt_n = __builtin_sqrt((double)C.dim);
    double x[E], gx[E];
    if (micro) {
#pragma unroll
        for (int d = 0; d < E; ++d) o.v[d] = s.v[d];
        o.ke = s.ke + l_esh_update<NP>(s.g, o.v, sqrt_n * epsilon / 2., C.dim);
        const double eps_n = epsilon * sqrt_n;
#pragma unroll
        for (int d = 0; d < E; ++d) o.z[d] = __builtin_fma(eps_n, o.v[d], s.z[d]);
    } else {
        const double2 sc2 = dsincos_impl(epsilon);
        const double es = sc2.x, ec = sc2.y, nes = -es;
#pragma unroll
        for (int d = 0; d < E; ++d) {
            const double vh = __builtin_fma(half, s.z[d] + s.g[d], s.v[d]);
            o.z[d] = __builtin_fma(s.z[d], ec, vh * es);
            o.v[d] = __builtin_fma(s.z[d], nes, vh * ec);
        }
    }
#pragma unroll
    for (int d = 0; d < E; ++d) {
        const double t = o.z[d] * C.sig[d];
        x[d] = __builtin_fma(1.0, C.mu[d], t);
    }
    o.logp = C.dens.eval(x, gx, C.dim);
#pragma unroll
    for (int d = 0; d < E; ++d) o.g[d] = gx[d] * C.sig[d];
    if (micro) {
        o.ke = o.ke + l_esh_update<NP>(o.g, o.v, sqrt_n * epsilon / 2., C.dim);
    } else {
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int d = 2 * l + k;
                o.v[d] = __builtin_fma(half, o.z[d] + o.g[d], o.v[d]);
                acc = __builtin_fma(o.v[d], o.v[d], acc);
            }
            p[l] = acc;
        }
        o.ke = 0.5 * pair_tree<NP>(p);
    }
}
// leapfrog's divergence criterion (transformed_hamiltonian.rs:583-590)
template <int NP, class LD>
NM_DEV bool l_bad_energy(const LCtx<NP, LD>& C, double energy_error, double max_energy_error) {
    if constexpr (lkin_trait<LD>::value) {
        if (C.sc.kin == NM_TRAJ_MICROCANONICAL) return (__builtin_fabs(energy_error) >= max_energy_error) | !is_finite(energy_error);
    }
    return (energy_error > max_energy_error) | !is_finite(energy_error);
}
// 1/2 |v|^2 in the engine's order
template <int NP>
NM_DEV double l_kinetic(const double (&v)[2 * NP]) {
    double p[NP];
#pragma unroll
    for (int l = 0; l < NP; ++l) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) acc = __builtin_fma(v[2 * l + k], v[2 * l + k], acc);
        p[l] = acc;
    }
    return 0.5 * pair_tree<NP>(p);
}
template <int NP, class LD>
NM_DEV void l_leapfrog(LCtx<NP, LD>& C, const LPt<NP>& s, LPt<NP>& o, double epsilon) {
    if constexpr (lkin_trait<LD>::value) {
        if (C.sc.kin != NM_TRAJ_EUCLIDEAN) { l_leapfrog_kin(C, s, o, epsilon); return; }
    }
    constexpr int E = 2 * NP;
    const double half = epsilon / 2.;
    double x[E], gx[E];
#pragma unroll
    for (int d = 0; d < E; ++d) {
        const double vh = __builtin_fma(half, s.g[d], s.v[d]);
        o.v[d] = vh;
        o.z[d] = __builtin_fma(epsilon, vh, s.z[d]);
        const double t = o.z[d] * C.sig[d];
        x[d] = __builtin_fma(1.0, C.mu[d], t);
    }
    o.logp = C.dens.eval(x, gx, C.dim);
    double p[NP];
#pragma unroll
    for (int l = 0; l < NP; ++l) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int d = 2 * l + k;
            o.g[d] = gx[d] * C.sig[d];
            o.v[d] = __builtin_fma(half, o.g[d], o.v[d]);
            acc = __builtin_fma(o.v[d], o.v[d], acc);
        }
        p[l] = acc;
    }
    o.ke = 0.5 * pair_tree<NP>(p);
}

// The three is_turning tests a sub-tree merge makes (src/nuts.rs:143-161) against the just finished leaf O; also the top-level tests
// of a finished doubling (A = the main tree).  Test i is (A_i, B_i) in generation order: fwd: start = A_i, end = B_i; else start = B_i,
// end = A_i.   za: slot of A.first's z (its v is the next slot);  zl: A.last;  zb: B.first, or -1: B.first = the point `bf`.
//
// Shape (round 3, from the phase timeline: the first form waited for HBM once per ELEMENT — four loads, a branch on `fwd`, a wait):
//  * no branch: the sums are always formed the forward way, s = (end_z + 0) - start_z with start = A.  Swapping start and end negates s
//    exactly, so the backward test's two sums are the NEGATED forward sums with their roles exchanged (fma and the pair tree commute
//    with negation; only the sign of an exact zero can differ, and the tests compare with zero): backward turning = any sum > 0;
//  * all operand loads of a chunk (all of them up to 5 pairs, 4 pairs at a time for 8) are issued before the first use: one round trip.
template <int NP, class LD>
NM_DEV bool l_merge_turning(const LCtx<NP, LD>& C, int za, int zl, int zb, const LPt<NP>& bf, const LPt<NP>& o, bool fwd) {
    constexpr int E = 2 * NP;
#ifdef NM_LANE_TURN_CP
    constexpr int CP = NM_LANE_TURN_CP < NP ? NM_LANE_TURN_CP : NP;
#else
    constexpr int CP = NP < 3 ? NP : 3;              // pairs per chunk (measured on K4: 3 -> 6.13e9, 2 -> 5.96e9, all -> 5.1e9 leapfrogs/s)
#endif
    const bool b_reg = zb < 0;
    const int zbs = b_reg ? za : zb;                 // (a valid slot; what it returns is dropped when B.first is `bf`)
    double p[6][NP];
#pragma unroll
    for (int c0 = 0; c0 < NP; c0 += CP) {
        double az[2 * CP], av[2 * CP], lz[2 * CP], lv[2 * CP], bz[2 * CP], bv[2 * CP];
#pragma unroll
        for (int i = 0; i < 2 * CP; ++i) {
            const int d = 2 * c0 + i;
            if (d < E) {
                az[i] = C.ldSe(za, d); av[i] = C.ldSe(za + 1, d);
                lz[i] = C.ldSe(zl, d); lv[i] = C.ldSe(zl + 1, d);
                bz[i] = C.ldSe(zbs, d); bv[i] = C.ldSe(zbs + 1, d);
            }
        }
#pragma unroll
        for (int l = c0; l < c0 + CP; ++l) {
            if (l < NP) {
                double a[6] = {0., 0., 0., 0., 0., 0.};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int d = 2 * l + k, i = d - 2 * c0;
                    const double bzz = b_reg ? bf.z[d] : bz[i], bvv = b_reg ? bf.v[d] : bv[i];
                    turn_acc(az[i], av[i], o.z[d], o.v[d], a[0], a[1]);
                    turn_acc(lz[i], lv[i], o.z[d], o.v[d], a[2], a[3]);
                    turn_acc(az[i], av[i], bzz, bvv, a[4], a[5]);
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) p[j][l] = a[j];
            }
        }
        if (CP < NP) __builtin_amdgcn_sched_barrier(0);      // the next chunk's loads stay behind this chunk's arithmetic: CP bounds the registers
    }
    bool turning = false;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double sj = pair_tree<NP>(p[j]);
        turning = turning | (fwd ? (sj < 0.) : (sj > 0.));
    }
    return turning;
}

// The first form of the same tests: operands read one element at a time, a branch on `fwd` per element (one HBM round trip per
// element).  Kept for the warm-up kernel (TUNE = true), whose register allocation the batched form upsets: measured on K4's 400
// warm-up draws, 154 ms with this form against 205 ms with the batched one, while the sampling kernel gains 8 % from the batched form
// (profiles/r03z_*).
template <int NP, class LD>
NM_DEV bool l_merge_turning_streamed(const LCtx<NP, LD>& C, int za, int zl, int zb, const LPt<NP>& bf, const LPt<NP>& o, bool fwd) {
    constexpr int E = 2 * NP;
    double p[6][NP];
#pragma unroll
    for (int l = 0; l < NP; ++l) {
        double a[6] = {0., 0., 0., 0., 0., 0.};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int d = 2 * l + k;
            const double az = C.ldSe(za, d), av = C.ldSe(za + 1, d);
            const double lz = C.ldSe(zl, d), lv = C.ldSe(zl + 1, d);
            const double bz = zb >= 0 ? C.ldSe(zb, d) : bf.z[d], bv = zb >= 0 ? C.ldSe(zb + 1, d) : bf.v[d];
            if (fwd) {
                turn_acc(az, av, o.z[d], o.v[d], a[0], a[1]);
                turn_acc(lz, lv, o.z[d], o.v[d], a[2], a[3]);
                turn_acc(az, av, bz, bv, a[4], a[5]);
            } else {
                turn_acc(o.z[d], o.v[d], az, av, a[0], a[1]);
                turn_acc(o.z[d], o.v[d], lz, lv, a[2], a[3]);
                turn_acc(bz, bv, az, av, a[4], a[5]);
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) p[i][l] = a[i];
    }
    bool turning = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) turning = turning | (pair_tree<NP>(p[i]) < 0.);
    return turning;
}

// is_turning sums of one (start, end) pair of points over the chain: the two scalar_prods3 results
template <int NP>
NM_DEV void l_turn_sums(const double (&zs)[2 * NP], const double (&vs)[2 * NP], const double (&ze)[2 * NP], const double (&ve)[2 * NP],
                        double& t1, double& t2) {
    double p1[NP], p2[NP];
#pragma unroll
    for (int l = 0; l < NP; ++l) {
        double a1 = 0., a2 = 0.;
#pragma unroll
        for (int k = 0; k < 2; ++k) turn_acc(zs[2 * l + k], vs[2 * l + k], ze[2 * l + k], ve[2 * l + k], a1, a2);
        p1[l] = a1; p2[l] = a2;
    }
    t1 = pair_tree<NP>(p1); t2 = pair_tree<NP>(p2);
}
template <int NP>
NM_DEV bool l_turning(const double (&az)[2 * NP], const double (&av)[2 * NP], const double (&bz)[2 * NP], const double (&bv)[2 * NP], bool fwd) {
    double s1, s2;
    if (fwd) l_turn_sums<NP>(az, av, bz, bv, s1, s2);
    else l_turn_sums<NP>(bz, bv, az, av, s1, s2);
    return (s1 < 0.) | (s2 < 0.);
}

template <int NP, class LD>
NM_DEV bool l_merge_weights(LCtx<NP, LD>& C, double a_log_size, double b_log_size, bool is_main, double& total, bool& fatal) {
    total = llogaddexp(a_log_size, b_log_size);
    const double self_log_size = is_main ? a_log_size : total;
    if (b_log_size >= self_log_size) return true;
    const int b = C.rng.random_bool(lexp(b_log_size - self_log_size));
    if (b < 0) { fatal = true; return false; }
    return b == 1;
}
template <int NP, class LD>
NM_DEV int l_cand_to_pool(LCtx<NP, LD>& C, uint32_t& used, const double (&z)[2 * NP]) {
    const int p = (int)__builtin_ctz(~used);
    used |= 1u << p;
    C.stS(z, slot_C(C.md, p));
    return p;
}

// nuts::draw for the lane's chain (reference src/nuts.rs:281-388): the port of nuts_transition / g_transition
template <bool BATCHED_TESTS, int NP, class LD>
NM_DEV uint64_t l_transition(LCtx<NP, LD>& C, LAccept& col, DrawResult& R, double (&zc)[2 * NP]) {
    constexpr int E = 2 * NP;
    const nm_settings& s = C.P.s;
    ChainScalars& sc = C.sc;
    const int MD = C.md;
    LPt<NP> Ep, Op;
    if (sc.mm_id != sc.transform_id) {        // lazy re-whitening after the last mass-matrix update (diagonal.rs:210-221)
        double x[E], gx[E], isig[E];
        C.ldW(x, P_X); C.ldW(gx, P_GX); C.ldW(isig, P_ISIG);
#pragma unroll
        for (int d = 0; d < E; ++d) {
            const double t = __builtin_fma(-1.0, C.mu[d], x[d]);
            Ep.z[d] = isig[d] * t;
            Ep.g[d] = gx[d] * C.sig[d];
        }
        C.stW(Ep.z, P_Z); C.stW(Ep.g, P_GZ);
        sc.logdet = sc.mm_logdet;
        sc.transform_id = sc.mm_id;
    } else {
        C.ldW(Ep.z, P_Z);
        C.ldW(Ep.g, P_GZ);
    }
    // array_gaussian: the stream order is the element order.  ONE copy of the ziggurat (a loop over the elements, through the
    // staging slot; its rows beyond dim are zero since the allocation) instead of 2 NP inlined ones
    // (staged in LDS: a store to the scratch in HBM followed by the next sample's table look-up serialises on the one memory counter)
    C.draw_normals(Ep.v);
    bool micro_ = false;
    if constexpr (lkin_trait<LD>::value) micro_ = sc.kin == NM_TRAJ_MICROCANONICAL;
    if (micro_) l_normalize<NP>(Ep.v);       // initialize_trajectory (transformed_hamiltonian.rs:697-727): the momentum on the unit sphere
    C.stS(Ep.z, EDGE0_Z); C.stS(Ep.v, EDGE0_V); C.stS(Ep.g, EDGE0_G);
    const double logdet = sc.logdet;
    double ke_init;
    if (micro_) ke_init = 0.0;
    else {
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 2; ++k) acc = __builtin_fma(Ep.v[2 * l + k], Ep.v[2 * l + k], acc);
            p[l] = acc;
        }
        ke_init = 0.5 * pair_tree<NP>(p);
    }
    Ep.ke = ke_init;
    [[maybe_unused]] double left_ke = ke_init, right_ke = ke_init;   // the edges' kinetic_energy: an input of the microcanonical leapfrog only
    const double e0 = ke_init - (sc.logp + logdet);
    R.e0 = e0;
    col.register_init(e0);
    NM_LP(C, 0);
    int left_slot = 0, right_slot = 0;
    bool o_is_edge = false;
    int o_edge_sign = 0;
    uint64_t depth = 0;
    double log_size = 0.;
    int64_t left_idx = 0, right_idx = 0;
    CandRef mc = {-1, sc.logp, ke_init, 0};
    uint32_t used = 0;

    uint64_t mindepth = s.mindepth, maxdepth = s.maxdepth;
    if (s.has_target_integration_time) {
        const double q = __builtin_ceil(s.target_integration_time / sc.step_size);
        const uint64_t max_steps = q >= 18446744073709551616.0 ? ~0ull : (q > 0 ? (uint64_t)q : 0ull);
        const uint64_t fl = 63 - __builtin_clzll(max_steps | 1ull);
        const uint64_t ce = ((max_steps & (max_steps - 1)) == 0) ? fl : fl + 1;
        mindepth = fl > s.mindepth ? fl : s.mindepth;
        const uint64_t xd = ce > mindepth ? ce : mindepth;
        maxdepth = xd < s.maxdepth ? xd : s.maxdepth;
    }
    R.diverging = false; R.reached_maxdepth = false; R.has_divergence_energy_error = false; R.has_div_end = true;
    R.divergence_energy_error = 0.; R.div_start_idx = 0;
    const bool want_div = C.P.out_div_start || C.P.out_div_start_grad || C.P.out_div_end;
    bool fatal = false;
    bool in_extra = false;
    uint64_t extra_left = 0;
    int sign = 1;

    for (;;) {
        bool check;
        if (!in_extra) {
            if (!(depth < maxdepth)) { R.reached_maxdepth = true; break; }
            sign = C.rng.random_bool_std() ? 1 : -1;
            check = (s.check_turning != 0) && !(depth < mindepth);
        } else {
            if (extra_left == 0) break;
            extra_left -= 1;
            check = false;
        }
        const bool fwd = sign > 0;
        const int64_t edge_idx = fwd ? right_idx : left_idx;
        const uint64_t nleaf = 1ull << depth;
        const uint32_t used_before = used;
        const double epsilon = (double)sign * sc.step_size * 1.0;
        int stop = STOP_NONE;
        double sub_log_size = 0.;
        CandRef sub_cand = {-2, 0., 0., 0};
        const bool reuse_edge = o_is_edge && o_edge_sign == sign;
        o_is_edge = false;

#define NM_L_ACCOUNT(START, PT, WOUT)                                                                      \
        {                                                                                                 \
            const double energy_ = (PT).ke - ((PT).logp + logdet);                                        \
            const double err_ = energy_ - e0;                                                             \
            if (l_bad_energy(C, err_, s.max_energy_error)) {                                              \
                col.register_divergent();                                                                 \
                R.diverging = true; R.has_divergence_energy_error = true; R.divergence_energy_error = err_; \
                R.div_start_idx = (PT).idx - (int64_t)sign;                                               \
                if (want_div) { C.stS((START).z, slot_F(0)); C.stS((PT).z, slot_F(0) + 1); }              \
                stop = STOP_DIVERGING;                                                                    \
            } else {                                                                                      \
                col.register_ok(energy_);                                                                 \
                WOUT = -err_;                                                                             \
            }                                                                                             \
        }

        NM_LP(C, 1);
        if (depth == 0) {
            l_leapfrog(C, Ep, Op, epsilon);
            NM_LP(C, 2);
            Op.idx = edge_idx + (int64_t)sign;
            NM_L_ACCOUNT(Ep, Op, sub_log_size)
            sub_cand = {-2, Op.logp, Op.ke, Op.idx};
            NM_LP(C, 3);
        } else {
            if (!reuse_edge) {
                C.ld_edge(Op, fwd ? right_slot : left_slot);
                if constexpr (lkin_trait<LD>::value) Op.ke = fwd ? right_ke : left_ke;
            }
            NM_LP(C, 1);
            for (uint64_t n = 0; n < nleaf; n += 2) {
                double wE = 0., wO = 0.;
                l_leapfrog(C, Op, Ep, epsilon);
                NM_LP(C, 2);
                Ep.idx = edge_idx + (int64_t)sign * (int64_t)(n + 1);
                NM_L_ACCOUNT(Op, Ep, wE)
                NM_LP(C, 3);
                if (stop != STOP_NONE) break;
                l_leapfrog(C, Ep, Op, epsilon);
                NM_LP(C, 2);
                Op.idx = edge_idx + (int64_t)sign * (int64_t)(n + 2);
                NM_L_ACCOUNT(Ep, Op, wO)
                NM_LP(C, 3);
                if (stop != STOP_NONE) break;
                const uint64_t nn = n + 1;
                const int t = (int)__builtin_ctzll(~nn);
                uint32_t turn_bits = 0;
                if (check) {
                    if (l_turning<NP>(Ep.z, Ep.v, Op.z, Op.v, fwd)) turn_bits |= 2u;
                    for (int k = 2; k <= t && turn_bits == 0; ++k) {
                        // (A.first, B.last) (A.last, B.last) (A.first, B.first) in generation order  [src/nuts.rs:143-161]
                        const uint64_t a_first = nn + 1 - (1ull << k);
                        const int fa = a_first == 0 ? (int)depth : (int)__builtin_ctzll(a_first);
                        const bool tk = BATCHED_TESTS ? l_merge_turning(C, slot_F(fa), slot_L(MD, k - 1), k == 2 ? -1 : slot_F(k - 1), Ep, Op, fwd)
                                                      : l_merge_turning_streamed(C, slot_F(fa), slot_L(MD, k - 1), k == 2 ? -1 : slot_F(k - 1), Ep, Op, fwd);
                        if (tk) turn_bits |= 1u << k;
                    }
                }
                NM_LP(C, 4);
                {
                    double total;
                    const bool take = l_merge_weights(C, wE, wO, false, total, fatal);
                    sub_cand = take ? CandRef{-2, Op.logp, Op.ke, Op.idx} : CandRef{-3, Ep.logp, Ep.ke, Ep.idx};
                    sub_log_size = total;
                    if (fatal) { stop = STOP_FATAL; break; }
                    if (turn_bits & 2u) { stop = STOP_TURNING; break; }
                }
                for (int k = 2; k <= t; ++k) {
                    double a_log_size; CandRef ac;
                    C.pend.get(k - 1, a_log_size, ac);
                    double total;
                    const bool take = l_merge_weights(C, a_log_size, sub_log_size, false, total, fatal);
                    if (take) {
                        used &= ~(1u << ac.slot);
                    } else {
                        if (sub_cand.slot >= 0) used &= ~(1u << sub_cand.slot);
                        sub_cand = ac;
                    }
                    sub_log_size = total;
                    if (fatal) { stop = STOP_FATAL; break; }
                    if ((turn_bits >> k) & 1u) { stop = STOP_TURNING; break; }
                }
                NM_LP(C, 5);
                if (stop != STOP_NONE) break;
                if ((n & 3) == 0 && depth > 1) {
                    const int fs = slot_F(n == 0 ? (int)depth : (int)__builtin_ctzll(n));
                    C.stS(Ep.z, fs);
                    C.stS(Ep.v, fs + 1);
                }
                if (n + 2 < nleaf) {
                    C.stS(Op.z, slot_L(MD, t)); C.stS(Op.v, slot_L(MD, t) + 1);
                    if (sub_cand.slot == -2) sub_cand.slot = l_cand_to_pool(C, used, Op.z);
                    else if (sub_cand.slot == -3) sub_cand.slot = l_cand_to_pool(C, used, Ep.z);
                    C.pend.put(t, sub_log_size, sub_cand);
                }
                NM_LP(C, 6);
            }
        }
#undef NM_L_ACCOUNT
        if (stop == STOP_FATAL) { fatal = true; break; }
        if (stop == STOP_DIVERGING) { used = used_before; break; }
        if (stop == STOP_TURNING) {
            used = used_before;
            if (!in_extra) { in_extra = true; extra_left = s.extra_doublings; }
            continue;
        }
        // top-level U-turn tests of the finished sub-tree (last leaf O) against the main tree (src/nuts.rs:143-161)
        bool turning = false;
        if (check) {
            if (depth == 0) turning = l_turning<NP>(Ep.z, Ep.v, Op.z, Op.v, fwd);
            else    // fwd: (tree.left, O) (tree.right, O) (tree.left, other.left); else (O, tree.right) (O, tree.left) (other.right, tree.right):
                    // the merge tests' shape with A = the main tree (first = its far end), B.first = the sub-tree's first point
                turning = BATCHED_TESTS ? l_merge_turning(C, EDGE0_Z + 3 * (fwd ? left_slot : right_slot), EDGE0_Z + 3 * (fwd ? right_slot : left_slot),
                                                          depth == 1 ? -1 : slot_F((int)depth), Ep, Op, fwd)
                                        : l_merge_turning_streamed(C, EDGE0_Z + 3 * (fwd ? left_slot : right_slot), EDGE0_Z + 3 * (fwd ? right_slot : left_slot),
                                                                   depth == 1 ? -1 : slot_F((int)depth), Ep, Op, fwd);
        }
        double total;
        const bool take = l_merge_weights(C, log_size, sub_log_size, true, total, fatal);
        if (fatal) break;
        if (take) {
            if (mc.slot >= 0) used &= ~(1u << mc.slot);
            if (sub_cand.slot == -2) sub_cand.slot = l_cand_to_pool(C, used, Op.z);
            else if (sub_cand.slot == -3) sub_cand.slot = l_cand_to_pool(C, used, Ep.z);
            mc = sub_cand;
        } else if (sub_cand.slot >= 0) {
            used &= ~(1u << sub_cand.slot);
        }
        const bool more = in_extra ? extra_left > 0 : (turning ? s.extra_doublings > 0 : depth + 1 < maxdepth);
        if (more) {
            int ns = fwd ? right_slot : left_slot;
            const int other_side = fwd ? left_slot : right_slot;
            if (ns == 0) ns = other_side == 1 ? 2 : 1;
            C.stS(Op.z, EDGE0_Z + 3 * ns); C.stS(Op.v, EDGE0_V + 3 * ns); C.stS(Op.g, EDGE0_G + 3 * ns);
            if (fwd) right_slot = ns; else left_slot = ns;
            o_is_edge = true; o_edge_sign = sign;
        }
        if (fwd) right_idx = Op.idx; else left_idx = Op.idx;
        if constexpr (lkin_trait<LD>::value) { if (fwd) right_ke = Op.ke; else left_ke = Op.ke; }
        depth += 1;
        log_size = total;
        if (turning && !in_extra) { in_extra = true; extra_left = s.extra_doublings; }
        NM_LP(C, 7);
    }
    NM_LP(C, 7);
    R.depth = depth;
    R.chosen = mc;
    if (fatal) return NM_CHAIN_LOGP_FATAL;
    if (mc.slot >= 0) C.ldS(zc, slot_C(MD, mc.slot));
    return NM_CHAIN_OK;
}

// ---- warm-up (lane forms of the adaptation in nuts_kernels.hpp / nuts_group_impl.hpp) ----
NM_DEV void l_stepsize_adapt_reset(ChainScalars& sc, const nm_settings& s, double initial_step) {
    sc.log_step = llog(initial_step);
    if (s.step_size_method == NM_STEP_ADAM) { sc.adam_m = 0.; sc.adam_v = 0.; sc.adam_t = 0; return; }
    sc.log_step_adapted = sc.log_step;
    sc.hbar = 0.;
    sc.mu = llog(10. * initial_step);
    sc.da_count = 1;
}
template <int NP, class LD>
NM_DEV void l_update_stepsize(LCtx<NP, LD>& C, bool use_best_guess) {
    const nm_settings& s = C.P.s;
    const double step = s.step_size_method == NM_STEP_FIXED ? s.fixed_step_size
                      : s.step_size_method == NM_STEP_ADAM ? lexp(C.sc.log_step)
                      : (use_best_guess ? lexp(C.sc.log_step_adapted) : lexp(C.sc.log_step));
    if (s.has_jitter) {
        const double v12 = u2d((C.rng.next_u64() >> 12) | 0x3ff0000000000000ull);
        const double j = (v12 - 1.0) * C.P.jitter_scale + C.P.jitter_low;
        C.sc.step_size = step * j;
    } else {
        C.sc.step_size = step;
    }
}
template <int NP, class LD>
NM_DEV void l_update_estimator(LCtx<NP, LD>& C, bool late) {
    const nm_settings& s = C.P.s;
    if (s.step_size_method == NM_STEP_FIXED) return;
    ChainScalars& sc = C.sc;
    const double accept_stat = late ? sc.last_sym_mean_tree_accept : sc.last_mean_tree_accept;
    if (s.step_size_method == NM_STEP_ADAM) {
        const double gradient = accept_stat - s.target_accept;
        sc.adam_t += 1;
        sc.adam_m = s.adam_beta1 * sc.adam_m + (1.0 - s.adam_beta1) * gradient;
        sc.adam_v = s.adam_beta2 * sc.adam_v + (1.0 - s.adam_beta2) * gradient * gradient;
        const double m_hat = sc.adam_m / (1.0 - powi_rs(s.adam_beta1, (int32_t)sc.adam_t));
        const double v_hat = sc.adam_v / (1.0 - powi_rs(s.adam_beta2, (int32_t)sc.adam_t));
        sc.log_step += s.adam_learning_rate * m_hat / (__builtin_sqrt(v_hat) + s.adam_epsilon);
        return;
    }
    const double w = 1. / ((double)sc.da_count + s.da_t0);
    sc.hbar = (1. - w) * sc.hbar + w * (s.target_accept - accept_stat);
    sc.log_step = sc.mu - sc.hbar * __builtin_sqrt((double)sc.da_count) / s.da_gamma;
    sc.log_step = fmin_rs(sc.log_step, C.P.ln_max_step);
    const double mk = lexp(-s.da_k * llog((double)sc.da_count));
    sc.log_step_adapted = mk * sc.log_step + (1. - mk) * sc.log_step_adapted;
    sc.da_count += 1;
}
template <int E>
NM_DEV void l_running_variance_add(double (&mean)[E], double (&var)[E], uint64_t new_count, const double (&value)[E]) {
    if (new_count == 1) {
#pragma unroll
        for (int d = 0; d < E; ++d) mean[d] = value[d];
        return;
    }
    const double diff_scale = 1.0 / (double)new_count;
#pragma unroll
    for (int d = 0; d < E; ++d) {
        const double diff = value[d] - mean[d];
        mean[d] = mean[d] + diff * diff_scale;
        var[d] = var[d] + diff * diff;
    }
}
template <int NP, class LD>
NM_DEV void l_commit_mass_matrix(LCtx<NP, LD>& C, const double (&sig)[2 * NP], const double (&isig)[2 * NP], const double (&mu)[2 * NP]) {
    constexpr int E = 2 * NP;
    C.stW(sig, P_SIG); C.stW(isig, P_ISIG); C.stW(mu, P_MU);
#pragma unroll
    for (int d = 0; d < E; ++d) { C.sig[d] = sig[d]; C.mu[d] = mu[d]; }
    double p[NP];
#pragma unroll
    for (int l = 0; l < NP; ++l) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int d = 2 * l + k;
            const bool valid = d < C.dim;
            acc = acc + (valid ? llog(valid ? isig[d] : 1.0) : 0.0);
        }
        p[l] = acc;
    }
    C.sc.mm_logdet = pair_tree<NP>(p);
    C.sc.mm_id += 1;
}
template <int NP, class LD>
NM_DEV bool l_mass_matrix_adapt(LCtx<NP, LD>& C, const double (&dm)[2 * NP], const double (&dv)[2 * NP], const double (&gm)[2 * NP], const double (&gv)[2 * NP]) {
    constexpr int E = 2 * NP;
    if (C.sc.cnt_fg < 3) return false;
    double sig[E], isig[E], mu[E];
    C.ldW(isig, P_ISIG);
    if (C.P.s.use_grad_based_estimate) {
#pragma unroll
        for (int d = 0; d < E; ++d) {
            const bool valid = d < C.dim;
            double val = __builtin_sqrt(dv[d] / gv[d]);
            double sd = C.sig[d], isd = isig[d];
            if (!(!is_finite(val) | (val == 0.0))) {
                val = clampd(val, 1e-20, 1e20);
                sd = __builtin_sqrt(val);
                isd = __builtin_sqrt(1.0 / val);
            }
            const double var = sd * sd;
            double mean = var * gm[d];
            mean = __builtin_fma(1.0, dm[d], mean);
            sig[d] = valid ? sd : 0.0;
            isig[d] = valid ? isd : 0.0;
            mu[d] = valid ? mean : 0.0;
        }
    } else {
        const double scale = 1.0 / (double)C.sc.cnt_fg;
#pragma unroll
        for (int d = 0; d < E; ++d) {
            const bool valid = d < C.dim;
            const double dd = dv[d] * scale;
            double sd = C.sig[d], isd = isig[d];
            if (!(!is_finite(dd) | (dd == 0.0))) {
                const double val = clampd(dd, 1e-20, 1e20);
                sd = __builtin_sqrt(val);
                isd = __builtin_sqrt(1.0 / val);
            }
            sig[d] = valid ? sd : 0.0;
            isig[d] = valid ? isd : 0.0;
            mu[d] = valid ? dm[d] : 0.0;
        }
    }
    l_commit_mass_matrix(C, sig, isig, mu);
    return true;
}
// stepsize::Strategy::init (src/stepsize/adapt.rs:91-199): the step-size search at x, after the first mass-matrix update
template <int NP, class LD>
NM_DEV uint64_t l_stepsize_init(LCtx<NP, LD>& C, const double (&x)[2 * NP]) {
    constexpr int E = 2 * NP;
    const nm_settings& s = C.P.s;
    if (s.step_size_method == NM_STEP_FIXED) { C.sc.step_size = s.fixed_step_size; return NM_CHAIN_OK; }
    LPt<NP> st;
    {   // Hamiltonian::init_state (transformed_hamiltonian.rs:640-661, check_all :310-324)
        double gx[E], isig[E];
        st.logp = C.dens.eval(x, gx, C.dim);
        C.ldW(isig, P_ISIG);
        bool ok = true;
#pragma unroll
        for (int d = 0; d < E; ++d) {
            const double t = __builtin_fma(-1.0, C.mu[d], x[d]);
            st.z[d] = isig[d] * t;
            st.g[d] = gx[d] * C.sig[d];
            const bool valid = d < C.dim;
            ok = ok && (!valid || (is_finite(st.z[d]) && is_finite(st.g[d]) && st.g[d] != 0.0 && is_finite(gx[d]) && is_finite(x[d])));
        }
        if (!ok) return NM_CHAIN_BAD_INIT;
    }
    const double logdet = C.sc.mm_logdet;
    double ke0;
    {
        C.draw_normals(st.v);
        bool micro_ = false;
        if constexpr (lkin_trait<LD>::value) micro_ = C.sc.kin == NM_TRAJ_MICROCANONICAL;
        if (micro_) l_normalize<NP>(st.v);
        C.stS(st.v, STAGE_V);
        ke0 = micro_ ? 0.0 : l_kinetic<NP>(st.v);
    }
    st.ke = ke0;
    const double e0 = ke0 - (st.logp + logdet);
    LAccept col;
    C.sc.step_size = s.initial_step;
    int dir = 0;
    for (int it = 0; it < 101; ++it) {
        LPt<NP> o;
        const int sign = it == 0 ? 1 : dir;
        col.register_init(e0);
        l_leapfrog(C, st, o, (double)sign * C.sc.step_size * 1.0);
        const double energy = o.ke - (o.logp + logdet);
        const double err = energy - e0;
        if (l_bad_energy(C, err, 1000.0)) {
            if (it > 0) C.sc.step_size = s.initial_step;
            return NM_CHAIN_OK;
        }
        col.register_ok(energy);
        const double accept = col.mean();
        if (it == 0) { dir = accept > s.target_accept ? 1 : -1; continue; }
        if (dir > 0) {
            if ((accept <= s.target_accept) | (C.sc.step_size > 1e5)) { l_stepsize_adapt_reset(C.sc, s, C.sc.step_size); return NM_CHAIN_OK; }
            C.sc.step_size *= 2.;
        } else {
            if ((accept >= s.target_accept) | (C.sc.step_size < 1e-10)) { l_stepsize_adapt_reset(C.sc, s, C.sc.step_size); return NM_CHAIN_OK; }
            C.sc.step_size /= 2.;
        }
    }
    C.sc.step_size = s.initial_step;
    return NM_CHAIN_OK;
}
// GlobalStrategy::adapt (src/adapt_strategy.rs:121-222); x, gx = the chosen draw
template <bool TUNE, int NP, class LD>
NM_DEV uint64_t l_adapt(LCtx<NP, LD>& C, LAccept& col, bool is_good, const double (&x)[2 * NP], const double (&gx)[2 * NP]) {
    constexpr int E = 2 * NP;
    const nm_settings& s = C.P.s;
    ChainScalars& sc = C.sc;
    const uint64_t draw = sc.draw_count;
    sc.last_mean_tree_accept = col.mean();
    sc.last_sym_mean_tree_accept = col.mean_sym();
    sc.last_n_steps = col.count;
    sc.last_max_energy_error = col.max_energy_error;
    if (!TUNE || draw >= s.num_tune) {     // the sampling kernel (TUNE = false) is only launched once every chain is there
        l_update_stepsize(C, true);
        sc.tuning = 0;
        return NM_CHAIN_OK;
    }
    if (draw < C.P.final_step_size_window) {
        const bool is_early = draw < C.P.early_end;
        if (!is_early && draw == C.P.early_end)
            sc.current_window_size = sc.current_window_size > sc.cnt_bg ? sc.current_window_size : sc.cnt_bg;
        const uint64_t switch_freq = is_early ? s.early_mass_matrix_switch_freq : sc.current_window_size;
        double fdm[E], fdv[E], fgm[E], fgv[E], bdm[E], bdv[E], bgm[E], bgv[E];
        C.ldW(fdm, E_DM); C.ldW(fdv, E_DV); C.ldW(fgm, E_GM); C.ldW(fgv, E_GV);
        C.ldW(bdm, B_DM); C.ldW(bdv, B_DV); C.ldW(bgm, B_GM); C.ldW(bgv, B_GV);
        bool dirty = false;
        if (is_good) {
            sc.cnt_fg += 1;
            sc.cnt_bg += 1;
            l_running_variance_add<E>(fdm, fdv, sc.cnt_fg, x);
            l_running_variance_add<E>(fgm, fgv, sc.cnt_fg, gx);
            l_running_variance_add<E>(bdm, bdv, sc.cnt_bg, x);
            l_running_variance_add<E>(bgm, bgv, sc.cnt_bg, gx);
            dirty = true;
        }
        const bool could_switch = sc.cnt_bg >= switch_freq;
        uint64_t next_window_size;
        if (is_early) next_window_size = s.early_mass_matrix_switch_freq;
        else {
            const double gv = (double)sc.current_window_size * s.mass_matrix_window_growth;
            const double fl = __builtin_floor(gv);
            const uint64_t grown = (uint64_t)((gv - fl >= 0.5) ? fl + 1.0 : fl);
            next_window_size = sc.current_window_size + 1 > grown ? sc.current_window_size + 1 : grown;
        }
        const bool is_late = next_window_size + draw > C.P.final_step_size_window;
        bool force_update = false;
        if (could_switch && !is_late) {
#pragma unroll
            for (int d = 0; d < E; ++d) {
                fdm[d] = bdm[d]; fdv[d] = bdv[d]; fgm[d] = bgm[d]; fgv[d] = bgv[d];
                bdm[d] = 0.0; bdv[d] = 0.0; bgm[d] = 0.0; bgv[d] = 0.0;
            }
            sc.cnt_fg = sc.cnt_bg;
            sc.cnt_bg = 0;
            force_update = true;
            dirty = true;
            if (!is_early) sc.current_window_size = next_window_size;
        }
        if (dirty) {
            C.stW(bdm, B_DM); C.stW(bdv, B_DV); C.stW(bgm, B_GM); C.stW(bgv, B_GV);
            C.stW(fdm, E_DM); C.stW(fdv, E_DV); C.stW(fgm, E_GM); C.stW(fgv, E_GV);
        }
        bool did_change = false;
        if (force_update | (draw - sc.last_update >= s.mass_matrix_update_freq)) did_change = l_mass_matrix_adapt(C, fdm, fdv, fgm, fgv);
        if (did_change) sc.last_update = draw;
        l_update_estimator(C, is_late);
        if (did_change & (sc.has_initial_mass_matrix != 0)) {
            sc.has_initial_mass_matrix = 0;
            return l_stepsize_init(C, x);
        }
        l_update_stepsize(C, false);
        return NM_CHAIN_OK;
    }
    l_update_estimator(C, true);
    l_update_stepsize(C, draw == s.num_tune - 1);
    return NM_CHAIN_OK;
}

template <int NP, class LD>
NM_DEV void l_write_row(LCtx<NP, LD>& C, double* base, size_t row, const double (&t)[2 * NP]) {
    if (!base) return;
    double* dst = base + row;
#pragma unroll
    for (int d = 0; d < 2 * NP; ++d) if (d < C.dim) dst[d] = t[d];
}
// DivergenceInfo.{start_location, start_gradient, end_location} (transformed_hamiltonian.rs:590-604), as emit_divergence_vectors
template <int NP, class LD>
NM_DEV void l_emit_divergence_vectors(LCtx<NP, LD>& C, int64_t start_idx, size_t row) {
    constexpr int E = 2 * NP;
    const KParams& P = C.P;
    double x[E], gx[E], zt[E];
    if (start_idx == 0) {
        C.ldW(x, P_X); C.ldW(gx, P_GX);
    } else {
        C.ldS(zt, slot_F(0));
#pragma unroll
        for (int d = 0; d < E; ++d) x[d] = __builtin_fma(1.0, C.mu[d], zt[d] * C.sig[d]);
        (void)C.dens.eval(x, gx, C.dim);
    }
    l_write_row(C, P.out_div_start, row, x);
    l_write_row(C, P.out_div_start_grad, row, gx);
    C.ldS(zt, slot_F(0) + 1);
#pragma unroll
    for (int d = 0; d < E; ++d) x[d] = __builtin_fma(1.0, C.mu[d], zt[d] * C.sig[d]);
    l_write_row(C, P.out_div_end, row, x);
}

// NutsChain::draw (reference src/chain.rs:151-188) + the scalar statistics of expanded_draw (:190-232)
template <bool TUNE, int NP, class LD>
NM_DEV void l_chain_draw(LCtx<NP, LD>& C, uint64_t chain, uint64_t t_out) {
    constexpr int E = 2 * NP;
    const KParams& P = C.P;
    ChainScalars& sc = C.sc;
    LAccept col;
    DrawResult R;
    double x[E], gx[E], z[E], gz[E];
    NM_LP(C, 11);
    const uint64_t st = l_transition<!TUNE>(C, col, R, z);
    NM_LP(C, 7);
    nm_draw_stats out;
    out.draw = sc.draw_count; out.chain = P.chain_id_offset + chain;
    if (st != NM_CHAIN_OK) {
        sc.status = st;
        if (P.out_stats) {
            nm_draw_stats zz = {};
            zz.draw = sc.draw_count; zz.chain = P.chain_id_offset + chain; zz.chain_status = st;
            P.out_stats[t_out * P.n_chains + chain] = zz;
        }
        return;
    }
    const size_t row = (size_t)(t_out * P.n_chains + chain) * P.dim;
    if (R.diverging && (P.out_div_start || P.out_div_start_grad || P.out_div_end))
        l_emit_divergence_vectors(C, R.div_start_idx, row);          // before P_X / P_GX take the new draw
    if (R.chosen.slot == -1 && !sc.px_stale) {
        C.ldW(x, P_X); C.ldW(gx, P_GX);
        C.ldW(z, P_Z); C.ldW(gz, P_GZ);
    } else {
        if (R.chosen.slot == -1) C.ldW(z, P_Z);
#pragma unroll
        for (int d = 0; d < E; ++d) {
            const double tt = z[d] * C.sig[d];
            x[d] = __builtin_fma(1.0, C.mu[d], tt);
        }
        (void)C.dens.eval(x, gx, C.dim);
#pragma unroll
        for (int d = 0; d < E; ++d) gz[d] = gx[d] * C.sig[d];
        const bool need_x = sc.tuning || t_out + 1 == P.n_draws || P.out_div_start || P.out_div_start_grad;
        if (need_x) { C.stW(x, P_X); C.stW(gx, P_GX); }
        sc.px_stale = need_x ? 0 : 1;
        C.stW(z, P_Z); C.stW(gz, P_GZ);
        sc.logp = R.chosen.logp;
    }
    const int64_t idx = R.chosen.idx;
    NM_LP(C, 8);
    l_write_row(C, P.out_positions, row, x);
    l_write_row(C, P.out_gradient, row, gx);                     // PointStats (transformed_hamiltonian.rs:122-157)
    l_write_row(C, P.out_tpos, row, z);
    l_write_row(C, P.out_tgrad, row, gz);
    double fd;
    {
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int d = 2 * l + k; acc = acc + (z[d] + gz[d]) * (z[d] + gz[d]); }
            p[l] = acc;
        }
        fd = pair_tree<NP>(p);
    }
    const double energy = R.chosen.ke - (R.chosen.logp + sc.logdet);
    const int64_t trans_id = sc.transform_id;
    sc.total_steps += col.count;
    const bool is_good = R.diverging ? ((idx < 0 ? -idx : idx) > 4) : (idx != 0);     // DrawGradCollector (adapt/diagonal.rs:73-83)
    NM_LP(C, 9);
    const uint64_t ast = l_adapt<TUNE>(C, col, is_good, x, gx);
    NM_LP(C, 10);
    if (ast != NM_CHAIN_OK) sc.status = ast;
    out.depth = R.depth; out.maxdepth_reached = R.reached_maxdepth; out.diverging = R.diverging;
    out.tuning = sc.tuning; out.n_steps = sc.last_n_steps;
    out.index_in_trajectory = idx; out.transformation_index = trans_id;
    out.step_size = sc.step_size;
    out.step_size_bar = P.s.step_size_method == NM_STEP_FIXED ? P.s.fixed_step_size
                      : P.s.step_size_method == NM_STEP_ADAM ? lexp(sc.log_step) : lexp(sc.log_step_adapted);
    out.mean_tree_accept = sc.last_mean_tree_accept; out.mean_tree_accept_sym = sc.last_sym_mean_tree_accept;
    out.max_energy_error = sc.last_max_energy_error;
    out.logp = R.chosen.logp; out.energy = energy; out.energy_error = energy - R.e0;
    out.fisher_distance = fd;
    out.divergence_energy_error = (R.diverging && R.has_divergence_energy_error) ? R.divergence_energy_error : __builtin_nan("");
    out.chain_status = ast;
    out.transformation_update_id = -1;
    out.num_eigenvalues = 0;
    out.energy_change = __builtin_nan(""); out.average_step_size = __builtin_nan("");
    if (sc.mm_id != sc.stats_last_id) {                         // DiagMassMatrix::extract_stats (transform/diagonal.rs:48-70)
        out.transformation_update_id = sc.mm_id;
        l_write_row(C, P.out_mm_inv, row, C.sig);
        l_write_row(C, P.out_mm_mu, row, C.mu);
    }
    sc.stats_last_id = sc.mm_id;
    if (P.out_stats) P.out_stats[t_out * P.n_chains + chain] = out;
    sc.draw_count += 1;
    NM_LP(C, 11);
}

// =================================================================================================================
// UNSYNCHRONISED DRAWS (round 4).  In nuts_lane_draw_kernel the 64 chains of a wavefront start every draw together:
// a lane whose tree has ended idles until the wavefront's longest tree of that draw is done — in the warm-up, where step sizes and so
// tree depths differ most, two thirds of the leapfrog slots are idle (23 slots per draw for a mean of 7.7 leaves per lane).
// Here a lane's transition is a STATE MACHINE around ONE leapfrog site: every round each lane that is inside a tree takes one leapfrog
// of ITS tree (whatever doubling, direction or leaf it is at) and advances its own bookkeeping.  Draws begin and end at EPOCH boundaries
// (every NM_LANE_EPOCH rounds): the begin / end phases — momentum refresh, the chosen point's recomputation, adaptation, the statistics
// row: as long as several leaves — run once per epoch for all lanes that are ready instead of once per round for whoever is ready.
// Same arithmetic per chain, same generator stream per chain: the same bits as every other kernel.
// =================================================================================================================
#ifndef NM_LANE_EPOCH
#define NM_LANE_EPOCH 4
#endif
enum LanePhase : int { LP_BEGIN = 0, LP_LEAF = 1, LP_END = 2, LP_DONE = 3 };

template <int NP>
struct LTree {                       // the locals of l_transition, kept across rounds
    double e0, logdet, log_size, sub_log_size, wE, eps;
    double E_logp, E_ke; int64_t E_idx;
    int left_slot, right_slot, o_edge_sign, sign, stop;
    bool o_is_edge, in_extra, check, fwd;
    int64_t left_idx, right_idx, edge_idx;
    uint64_t depth, mindepth, maxdepth, extra_left, n, nleaf;
    CandRef mc, sub_cand;
    uint32_t used, used_before;
};

// head of the doubling loop (src/nuts.rs:330-348); returns false when the tree is complete
template <int NP, class LD>
NM_DEV bool l_doubling_begin(LCtx<NP, LD>& C, LTree<NP>& T, DrawResult& R, LPt<NP>& cur) {
    const nm_settings& s = C.P.s;
    if (!T.in_extra) {
        if (!(T.depth < T.maxdepth)) { R.reached_maxdepth = true; return false; }
        T.sign = C.rng.random_bool_std() ? 1 : -1;
        T.check = (s.check_turning != 0) && !(T.depth < T.mindepth);
    } else {
        if (T.extra_left == 0) return false;
        T.extra_left -= 1;
        T.check = false;
    }
    T.fwd = T.sign > 0;
    T.edge_idx = T.fwd ? T.right_idx : T.left_idx;
    T.nleaf = 1ull << T.depth;
    T.n = 0;
    T.used_before = T.used;
    T.eps = (double)T.sign * C.sc.step_size * 1.0;
    T.stop = STOP_NONE;
    T.sub_log_size = 0.;
    T.sub_cand = CandRef{-2, 0., 0., 0};
    const bool reuse_edge = T.o_is_edge && T.o_edge_sign == T.sign;
    T.o_is_edge = false;
    if (T.depth > 0 && !reuse_edge) C.ld_edge(cur, T.fwd ? T.right_slot : T.left_slot);     // (depth 0: cur is the initial point)
    return true;
}

// the start of a draw (the prologue of l_transition)
template <int NP, class LD>
NM_DEV void l_draw_begin(LCtx<NP, LD>& C, LTree<NP>& T, LAccept& col, DrawResult& R, LPt<NP>& cur) {
    constexpr int E = 2 * NP;
    const nm_settings& s = C.P.s;
    ChainScalars& sc = C.sc;
    if (sc.mm_id != sc.transform_id) {        // lazy re-whitening after the last mass-matrix update (diagonal.rs:210-221)
        double x[E], gx[E], isig[E];
        C.ldW(x, P_X); C.ldW(gx, P_GX); C.ldW(isig, P_ISIG);
#pragma unroll
        for (int d = 0; d < E; ++d) {
            const double t = __builtin_fma(-1.0, C.mu[d], x[d]);
            cur.z[d] = isig[d] * t;
            cur.g[d] = gx[d] * C.sig[d];
        }
        C.stW(cur.z, P_Z); C.stW(cur.g, P_GZ);
        sc.logdet = sc.mm_logdet;
        sc.transform_id = sc.mm_id;
    } else {
        C.ldW(cur.z, P_Z);
        C.ldW(cur.g, P_GZ);
    }
    C.draw_normals(cur.v);
    C.stS(cur.z, EDGE0_Z); C.stS(cur.v, EDGE0_V); C.stS(cur.g, EDGE0_G);
    T.logdet = sc.logdet;
    double ke_init;
    {
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 2; ++k) acc = __builtin_fma(cur.v[2 * l + k], cur.v[2 * l + k], acc);
            p[l] = acc;
        }
        ke_init = 0.5 * pair_tree<NP>(p);
    }
    T.e0 = ke_init - (sc.logp + T.logdet);
    R.e0 = T.e0;
    col.register_init(T.e0);
    T.left_slot = 0; T.right_slot = 0; T.o_is_edge = false; T.o_edge_sign = 0;
    T.depth = 0; T.log_size = 0.; T.left_idx = 0; T.right_idx = 0;
    T.mc = CandRef{-1, sc.logp, ke_init, 0};
    T.used = 0;
    uint64_t mindepth = s.mindepth, maxdepth = s.maxdepth;
    if (s.has_target_integration_time) {
        const double q = __builtin_ceil(s.target_integration_time / sc.step_size);
        const uint64_t max_steps = q >= 18446744073709551616.0 ? ~0ull : (q > 0 ? (uint64_t)q : 0ull);
        const uint64_t fl = 63 - __builtin_clzll(max_steps | 1ull);
        const uint64_t ce = ((max_steps & (max_steps - 1)) == 0) ? fl : fl + 1;
        mindepth = fl > s.mindepth ? fl : s.mindepth;
        const uint64_t xd = ce > mindepth ? ce : mindepth;
        maxdepth = xd < s.maxdepth ? xd : s.maxdepth;
    }
    T.mindepth = mindepth; T.maxdepth = maxdepth;
    R.diverging = false; R.reached_maxdepth = false; R.has_divergence_energy_error = false; R.has_div_end = true;
    R.divergence_energy_error = 0.; R.div_start_idx = 0;
    T.in_extra = false; T.extra_left = 0; T.sign = 1;
}

// One leaf of the lane's tree has been integrated: prv -> cur.  Its accounting, the sub-tree merges it completes, the end of its
// doubling, the head of the next one.  Returns false when the tree has ended (R.chosen is set; `fatal`: the chain stops).
template <bool BATCHED_TESTS, int NP, class LD>
NM_DEV bool l_leaf_done(LCtx<NP, LD>& C, LTree<NP>& T, LAccept& col, DrawResult& R, LPt<NP>& prv, LPt<NP>& cur, bool& fatal) {
    const nm_settings& s = C.P.s;
    const int MD = C.md;
    const bool want_div = C.P.out_div_start || C.P.out_div_start_grad || C.P.out_div_end;
    const uint64_t n = T.n;
    cur.idx = T.edge_idx + (int64_t)T.sign * (int64_t)(n + 1);
    double w = 0.;
    {
        const double energy_ = cur.ke - (cur.logp + T.logdet);
        const double err_ = energy_ - T.e0;
        if ((err_ > s.max_energy_error) | !is_finite(err_)) {
            col.register_divergent();
            R.diverging = true; R.has_divergence_energy_error = true; R.divergence_energy_error = err_;
            R.div_start_idx = cur.idx - (int64_t)T.sign;
            if (want_div) { C.stS(prv.z, slot_F(0)); C.stS(cur.z, slot_F(0) + 1); }
            T.stop = STOP_DIVERGING;
        } else {
            col.register_ok(energy_);
            w = -err_;
        }
    }
    bool over = false;                 // the doubling is over (stopped, or its last leaf is done)
    if (T.depth == 0) {
        if (T.stop == STOP_NONE) { T.sub_log_size = w; T.sub_cand = CandRef{-2, cur.logp, cur.ke, cur.idx}; }
        over = true;
    } else if ((n & 1) == 0) {
        T.wE = w; T.E_logp = cur.logp; T.E_ke = cur.ke; T.E_idx = cur.idx;
        if (T.stop == STOP_NONE) { T.n = n + 1; return true; }
        over = true;
    } else {
        if (T.stop == STOP_NONE) {
            const uint64_t nn = n;
            const int t = (int)__builtin_ctzll(~nn);
            uint32_t turn_bits = 0;
            if (T.check) {
                if (l_turning<NP>(prv.z, prv.v, cur.z, cur.v, T.fwd)) turn_bits |= 2u;
                for (int k = 2; k <= t && turn_bits == 0; ++k) {
                    const uint64_t a_first = nn + 1 - (1ull << k);
                    const int fa = a_first == 0 ? (int)T.depth : (int)__builtin_ctzll(a_first);
                    const bool tk = BATCHED_TESTS ? l_merge_turning(C, slot_F(fa), slot_L(MD, k - 1), k == 2 ? -1 : slot_F(k - 1), prv, cur, T.fwd)
                                                  : l_merge_turning_streamed(C, slot_F(fa), slot_L(MD, k - 1), k == 2 ? -1 : slot_F(k - 1), prv, cur, T.fwd);
                    if (tk) turn_bits |= 1u << k;
                }
            }
            {
                double total;
                const bool take = l_merge_weights(C, T.wE, w, false, total, fatal);
                T.sub_cand = take ? CandRef{-2, cur.logp, cur.ke, cur.idx} : CandRef{-3, T.E_logp, T.E_ke, T.E_idx};
                T.sub_log_size = total;
                if (fatal) T.stop = STOP_FATAL;
                else if (turn_bits & 2u) T.stop = STOP_TURNING;
            }
            for (int k = 2; k <= t && T.stop == STOP_NONE; ++k) {
                double a_log_size; CandRef ac;
                C.pend.get(k - 1, a_log_size, ac);
                double total;
                const bool take = l_merge_weights(C, a_log_size, T.sub_log_size, false, total, fatal);
                if (take) {
                    T.used &= ~(1u << ac.slot);
                } else {
                    if (T.sub_cand.slot >= 0) T.used &= ~(1u << T.sub_cand.slot);
                    T.sub_cand = ac;
                }
                T.sub_log_size = total;
                if (fatal) T.stop = STOP_FATAL;
                else if ((turn_bits >> k) & 1u) T.stop = STOP_TURNING;
            }
            if (T.stop == STOP_NONE) {
                const uint64_t ne = n - 1;
                if ((ne & 3) == 0 && T.depth > 1) {
                    const int fs = slot_F(ne == 0 ? (int)T.depth : (int)__builtin_ctzll(ne));
                    C.stS(prv.z, fs);
                    C.stS(prv.v, fs + 1);
                }
                if (n + 1 < T.nleaf) {
                    C.stS(cur.z, slot_L(MD, t)); C.stS(cur.v, slot_L(MD, t) + 1);
                    if (T.sub_cand.slot == -2) T.sub_cand.slot = l_cand_to_pool(C, T.used, cur.z);
                    else if (T.sub_cand.slot == -3) T.sub_cand.slot = l_cand_to_pool(C, T.used, prv.z);
                    C.pend.put(t, T.sub_log_size, T.sub_cand);
                    T.n = n + 1;
                    return true;                                  // the doubling goes on
                }
            }
        }
        over = true;
    }
    (void)over;
    // ---- the doubling is over
    if (T.stop == STOP_FATAL || fatal) { fatal = true; R.depth = T.depth; R.chosen = T.mc; return false; }
    if (T.stop == STOP_DIVERGING) { T.used = T.used_before; R.depth = T.depth; R.chosen = T.mc; return false; }
    if (T.stop == STOP_TURNING) {
        T.used = T.used_before;
        if (!T.in_extra) { T.in_extra = true; T.extra_left = s.extra_doublings; }
    } else {
        // top-level U-turn tests of the finished sub-tree (last leaf cur) against the main tree (src/nuts.rs:143-161)
        bool turning = false;
        if (T.check) {
            if (T.depth == 0) turning = l_turning<NP>(prv.z, prv.v, cur.z, cur.v, T.fwd);
            else
                turning = BATCHED_TESTS ? l_merge_turning(C, EDGE0_Z + 3 * (T.fwd ? T.left_slot : T.right_slot), EDGE0_Z + 3 * (T.fwd ? T.right_slot : T.left_slot),
                                                          T.depth == 1 ? -1 : slot_F((int)T.depth), prv, cur, T.fwd)
                                        : l_merge_turning_streamed(C, EDGE0_Z + 3 * (T.fwd ? T.left_slot : T.right_slot), EDGE0_Z + 3 * (T.fwd ? T.right_slot : T.left_slot),
                                                                   T.depth == 1 ? -1 : slot_F((int)T.depth), prv, cur, T.fwd);
        }
        double total;
        const bool take = l_merge_weights(C, T.log_size, T.sub_log_size, true, total, fatal);
        if (fatal) { R.depth = T.depth; R.chosen = T.mc; return false; }
        if (take) {
            if (T.mc.slot >= 0) T.used &= ~(1u << T.mc.slot);
            if (T.sub_cand.slot == -2) T.sub_cand.slot = l_cand_to_pool(C, T.used, cur.z);
            else if (T.sub_cand.slot == -3) T.sub_cand.slot = l_cand_to_pool(C, T.used, prv.z);
            T.mc = T.sub_cand;
        } else if (T.sub_cand.slot >= 0) {
            T.used &= ~(1u << T.sub_cand.slot);
        }
        const bool more = T.in_extra ? T.extra_left > 0 : (turning ? s.extra_doublings > 0 : T.depth + 1 < T.maxdepth);
        if (more) {
            int ns = T.fwd ? T.right_slot : T.left_slot;
            const int other_side = T.fwd ? T.left_slot : T.right_slot;
            if (ns == 0) ns = other_side == 1 ? 2 : 1;
            C.stS(cur.z, EDGE0_Z + 3 * ns); C.stS(cur.v, EDGE0_V + 3 * ns); C.stS(cur.g, EDGE0_G + 3 * ns);
            if (T.fwd) T.right_slot = ns; else T.left_slot = ns;
            T.o_is_edge = true; T.o_edge_sign = T.sign;
        }
        if (T.fwd) T.right_idx = cur.idx; else T.left_idx = cur.idx;
        T.depth += 1;
        T.log_size = total;
        if (turning && !T.in_extra) { T.in_extra = true; T.extra_left = s.extra_doublings; }
    }
    if (!l_doubling_begin(C, T, R, cur)) { R.depth = T.depth; R.chosen = T.mc; return false; }
    return true;
}

// the end of a draw: l_chain_draw behind its transition
template <bool TUNE, int NP, class LD>
NM_DEV void l_draw_end(LCtx<NP, LD>& C, uint64_t chain, uint64_t t_out, LAccept& col, const DrawResult& R, bool fatal) {
    constexpr int E = 2 * NP;
    const KParams& P = C.P;
    ChainScalars& sc = C.sc;
    nm_draw_stats out;
    out.draw = sc.draw_count; out.chain = P.chain_id_offset + chain;
    if (fatal) {
        sc.status = NM_CHAIN_LOGP_FATAL;
        if (P.out_stats) {
            nm_draw_stats zz = {};
            zz.draw = sc.draw_count; zz.chain = P.chain_id_offset + chain; zz.chain_status = NM_CHAIN_LOGP_FATAL;
            P.out_stats[t_out * P.n_chains + chain] = zz;
        }
        return;
    }
    double x[E], gx[E], z[E], gz[E];
    const size_t row = (size_t)(t_out * P.n_chains + chain) * P.dim;
    if (R.diverging && (P.out_div_start || P.out_div_start_grad || P.out_div_end))
        l_emit_divergence_vectors(C, R.div_start_idx, row);          // before P_X / P_GX take the new draw
    if (R.chosen.slot == -1 && !sc.px_stale) {
        C.ldW(x, P_X); C.ldW(gx, P_GX);
        C.ldW(z, P_Z); C.ldW(gz, P_GZ);
    } else {
        if (R.chosen.slot == -1) C.ldW(z, P_Z);
        else C.ldS(z, slot_C(C.md, R.chosen.slot));
#pragma unroll
        for (int d = 0; d < E; ++d) {
            const double tt = z[d] * C.sig[d];
            x[d] = __builtin_fma(1.0, C.mu[d], tt);
        }
        (void)C.dens.eval(x, gx, C.dim);
#pragma unroll
        for (int d = 0; d < E; ++d) gz[d] = gx[d] * C.sig[d];
        const bool need_x = sc.tuning || t_out + 1 == P.n_draws || P.out_div_start || P.out_div_start_grad;
        if (need_x) { C.stW(x, P_X); C.stW(gx, P_GX); }
        sc.px_stale = need_x ? 0 : 1;
        C.stW(z, P_Z); C.stW(gz, P_GZ);
        sc.logp = R.chosen.logp;
    }
    const int64_t idx = R.chosen.idx;
    l_write_row(C, P.out_positions, row, x);
    l_write_row(C, P.out_gradient, row, gx);
    l_write_row(C, P.out_tpos, row, z);
    l_write_row(C, P.out_tgrad, row, gz);
    double fd;
    {
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int d = 2 * l + k; acc = acc + (z[d] + gz[d]) * (z[d] + gz[d]); }
            p[l] = acc;
        }
        fd = pair_tree<NP>(p);
    }
    const double energy = R.chosen.ke - (R.chosen.logp + sc.logdet);
    const int64_t trans_id = sc.transform_id;
    sc.total_steps += col.count;
    const bool is_good = R.diverging ? ((idx < 0 ? -idx : idx) > 4) : (idx != 0);
    const uint64_t ast = l_adapt<TUNE>(C, col, is_good, x, gx);
    if (ast != NM_CHAIN_OK) sc.status = ast;
    out.depth = R.depth; out.maxdepth_reached = R.reached_maxdepth; out.diverging = R.diverging;
    out.tuning = sc.tuning; out.n_steps = sc.last_n_steps;
    out.index_in_trajectory = idx; out.transformation_index = trans_id;
    out.step_size = sc.step_size;
    out.step_size_bar = P.s.step_size_method == NM_STEP_FIXED ? P.s.fixed_step_size
                      : P.s.step_size_method == NM_STEP_ADAM ? lexp(sc.log_step) : lexp(sc.log_step_adapted);
    out.mean_tree_accept = sc.last_mean_tree_accept; out.mean_tree_accept_sym = sc.last_sym_mean_tree_accept;
    out.max_energy_error = sc.last_max_energy_error;
    out.logp = R.chosen.logp; out.energy = energy; out.energy_error = energy - R.e0;
    out.fisher_distance = fd;
    out.divergence_energy_error = (R.diverging && R.has_divergence_energy_error) ? R.divergence_energy_error : __builtin_nan("");
    out.chain_status = ast;
    out.transformation_update_id = -1;
    out.num_eigenvalues = 0;
    out.energy_change = __builtin_nan(""); out.average_step_size = __builtin_nan("");
    if (sc.mm_id != sc.stats_last_id) {
        out.transformation_update_id = sc.mm_id;
        l_write_row(C, P.out_mm_inv, row, C.sig);
        l_write_row(C, P.out_mm_mu, row, C.mu);
    }
    sc.stats_last_id = sc.mm_id;
    if (P.out_stats) P.out_stats[t_out * P.n_chains + chain] = out;
    sc.draw_count += 1;
}

// (round 5: the launch form with unsynchronised draws — l_run_rounds / nuts_lane_rounds_kernel, `lane_chains = 3` — was bit-exact and slower
// in every measurement of round 4 (K4 warm-up 328 against 157 ms) and is removed; the per-lane step functions above are unreferenced
// templates now: no code is generated for them)

template <int NP>
struct alignas(16) LaneShared {
#if NM_LANE_PROF
    unsigned long long lp_clock;
#endif
    uint32_t rng_ring[32 * 64];                           // [word & 31][lane]
    uint64_t pend[(LMAXDEPTH - 1) * 4 * 64];              // [level - 1][word][lane]: levels 1 .. maxdepth + extra_doublings - 2 are used
    double zig[(NP == 8 ? 1 : 2) * 257];                  // the ziggurat's x table and (where 4 blocks per CU still fit: 40 KB each) its f table
    double stage[2 * NP * 64];                            // [element][lane]: the momentum refresh's normals
};

// One block = one wavefront = 64 chains; blocks stride over the chains.  TUNE = true: the adaptation compiled in (launches that
// start inside the warm-up).
template <class Dens, int NP, bool TUNE>
__global__ __launch_bounds__(64, 1) void nuts_lane_draw_kernel(const KParams P, const LaneParams LP) {
    using LD0 = typename LaneDensity<Dens, NP>::type;
    using LD = typename std::conditional<kin_trait<Dens>::value, LKin<LD0>, LD0>::type;
    constexpr int E = 2 * NP;
    __shared__ LaneShared<NP> sh;
    for (int i = (int)threadIdx.x; i < 257; i += 64) {
        sh.zig[i] = P.zig_x[i];
        if (NP != 8) sh.zig[(NP != 8 ? 257 : 0) + i] = P.zig_f[i];
    }
    for (int i = (int)threadIdx.x; i < 2 * NP * 64; i += 64) sh.stage[i] = 0.0;
    dm_init_lds();
    __syncthreads();
    const int l = (int)threadIdx.x;
    for (uint64_t base = (uint64_t)blockIdx.x * 64; base < P.n_chains; base += (uint64_t)gridDim.x * 64) {
        const uint64_t chain = base + (uint64_t)l;
        if (chain < P.n_chains) {
            // (measured: leaving the scalars in their global record saves 74 registers and 350 B of scratch per lane, +10 % on deep trees,
            // but the short trees of the 8-schools model pay for the uncoalesced accesses at every draw's end: K4 3.78e9 -> 3.47e9)
            ChainScalars sc = P.sc[chain];
            LCtx<NP, LD> C(P, sc);
            C.dim = (int)P.dim;
            C.md = (int)P.layout_md;
            C.rw = make_rsrc(LP.lws + (size_t)blockIdx.x * NUM_PSLOT * E * 64, (uint64_t)NUM_PSLOT * E * 512);
            C.rsv = make_rsrc(LP.lsv + (size_t)blockIdx.x * LP.nslots * E * 64, (uint64_t)LP.nslots * E * 512);
            C.l8 = l * 8;
#if NM_LANE_PROF
            C.lp_clock = &sh.lp_clock;
            if (blockIdx.x == 0) sh.lp_clock = __builtin_amdgcn_s_memtime();
#endif
            C.pend.base = sh.pend + l;
            C.zig = {sh.zig, NP != 8 ? sh.zig + 257 : P.zig_f};
            C.stage = sh.stage + l;
            {   // the chain's persistent vectors: pvec[chain][slot][d] -> lane-major workspace
                const double* pv = P.pvec + (size_t)chain * NUM_PSLOT * P.dpad;
                for (int s_ = 0; s_ < (int)NUM_PSLOT; ++s_)
#pragma unroll
                    for (int e = 0; e < E; ++e) C.stWe(s_, e, pv[(size_t)s_ * P.dpad + e]);
            }
            C.ldW(C.sig, P_SIG); C.ldW(C.mu, P_MU);
            C.rng.init(sc.key, sc.rng_pos, sh.rng_ring + l);
            C.dens.init(P.logp_params, C.dim);
            if (sc.status == NM_CHAIN_OK) {
                for (uint64_t t = 0; t < P.n_draws; ++t) {
                    l_chain_draw<TUNE>(C, chain, t);
                    if (sc.status != NM_CHAIN_OK) break;
                }
            }
            sc.rng_pos = C.rng.pos;
            {
                double* pv = P.pvec + (size_t)chain * NUM_PSLOT * P.dpad;
                for (int s_ = 0; s_ < (int)NUM_PSLOT; ++s_)
#pragma unroll
                    for (int e = 0; e < E; ++e) pv[(size_t)s_ * P.dpad + e] = C.ldWe(s_, e);
            }
            P.sc[chain] = sc;
        }
    }
}


}  // namespace lane
}  // namespace nm
