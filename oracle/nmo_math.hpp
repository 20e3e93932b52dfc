// nmo_math.hpp — scalar math + vector primitives of the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY.  This directory is the parity oracle: a scalar CPU restatement of the
// reference algorithm.  Nothing in the product (nuts_rs_amd/, libnuts_amd.so) may include, link or call
// it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// Follows (reference = pymc-devs/nuts-rs 0.18.3, paths relative to /root/reference):
//   src/math/util.rs:6-19      logaddexp
//   src/math/util.rs:21-505    per-element formulas of multiply / scalar_prods3 / vector_dot / axpy / axpy_out
//   src/math/cpu_math.rs:235-330, :553-559, :605-738   sq_norm_sum, sum_ln, variance / sigma updates
//
// Two knobs that the reference leaves platform-defined are explicit here (struct MathCfg):
//   * reduce order.  The reference sums with 4 SIMD accumulators whose width pulp picks at run time
//     (util.rs:357-395), so its own results differ between machines by a few ulp.  REDUCE_REF_SIMD
//     restates that structure for a given lane count; REDUCE_GPU is the fixed order the HIP engine uses
//     (per-lane serial partials over a [pair][thread] tiling, then an xor butterfly over the 64 lanes).
//   * transcendental functions.  Rust's f64::exp/ln/ln_1p/powf call the platform libm.  detmath=0 uses
//     this box's libm (what the reference would do here); detmath=1 uses the restated fdlibm-style
//     algorithms below, which the HIP engine implements operation-for-operation so that GPU and oracle
//     agree bit-for-bit (tests/ bound detmath-vs-libm by ulps).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace nmo {

enum { REDUCE_REF_SIMD = 0, REDUCE_GPU = 1 };

struct MathCfg {
    int64_t detmath = 0;
    int64_t reduce_mode = REDUCE_REF_SIMD;
    int64_t simd_lanes = 4;     // REDUCE_REF_SIMD: f64 lanes of the SIMD register (1, 2, 4, 8)
    int64_t gpu_threads = 64;   // REDUCE_GPU: threads cooperating on one chain (64 * waves)
    int64_t gpu_slice = 0;      // REDUCE_GPU: > 0: a chain wider than one block — consecutive slices of this many elements are reduced
                                // as above, each by its own block, and the slice totals added in slice order (engine: dim > 4096)
    int64_t lr_seq_dots = 0;    // the low-rank transformation's U'v as sequential fma dot products (the engine's matrix-core
                                // kernel for shared matrices: an MFMA accumulates over the inner index in ascending order);
                                // 2: the same, and from the moment the shared transformation is set (nmo_chain_set_transform) every
                                // reduction over dim uses the lockstep kernel's order (tile_order below)
    int64_t tile_order = 0;     // REDUCE_GPU: 1 = the order of the 16-chains-per-block lockstep kernel (nuts_lockstep.hpp): element d
                                // sits in stripe d / 16 on lane-row g = d % 4, slot r = (d % 16) / 4; a lane accumulates its slots in
                                // r order, a stripe's total is (p_0 + p_1) + (p_2 + p_3) over its four lane-rows, stripes are added in
                                // stripe order
};

static inline uint64_t f2u(double x) { uint64_t u; std::memcpy(&u, &x, 8); return u; }
static inline double u2f(uint64_t u) { double x; std::memcpy(&x, &u, 8); return x; }

// ---------------------------------------------------------------------------------------------
// Deterministic transcendental functions: table-driven exp / ln / ln_1p (Tang-style reductions, tables from
// tools/gen_detmath_tables.py), written so that every step is ONE IEEE-754 binary64 operation (+, -, *, fma,
// round-to-nearest-even integer, scaling by a power of two) and no division: the HIP engine executes the same
// sequence, so GPU and oracle agree bit for bit, and the dependent chain is ~12 operations long (these functions
// sit on the latency-critical scalar path of every tree merge).  Accuracy: < 1 ulp (tests bound it against libm).
//   exp:  x = n (ln2/64) + r, n = 64 k + j:  exp x = 2^k T[j] (1 + p(r)),  p = expm1 by a degree-6 polynomial
//   ln :  x = 2^k m, m in [sqrt 1/2, sqrt 2), j = rint(64 m), z = m R[j] - 1 (one fma):
//         ln x = k ln2 + (-ln R[j]) + log1p(z),  log1p by a degree-10 polynomial, hi/lo sums kept apart
// ---------------------------------------------------------------------------------------------
#include "nmo_detmath_tables.hpp"
static const double DM_T_HI[64] = DM_EXP_T_HI, DM_T_LO[64] = DM_EXP_T_LO;
static const double DM_R[47] = DM_LOG_R, DM_F_HI[47] = DM_LOG_F_HI, DM_F_LO[47] = DM_LOG_F_LO;

static inline double det_exp(double x) {
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    const double nf = std::nearbyint(x * DM_EXP_INV_L);
    const double r1 = std::fma(-nf, DM_EXP_L_HI, x);
    const double r = std::fma(-nf, DM_EXP_L_LO, r1);
    const int n = (int)nf;
    const int j = n & 63, k = n >> 6;
    const double r2 = r * r;
    const double a = std::fma(r, DM_EXP_E3, DM_EXP_E2);
    const double b = std::fma(r, DM_EXP_E5, DM_EXP_E4);
    const double q = std::fma(r2, std::fma(r2, DM_EXP_E6, b), a);
    const double p = std::fma(r2, q, r);                   // expm1(r)
    const double sum = std::fma(DM_T_HI[j], p, DM_T_LO[j]);
    return std::ldexp(DM_T_HI[j] + sum, k);
}

// ln(x) + c / x for finite x > 0 (c = 0: plain ln; ln_1p passes the rounding error of 1 + t)
static inline double det_log_core(double x, double c) {
    uint64_t u = f2u(x);
    int k = 0;
    if (u < 0x0010000000000000ull) {                       // subnormal
        x *= 1.80143985094819840000e+16;                   // 2^54
        u = f2u(x);
        k = -54;
    }
    const uint64_t mant = u & 0x000fffffffffffffull;
    const int up = mant >= 0x6a09e667f3bcdull;             // mantissa >= sqrt(2): halve it
    k += (int)(u >> 52) - 1023 + up;
    const double m = u2f(mant | ((uint64_t)(1023 - up) << 52));   // [sqrt 1/2, sqrt 2)
    const int j = (int)std::nearbyint(m * 64.0) - DM_LOG_J0;
    const double rj = DM_R[j];
    const double z = std::fma(m, rj, -1.0);
    const double dk = (double)k;
    const double z2 = z * z, z4 = z2 * z2;
    const double p01 = std::fma(z, DM_LOG_C3, DM_LOG_C2), p23 = std::fma(z, DM_LOG_C5, DM_LOG_C4);
    const double p45 = std::fma(z, DM_LOG_C7, DM_LOG_C6), p67 = std::fma(z, DM_LOG_C9, DM_LOG_C8);
    const double q0 = std::fma(z2, p23, p01), q1 = std::fma(z2, p67, p45);
    const double Q = std::fma(z4, std::fma(z4, DM_LOG_C10, q1), q0);
    // c / x ~ c * R[j] * 2^-k (c is a rounding error: a 1 % reciprocal is plenty)
    const int kc = k < -1000 ? -1000 : (k > 1000 ? 1000 : k);       // c != 0 only from ln_1p, where |k| <= 1024
    const double corr = (c * rj) * u2f((uint64_t)(1023 - kc) << 52);
    const double lo = std::fma(dk, DM_LN2_LO, DM_F_LO[j]) + corr;
    const double t = std::fma(z2, Q, lo);
    const double hk = dk * DM_LN2_HI;                      // exact
    const double s1 = hk + DM_F_HI[j], e1 = (hk - s1) + DM_F_HI[j];
    const double s2 = s1 + z, e2 = (s1 - s2) + z;
    return s2 + ((e1 + e2) + t);
}

static inline double det_log(double x) {
    if (x != x) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (std::isinf(x)) return x;
    return det_log_core(x, 0.0);
}

// ln(1 + x): ln of the rounded sum plus the first-order term of its rounding error
static inline double det_log1p(double x) {
    const double u = 1.0 + x;
    if (u == 1.0) return x;
    if (!(u == u) || std::isinf(u) || !(u > 0.0)) return det_log(u);
    return det_log_core(u, x - (u - 1.0));
}

// exp(x) - 1 (reference: f64::exp_m1 in the isokinetic momentum refresh, src/dynamics/transformed_hamiltonian.rs:800-801).
// |x| <= 0.35: the Taylor series to x^14 in Horner form (one fma per term); beyond: det_exp(x) - 1.  ~1-2 ulp.
static inline double det_expm1(double x) {
    if (!(std::fabs(x) <= 0.35)) return det_exp(x) - 1.0;
    double p = 1.1470745597729725e-11;                                 // 1/14!
    p = std::fma(x, p, 1.6059043836821613e-10);    // 1/13!
    p = std::fma(x, p, 2.08767569878681e-09);    // 1/12!
    p = std::fma(x, p, 2.505210838544172e-08);    // 1/11!
    p = std::fma(x, p, 2.755731922398589e-07);    // 1/10!
    p = std::fma(x, p, 2.7557319223985893e-06);    // 1/9!
    p = std::fma(x, p, 2.48015873015873e-05);    // 1/8!
    p = std::fma(x, p, 0.0001984126984126984);    // 1/7!
    p = std::fma(x, p, 0.001388888888888889);    // 1/6!
    p = std::fma(x, p, 0.008333333333333333);    // 1/5!
    p = std::fma(x, p, 0.041666666666666664);    // 1/4!
    p = std::fma(x, p, 0.16666666666666666);    // 1/3!
    p = std::fma(x, p, 0.5);
    return std::fma(x * x, p, x);
}

// sin and cos of x (reference: f64::sin / f64::cos of the step size, src/math/util.rs:580-581).  Cody-Waite reduction by
// pi/2 in two fma steps, then the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4]; every step is one
// binary64 operation, the engine runs the same sequence.  Odd / even symmetry is exact, like libm's.  Accuracy ~1 ulp for
// |x| up to a few thousand (step sizes), deterministic everywhere.
static inline void det_sincos(double x, double* sn, double* cs) {
    if (!(x == x) || std::isinf(x)) { *sn = NAN; *cs = NAN; return; }
    const double ax = std::fabs(x);
    const double nf = std::nearbyint(ax * 6.36619772367581382433e-01);
    double r = std::fma(-nf, 1.57079632679489655800e+00, ax);
    r = std::fma(-nf, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = std::fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = std::fma(z, ps, 2.75573137070700676789e-06);
    ps = std::fma(z, ps, -1.98412698298579493134e-04);
    ps = std::fma(z, ps, 8.33333333332248946124e-03);
    ps = std::fma(z, ps, -1.66666666666666324348e-01);
    const double sr = std::fma(r * z, ps, r);
    double pc = std::fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = std::fma(z, pc, -2.75573143513906633035e-07);
    pc = std::fma(z, pc, 2.48015872894767294178e-05);
    pc = std::fma(z, pc, -1.38888888888741095749e-03);
    pc = std::fma(z, pc, 4.16666666666666019037e-02);
    const double cr = std::fma(z * z, pc, std::fma(-0.5, z, 1.0));
    const int q = (int)((long long)nf & 3);
    double s_, c_;
    if (q == 0) { s_ = sr; c_ = cr; }
    else if (q == 1) { s_ = cr; c_ = -sr; }
    else if (q == 2) { s_ = -sr; c_ = -cr; }
    else { s_ = -cr; c_ = sr; }
    *sn = x < 0.0 ? -s_ : s_;
    *cs = c_;
}

struct Ctx {
    MathCfg cfg;
    double exp(double x) const { return cfg.detmath ? det_exp(x) : std::exp(x); }
    double ln(double x) const { return cfg.detmath ? det_log(x) : std::log(x); }
    double ln_1p(double x) const { return cfg.detmath ? det_log1p(x) : std::log1p(x); }
    // count.powf(-k) of dual averaging (reference src/stepsize/dual_avg.rs:60)
    double powf(double a, double b) const { return cfg.detmath ? det_exp(b * det_log(a)) : std::pow(a, b); }

    double exp_m1(double x) const { return cfg.detmath ? det_expm1(x) : std::expm1(x); }
    double sin(double x) const { if (!cfg.detmath) return std::sin(x); double s_, c_; det_sincos(x, &s_, &c_); return s_; }
    double cos(double x) const { if (!cfg.detmath) return std::cos(x); double s_, c_; det_sincos(x, &s_, &c_); return c_; }

    // reference src/math/util.rs:6-19
    double logaddexp(double a, double b) const {
        if (a == b) return a + ln(2.0);
        double diff = a - b;
        if (diff > 0.) return a + ln_1p(exp(-diff));
        if (diff < 0.) return b + ln_1p(exp(diff));
        return diff;  // NaN
    }

    // ----- ordered reductions ---------------------------------------------------------------
    // GPU order: element d lives in pair q=d/2 (component j=d%2) of thread t=q%T at step m=q/T;
    // a thread accumulates its elements in (m, j) order; wave totals come from an xor butterfly with
    // offsets 1,2,4,8,16,32; the W=T/64 wave totals are added in wave order.
    template <class Acc>
    double tile_reduce(size_t n, Acc&& step) const {
        double total = 0.0;
        for (size_t s0 = 0, b = 0; s0 < n; s0 += 16, ++b) {
            double p[4] = {0.0, 0.0, 0.0, 0.0};
            for (size_t d = s0; d < std::min(n, s0 + 16); ++d) p[d % 4] = step(p[d % 4], d);      // (ascending d = ascending r per lane-row)
            const double t = (p[0] + p[1]) + (p[2] + p[3]);
            total = b == 0 ? t : total + t;
        }
        return total;
    }
    template <class Acc>  // Acc(double acc, size_t d) -> double : one accumulation step
    double gpu_reduce(size_t n, Acc&& step) const {
        if (cfg.tile_order) return tile_reduce(n, step);
        const size_t S = (size_t)cfg.gpu_slice;
        if (S > 0 && n > S) {
            double total = 0.0;
            for (size_t off = 0, b = 0; off < n; off += S, ++b) {
                const double t = gpu_reduce_block(off, std::min(S, n - off), step);
                total = b == 0 ? t : total + t;
            }
            return total;
        }
        return gpu_reduce_block(0, n, step);
    }
    template <class Acc>
    double gpu_reduce_block(size_t off, size_t n, Acc&& step) const {
        const size_t T = (size_t)cfg.gpu_threads;
        std::vector<double> part(T, 0.0);
        for (size_t dl = 0; dl < n; ++dl) {
            const size_t d = off + dl;
            size_t q = dl / 2, t = q % T;
            part[t] = step(part[t], d);
        }
        // (a thread's elements are visited in increasing d, which is (m, j) order)
        double total = 0.0;
        for (size_t w = 0; w < T / 64; ++w) {
            double lane[64], nxt[64];
            for (int l = 0; l < 64; ++l) lane[l] = part[w * 64 + l];
            for (int s = 1; s < 64; s <<= 1) {
                for (int l = 0; l < 64; ++l) nxt[l] = lane[l] + lane[l ^ s];
                for (int l = 0; l < 64; ++l) lane[l] = nxt[l];
            }
            total = (w == 0) ? lane[0] : total + lane[0];
        }
        return total;
    }

    // sum of plain terms (user densities: sequential scalar sum, reference test_logps.rs:51-56)
    double sum_terms(const double* t, size_t n) const {
        if (cfg.reduce_mode == REDUCE_GPU)
            return gpu_reduce(n, [&](double acc, size_t d) { return acc + t[d]; });
        double s = 0.0;
        for (size_t i = 0; i < n; ++i) s += t[i];
        return s;
    }

    // reference src/math/util.rs:349-400 (vector_dot)
    double vector_dot(const double* a, const double* b, size_t n) const {
        if (cfg.reduce_mode == REDUCE_GPU)
            return gpu_reduce(n, [&](double acc, size_t d) { return std::fma(a[d], b[d], acc); });
        const size_t L = (size_t)cfg.simd_lanes;
        const size_t nvec = n / L, ngroups = nvec / 4;
        std::vector<double> acc(4 * L, 0.0);
        for (size_t gi = 0; gi < ngroups; ++gi)
            for (size_t r = 0; r < 4; ++r)
                for (size_t l = 0; l < L; ++l) {
                    size_t i = (gi * 4 + r) * L + l;
                    acc[r * L + l] = std::fma(a[i], b[i], acc[r * L + l]);
                }
        for (size_t vi = ngroups * 4; vi < nvec; ++vi)
            for (size_t l = 0; l < L; ++l) {
                size_t i = vi * L + l;
                acc[l] = std::fma(a[i], b[i], acc[l]);
            }
        double result = simd_combine(acc.data(), L);
        for (size_t i = nvec * L; i < n; ++i) result += a[i] * b[i];
        return result;
    }

    // U_k . v of apply_lowrank_transform (cpu_math.rs:350-357; faer's own order is not reproducible)
    double lowrank_dot(const double* a, const double* b, size_t n) const {
        if (!cfg.lr_seq_dots) return vector_dot(a, b, n);
        double acc = 0.0;
        for (size_t i = 0; i < n; ++i) acc = std::fma(a[i], b[i], acc);
        return acc;
    }

    // reference src/math/util.rs:221-347 (scalar_prods3): s=(p1+p2)-n1 ; (sum s*x, sum s*y)
    void scalar_prods3(const double* p1, const double* n1, const double* p2, const double* x,
                       const double* y, size_t n, double* o1, double* o2) const {
        if (cfg.reduce_mode == REDUCE_GPU) {
            *o1 = gpu_reduce(n, [&](double acc, size_t d) { return std::fma((p1[d] + p2[d]) - n1[d], x[d], acc); });
            *o2 = gpu_reduce(n, [&](double acc, size_t d) { return std::fma((p1[d] + p2[d]) - n1[d], y[d], acc); });
            return;
        }
        const size_t L = (size_t)cfg.simd_lanes;
        const size_t nvec = n / L, ngroups = nvec / 4;
        std::vector<double> a1(4 * L, 0.0), a2(4 * L, 0.0);
        for (size_t gi = 0; gi < ngroups; ++gi)
            for (size_t r = 0; r < 4; ++r)
                for (size_t l = 0; l < L; ++l) {
                    size_t i = (gi * 4 + r) * L + l;
                    double s = (p1[i] + p2[i]) - n1[i];
                    a1[r * L + l] = std::fma(s, x[i], a1[r * L + l]);
                    a2[r * L + l] = std::fma(s, y[i], a2[r * L + l]);
                }
        for (size_t vi = ngroups * 4; vi < nvec; ++vi)
            for (size_t l = 0; l < L; ++l) {
                size_t i = vi * L + l;
                double s = (p1[i] + p2[i]) - n1[i];
                a1[l] = std::fma(s, x[i], a1[l]);
                a2[l] = std::fma(s, y[i], a2[l]);
            }
        double r1 = simd_combine(a1.data(), L), r2 = simd_combine(a2.data(), L);
        for (size_t i = nvec * L; i < n; ++i) {
            double s = p1[i] - n1[i] + p2[i];
            r1 += s * x[i];
            r2 += s * y[i];
        }
        *o1 = r1;
        *o2 = r2;
    }

    // reference src/math/cpu_math.rs:300-304 (array_sum_ln, sequential)
    double sum_ln(const double* a, size_t n) const {
        if (cfg.reduce_mode == REDUCE_GPU)
            return gpu_reduce(n, [&](double acc, size_t d) { return acc + ln(a[d]); });
        double s = 0.0;
        for (size_t i = 0; i < n; ++i) s += ln(a[i]);
        return s;
    }

    // reference src/math/cpu_math.rs:235-243 (sq_norm_sum)
    double sq_norm_sum(const double* x, const double* y, size_t n) const {
        if (cfg.reduce_mode == REDUCE_GPU)
            return gpu_reduce(n, [&](double acc, size_t d) { return acc + (x[d] + y[d]) * (x[d] + y[d]); });
        double s = 0.0;
        for (size_t i = 0; i < n; ++i) s += (x[i] + y[i]) * (x[i] + y[i]);
        return s;
    }

    // `v.iter().map(|x| x * x).sum::<f64>()` (reference src/math/cpu_math.rs:498, :517, :541): a sequential scalar sum
    double sum_sq(const double* a, size_t n) const {
        if (cfg.reduce_mode == REDUCE_GPU)
            return gpu_reduce(n, [&](double acc, size_t d) { return acc + a[d] * a[d]; });
        double s = 0.0;
        for (size_t i = 0; i < n; ++i) s += a[i] * a[i];
        return s;
    }
    // `p.zip(g).map(|(p, g)| p * g * inv).sum()` (reference src/math/cpu_math.rs:522-526)
    double sum_prod_scaled(const double* p, const double* g, double inv, size_t n) const {
        if (cfg.reduce_mode == REDUCE_GPU)
            return gpu_reduce(n, [&](double acc, size_t d) { return acc + p[d] * g[d] * inv; });
        double s = 0.0;
        for (size_t i = 0; i < n; ++i) s += p[i] * g[i] * inv;
        return s;
    }
    // elements below this index go through the SIMD registers (fused multiply-adds); the remainder is the scalar tail loop of
    // the reference's kernels, which is written WITHOUT mul_add (src/math/util.rs:561-565, :644-646, :713-715).  The engine's
    // arithmetic (REDUCE_GPU) has no tail: every element is fused.
    size_t simd_head(size_t n) const {
        if (cfg.reduce_mode == REDUCE_GPU) return n;
        return n - n % (size_t)cfg.simd_lanes;
    }

    // std_norm_flow (reference src/math/util.rs:507-589): pos_out = p cos e + v sin e ; vel = -p sin e + v cos e
    void std_norm_flow(const double* pos, double* pos_out, double* vel, double epsilon, size_t n) const {
        const double es = sin(epsilon), ec = cos(epsilon);
        const size_t h = simd_head(n);
        for (size_t i = 0; i < h; ++i) {
            const double p = pos[i], v = vel[i];
            pos_out[i] = std::fma(p, ec, v * es);
            vel[i] = std::fma(p, -es, v * ec);
        }
        for (size_t i = h; i < n; ++i) {
            const double p = pos[i], v = vel[i];
            pos_out[i] = p * ec + v * es;
            vel[i] = p * (-es) + v * ec;
        }
    }
    // std_norm_grad_flow / _inplace (reference src/math/util.rs:591-741): vel_out = vel + e (pos + grad)
    void std_norm_grad_flow(const double* pos, const double* grad, const double* vel, double* vel_out, double epsilon, size_t n) const {
        const size_t h = simd_head(n);
        for (size_t i = 0; i < h; ++i) vel_out[i] = std::fma(epsilon, pos[i] + grad[i], vel[i]);
        for (size_t i = h; i < n; ++i) vel_out[i] = vel[i] + epsilon * (pos[i] + grad[i]);
    }
    // array_normalize (reference src/math/cpu_math.rs:496-503)
    void array_normalize(double* v, size_t n) const {
        const double inv = 1.0 / std::sqrt(sum_sq(v, n));
        for (size_t i = 0; i < n; ++i) v[i] *= inv;
    }
    // esh_momentum_update (reference src/math/cpu_math.rs:505-551): the ESH momentum step on the unit sphere; returns the
    // kinetic-energy change
    double esh_momentum_update(const double* gradient, double* momentum, double step_size, size_t n) const {
        const double grad_norm = std::sqrt(sum_sq(gradient, n));
        const double inv_grad_norm = 1.0 / grad_norm;
        const double momentum_proj = sum_prod_scaled(momentum, gradient, inv_grad_norm, n);
        const double dims_m1 = (double)(n - 1);
        const double delta = step_size * grad_norm / dims_m1;
        const double zeta = exp(-delta);
        const double coeff_g = (1.0 - zeta) * (1.0 + zeta + momentum_proj * (1.0 - zeta));
        const double coeff_p = 2.0 * zeta;
        for (size_t i = 0; i < n; ++i) momentum[i] = coeff_g * (gradient[i] * inv_grad_norm) + coeff_p * momentum[i];
        const double inv = 1.0 / std::sqrt(sum_sq(momentum, n));
        for (size_t i = 0; i < n; ++i) momentum[i] *= inv;
        const double arg = momentum_proj + (1.0 - momentum_proj) * zeta * zeta;
        return (delta - 6.93147180559945286227e-01 + ln_1p(arg)) * dims_m1;
    }

private:
    // (acc0+acc1)+(acc2+acc3) lane-wise, then a halving horizontal sum of the register.
    static double simd_combine(const double* acc, size_t L) {
        std::vector<double> v(L);
        for (size_t l = 0; l < L; ++l) v[l] = (acc[l] + acc[L + l]) + (acc[2 * L + l] + acc[3 * L + l]);
        for (size_t h = L / 2; h >= 1; h /= 2) {
            for (size_t l = 0; l < h; ++l) v[l] = v[l] + v[l + h];
            if (h == 1) break;
        }
        return v[0];
    }
};

// ----- element-wise primitives (reference src/math/util.rs) -----------------------------------
// axpy_out: out = fma(a, x, y)            util.rs:448-505
static inline void axpy_out(const double* x, const double* y, double a, double* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = std::fma(a, x[i], y[i]);
}
// axpy: y = fma(a, x, y)                  util.rs:402-446
static inline void axpy(const double* x, double* y, double a, size_t n) {
    for (size_t i = 0; i < n; ++i) y[i] = std::fma(a, x[i], y[i]);
}
// multiply: out = x*y                     util.rs:21-112
static inline void multiply(const double* x, const double* y, double* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = x[i] * y[i];
}
static inline bool all_finite(const double* a, size_t n) {
    for (size_t i = 0; i < n; ++i) if (!std::isfinite(a[i])) return false;
    return true;
}
static inline bool all_finite_and_nonzero(const double* a, size_t n) {
    for (size_t i = 0; i < n; ++i) if (!std::isfinite(a[i]) || a[i] == 0.0) return false;
    return true;
}
// f64::clamp (NaN stays NaN)
static inline double clampd(double v, double lo, double hi) {
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

}  // namespace nmo
