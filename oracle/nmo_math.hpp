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
};

static inline uint64_t f2u(double x) { uint64_t u; std::memcpy(&u, &x, 8); return u; }
static inline double u2f(uint64_t u) { double x; std::memcpy(&x, &u, 8); return x; }

// ---------------------------------------------------------------------------------------------
// Deterministic transcendental functions (algorithms: Sun fdlibm e_exp.c / e_log.c, restated).
// Every operation is a single IEEE-754 binary64 operation; no contraction (built with
// -ffp-contract=off), so the same sequence on the GPU gives the same bits.
// ---------------------------------------------------------------------------------------------
static inline double det_exp(double x) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return INFINITY;
    if (x < -7.45133219101941108420e+02) return 0.0;
    double ax = std::fabs(x);
    double hi = x, lo = 0.0;
    int k = 0;
    if (ax > 0.34657359027997264) {                    // |x| > 0.5 ln2
        k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
        double t = (double)k;
        hi = x - t * ln2HI;
        lo = t * ln2LO;
        x = hi - lo;
    } else if (ax < 3.725290298461914e-09) {           // |x| < 2^-28
        return 1.0 + x;
    }
    double t = x * x;
    double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    // scale by 2^k
    if (k >= -1021) {
        if (k == 1024) return y * 2.0 * u2f((uint64_t)(1023 + 1023) << 52);
        return y * u2f((uint64_t)(1023 + k) << 52);
    }
    return y * u2f((uint64_t)(1023 + k + 1000) << 52) * 9.33263618503218878990e-302;  // 2^-1000
}

static inline double det_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = f2u(x);
    int32_t hx = (int32_t)(u >> 32);
    uint32_t lx = (uint32_t)u;
    int k = 0;
    if (hx < 0x00100000) {                             // x < 2^-1022, zero or negative
        if (((hx & 0x7fffffff) | lx) == 0) return -INFINITY;
        if (hx < 0) return NAN;
        k -= 54;
        x *= 1.80143985094819840000e+16;               // 2^54
        u = f2u(x);
        hx = (int32_t)(u >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    u = (u & 0xffffffffull) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);
    x = u2f(u);
    k += (i >> 20);
    double f = x - 1.0;
    double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {                 // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            return dk * ln2_hi + dk * ln2_lo;
        }
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f);
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    int32_t j = 0x6b851 - hx;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// ln(1+x) for x > -1 by the (1+x)-1 correction; ~1-2 ulp, enough for logaddexp (|x| <= 1 there).
static inline double det_log1p(double x) {
    double u = 1.0 + x;
    if (u == 1.0) return x;
    if (!(u == u) || std::isinf(u)) return det_log(u);
    return det_log(u) * (x / (u - 1.0));
}

struct Ctx {
    MathCfg cfg;
    double exp(double x) const { return cfg.detmath ? det_exp(x) : std::exp(x); }
    double ln(double x) const { return cfg.detmath ? det_log(x) : std::log(x); }
    double ln_1p(double x) const { return cfg.detmath ? det_log1p(x) : std::log1p(x); }
    // count.powf(-k) of dual averaging (reference src/stepsize/dual_avg.rs:60)
    double powf(double a, double b) const { return cfg.detmath ? det_exp(b * det_log(a)) : std::pow(a, b); }

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
    template <class Acc>  // Acc(double acc, size_t d) -> double : one accumulation step
    double gpu_reduce(size_t n, Acc&& step) const {
        const size_t T = (size_t)cfg.gpu_threads;
        std::vector<double> part(T, 0.0);
        for (size_t d = 0; d < n; ++d) {
            size_t q = d / 2, t = q % T;
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
