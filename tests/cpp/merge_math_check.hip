// merge_math_check.hip — HOST program (compiled with hipcc, runs on the CPU): the straight-line merge arithmetic of
// csrc/dev_math.hpp (merge_math_impl: exp_sl, log1p_unit) against the general-purpose routines it replaces (dexp_branchy / dlog1p_branchy /
// dlog_branchy, the general-purpose forms of rounds 1-4, in merge_weights' original sequence, reference src/nuts.rs:172-207, src/math/util.rs:6-19), bit for bit, over special
// values and random operands.  Exit code 0 = identical everywhere.
#include "../../nuts_rs_amd/csrc/dev_math.hpp"
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
using namespace nm;
static double logaddexp_ref(double a, double b) {
    if (a == b) return a + dlog_branchy<false>(2.0);
    double diff = a - b;
    if (diff > 0.) return a + dlog1p_branchy<false>(dexp_branchy<false>(-diff));
    if (diff < 0.) return b + dlog1p_branchy<false>(dexp_branchy<false>(diff));
    return diff;
}
struct Ref { double total; uint32_t flags; };
static Ref merge_ref(double a, double b, bool is_main, uint64_t w) {
    Ref r; r.total = logaddexp_ref(a, b); r.flags = 0;
    const double self = is_main ? a : r.total;
    if (b >= self) { r.flags = 1; return r; }
    const double p = dexp_branchy<false>(b - self);
    if (!(p >= 0.0 && p < 1.0)) { r.flags = p == 1.0 ? 1u : 4u; return r; }
    const uint64_t p_int = (uint64_t)(p * 18446744073709551616.0);
    r.flags = 2u | (w < p_int ? 1u : 0u);
    return r;
}
static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t next() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static double unif() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
int main() {
    long bad = 0, n = 0;
    if (d2u(dlog_branchy<false>(2.0)) != d2u(0x1.62e42fefa39efp-1)) { printf("dlog(2) = %a, the constant in merge_math_impl is wrong\n", dlog_branchy<false>(2.0)); return 2; }
    const double inf = INFINITY, nan = NAN;
    std::vector<double> sp = {0.0, -0.0, 1.0, -1.0, 1e-300, -1e-300, 5e-324, 700.0, -700.0, 745.0, -745.2, -746.0, 709.7, 709.8, 710.0, 1e308, -1e308, inf, -inf, nan,
                              0.5, -0.5, 36.0, -36.0, 37.5, -37.5, 1e-17, -1e-17, 0.6931471805599453, -0.6931471805599453, 3.5, -3.5, 52.0, -53.0, 1075.0, -1075.0};
    // exp_sl / log1p_unit on their own
    for (double x : sp) { n++; if (d2u(exp_sl(x)) != d2u(dexp_branchy<false>(x)) && !(exp_sl(x) != exp_sl(x) && dexp_branchy<false>(x) != dexp_branchy<false>(x))) { bad++; printf("exp_sl(%a) = %a, dexp = %a\n", x, exp_sl(x), dexp_branchy<false>(x)); } }
    for (int i = 0; i < 2000000; ++i) {
        const double x = (unif() - 0.5) * (i & 1 ? 1500.0 : 80.0);
        n++; if (d2u(exp_sl(x)) != d2u(dexp_branchy<false>(x))) { if (bad++ < 10) printf("exp_sl(%a) = %a, dexp = %a\n", x, exp_sl(x), dexp_branchy<false>(x)); }
        const double e = i & 2 ? dexp_branchy<false>(-unif() * 60.0) : unif();
        n++; if (d2u(log1p_unit(e)) != d2u(dlog1p_branchy<false>(e))) { if (bad++ < 10) printf("log1p_unit(%a) = %a, dlog1p = %a\n", e, log1p_unit(e), dlog1p_branchy<false>(e)); }
    }
    for (double e : {0.0, 1.0, 5e-324, 1e-300, 0x1p-53, 0x1p-52, 0x1.fffffffffffffp-1, 0.41421356237309503, 0.4142135623730951, 0.41421356237309515}) {
        n++; if (d2u(log1p_unit(e)) != d2u(dlog1p_branchy<false>(e))) { bad++; printf("log1p_unit(%a) = %a, dlog1p = %a\n", e, log1p_unit(e), dlog1p_branchy<false>(e)); }
    }
    // log_sl / log1p_sl on every kind of double
    auto same = [](double a, double b) { return d2u(a) == d2u(b) || (a != a && b != b); };
    for (double x : sp) {
        n += 2;
        if (!same(log_sl(x), dlog_branchy<false>(x))) { bad++; printf("log_sl(%a) = %a, dlog = %a\n", x, log_sl(x), dlog_branchy<false>(x)); }
        if (!same(log1p_sl(x), dlog1p_branchy<false>(x))) { bad++; printf("log1p_sl(%a) = %a, dlog1p = %a\n", x, log1p_sl(x), dlog1p_branchy<false>(x)); }
    }
    for (double x : {-1.0, -1.0 + 0x1p-53, -1.0 - 0x1p-52, 0x1p-1074, 0x1p-1022, 0x1.fffffffffffffp-1023, 0x1p-1021, 2.0, 0x1.6a09e667f3bcdp+0, 0x1.6a09e667f3bccp+0, 0x1.6a09e667f3bcep+0, 0x1.fffffffffffffp+1023}) {
        n += 2;
        if (!same(log_sl(x), dlog_branchy<false>(x))) { bad++; printf("log_sl(%a) = %a, dlog = %a\n", x, log_sl(x), dlog_branchy<false>(x)); }
        if (!same(log1p_sl(x), dlog1p_branchy<false>(x))) { bad++; printf("log1p_sl(%a) = %a, dlog1p = %a\n", x, log1p_sl(x), dlog1p_branchy<false>(x)); }
    }
    for (int i = 0; i < 3000000; ++i) {
        const int kind = i % 5;
        double x = kind == 0 ? unif() * 4.0 : kind == 1 ? u2d(next()) : kind == 2 ? dexp_branchy<false>((unif() - 0.5) * 1400.0) : kind == 3 ? u2d(next() & 0x000fffffffffffffull) : (unif() - 0.5) * 4.0;
        n += 2;
        if (!same(log_sl(x), dlog_branchy<false>(x))) { if (bad++ < 10) printf("log_sl(%a) = %a, dlog = %a\n", x, log_sl(x), dlog_branchy<false>(x)); }
        if (!same(log1p_sl(x), dlog1p_branchy<false>(x))) { if (bad++ < 10) printf("log1p_sl(%a) = %a, dlog1p = %a\n", x, log1p_sl(x), dlog1p_branchy<false>(x)); }
    }
    // the merge: every pair of special values, both kinds of tree, three words; then random operands
    auto check = [&](double a, double b, bool is_main, uint64_t w) {
        const Ref r = merge_ref(a, b, is_main, w);
        const MergeOut o = merge_math_impl(a, b, is_main ? 1u : 0u, (uint32_t)w, (uint32_t)(w >> 32));
        const bool same_total = d2u(r.total) == d2u(o.total) || (r.total != r.total && o.total != o.total);
        n++;
        if (!same_total || r.flags != o.flags) { if (bad++ < 20) printf("merge(%a, %a, main %d, w %016llx): total %a flags %u, reference %a flags %u\n", a, b, (int)is_main, (unsigned long long)w, o.total, o.flags, r.total, r.flags); }
    };
    for (double a : sp) for (double b : sp) for (int m = 0; m < 2; ++m) for (uint64_t w : {0ull, ~0ull, 0x8000000000000000ull}) check(a, b, m != 0, w);
    for (int i = 0; i < 3000000; ++i) {
        const int kind = i % 6;
        double a = (unif() - 0.5) * 40.0, b = (unif() - 0.5) * 40.0;
        if (kind == 1) b = a + (unif() - 0.5) * 1e-3;
        if (kind == 2) { a *= 40.0; b *= 40.0; }
        if (kind == 3) b = a;
        if (kind == 4) b = a - unif() * 800.0;
        uint64_t w = next();
        if (kind == 5) {                     // a word right at the Bernoulli threshold
            const Ref r = merge_ref(a, b, (i & 8) != 0, 0);
            if (r.flags & 2u) { const double self = (i & 8) ? a : r.total; const uint64_t pi = (uint64_t)(dexp_branchy<false>(b - self) * 18446744073709551616.0); w = pi + (uint64_t)((i >> 4) % 3) - 1; }
        }
        check(a, b, (i & 8) != 0, w);
    }
    printf("%ld comparisons, %ld mismatches\n", n, bad);
    return bad ? 1 : 0;
}
