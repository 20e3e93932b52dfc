// lowrank_host.cpp — the host side of the low-rank mass-matrix adaptation: `LowRankMassMatrixStrategy::compute_update`
// (reference src/transform/adapt/low_rank.rs:73-142 with rescale_points :161-226, estimate_mass_matrix :228-260 and
// spd_mean :262-290), restated in plain C++ with its own dense linear algebra (the reference uses faer: thin SVD,
// column-pivoted QR, self-adjoint eigendecomposition).
//
// What the reference needs from the two thin SVDs is only an orthonormal basis U of each window's column space (the bases
// are concatenated and re-orthonormalised, and everything downstream — the projected covariances, their SPD mean, the
// filtered eigenpairs mapped back — depends on the SPAN alone), so both SVDs and the pivoted QR of the reference are
// served by one routine here: a RANK-REVEALING Householder QR with column pivoting.  Where a window is rank deficient
// (fewer draws than dims; centred rows always lose one), faer's factorisations complete the basis with arbitrary
// orthonormal vectors; in exact arithmetic those directions carry eigenvalue exactly 1 and are filtered out, but in
// floating point they sit at the noise floor eps * |G^1/2 D G^1/2| ~ eps / gamma^2, which reaches O(1) for dim in the
// hundreds: the reference's result there is rounding noise.  Dropping the numerically null directions computes the
// exact-arithmetic value of the reference's algorithm (tests compare with the LAPACK restatement in oracle/lowrank.py,
// literal and rank-revealing).  Symmetric eigenproblems: Householder tridiagonalisation + implicit QL.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

typedef std::vector<double> V;
// column-major dense matrix, like faer's Mat
struct Mat {
    size_t r = 0, c = 0;
    V a;
    Mat() {}
    Mat(size_t r_, size_t c_) : r(r_), c(c_), a(r_ * c_, 0.0) {}
    double& operator()(size_t i, size_t j) { return a[j * r + i]; }
    double operator()(size_t i, size_t j) const { return a[j * r + i]; }
    double* col(size_t j) { return &a[j * r]; }
    const double* col(size_t j) const { return &a[j * r]; }
};

bool all_finite(const V& v) { for (double x : v) if (!std::isfinite(x)) return false; return true; }

Mat matmul(const Mat& A, const Mat& B) {                 // A (m x k) * B (k x n)
    Mat C(A.r, B.c);
    for (size_t j = 0; j < B.c; ++j)
        for (size_t l = 0; l < A.c; ++l) {
            const double b = B(l, j);
            const double* a = A.col(l);
            double* c = C.col(j);
            for (size_t i = 0; i < A.r; ++i) c[i] += a[i] * b;
        }
    return C;
}
Mat matmul_tn(const Mat& A, const Mat& B) {              // A' (k x m)' * B (k x n) -> m x n
    Mat C(A.c, B.c);
    for (size_t j = 0; j < B.c; ++j)
        for (size_t i = 0; i < A.c; ++i) {
            const double *a = A.col(i), *b = B.col(j);
            double s = 0.0;
            for (size_t l = 0; l < A.r; ++l) s += a[l] * b[l];
            C(i, j) = s;
        }
    return C;
}
Mat matmul_nt(const Mat& A, const Mat& B) {              // A (m x k) * B' (n x k)' -> m x n
    Mat C(A.r, B.r);
    for (size_t l = 0; l < A.c; ++l)
        for (size_t j = 0; j < B.r; ++j) {
            const double b = B(j, l);
            const double* a = A.col(l);
            double* c = C.col(j);
            for (size_t i = 0; i < A.r; ++i) c[i] += a[i] * b;
        }
    return C;
}

// Orthonormal basis of the column space of A (m x n) by Householder QR with column pivoting: as many columns as the
// numerical rank (pivot norm > RANK_TOL x the largest column norm).
constexpr double RANK_TOL = 1e-10;
Mat thin_q(Mat A) {
    const size_t m = A.r, n = A.c, k = std::min(m, n);
    V norms(n);
    std::vector<V> vs(k);
    V betas(k, 0.0);
    for (size_t j = 0; j < n; ++j) { double s = 0; for (size_t i = 0; i < m; ++i) s += A(i, j) * A(i, j); norms[j] = s; }
    double top = 0.0;
    for (size_t j = 0; j < n; ++j) top = std::max(top, norms[j]);
    size_t rank = k;
    for (size_t s = 0; s < k; ++s) {
        size_t piv = s;                                    // pivot: the remaining column of largest remaining norm
        for (size_t j = s + 1; j < n; ++j) if (norms[j] > norms[piv]) piv = j;
        if (!(norms[piv] > RANK_TOL * RANK_TOL * top)) { rank = s; break; }
        if (piv != s) {
            for (size_t i = 0; i < m; ++i) std::swap(A(i, s), A(i, piv));
            std::swap(norms[s], norms[piv]);
        }
        double nrm = 0;
        for (size_t i = s; i < m; ++i) nrm += A(i, s) * A(i, s);
        nrm = std::sqrt(nrm);
        V v(m - s, 0.0);
        if (nrm > 0.0) {
            const double alpha = A(s, s) >= 0 ? -nrm : nrm;
            for (size_t i = s; i < m; ++i) v[i - s] = A(i, s);
            v[0] -= alpha;
            double vn = 0;
            for (double x : v) vn += x * x;
            if (vn > 0.0) {
                betas[s] = 2.0 / vn;
                for (size_t j = s; j < n; ++j) {           // apply H = I - beta v v' to the trailing columns
                    double d = 0;
                    for (size_t i = s; i < m; ++i) d += v[i - s] * A(i, j);
                    d *= betas[s];
                    for (size_t i = s; i < m; ++i) A(i, j) -= d * v[i - s];
                }
            }
        }
        vs[s] = std::move(v);
        for (size_t j = s + 1; j < n; ++j) {               // remaining norms, recomputed (robust to cancellation)
            double t = 0;
            for (size_t i = s + 1; i < m; ++i) t += A(i, j) * A(i, j);
            norms[j] = t;
        }
    }
    Mat Q(m, rank);                                        // Q = H_0 H_1 ... H_{rank-1} [I_rank; 0]
    for (size_t j = 0; j < rank; ++j) Q(j, j) = 1.0;
    for (size_t s = rank; s-- > 0;) {
        if (betas[s] == 0.0) continue;
        const V& v = vs[s];
        for (size_t j = 0; j < rank; ++j) {
            double d = 0;
            for (size_t i = s; i < m; ++i) d += v[i - s] * Q(i, j);
            d *= betas[s];
            for (size_t i = s; i < m; ++i) Q(i, j) -= d * v[i - s];
        }
    }
    return Q;
}

// Symmetric eigendecomposition A = Z diag(w) Z', w ascending: Householder tridiagonalisation then implicit QL with
// accumulated transformations (the classical tred2 / tql2 pair).  Returns false if QL does not converge or A is not finite.
bool eigh(const Mat& Ain, V& w, Mat& Z) {
    const size_t n = Ain.r;
    if (!all_finite(Ain.a)) return false;
    Z = Ain;
    w.assign(n, 0.0);
    V e(n, 0.0);
    if (n == 0) return true;
    // ---- tridiagonalise (on the lower triangle, row-oriented form)
    for (size_t j = 0; j < n; ++j) w[j] = Z(n - 1, j);
    for (size_t i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (size_t k = 0; k < i; ++k) scale += std::fabs(w[k]);
        if (scale == 0.0) {
            e[i] = w[i - 1];
            for (size_t j = 0; j < i; ++j) { w[j] = Z(i - 1, j); Z(i, j) = 0.0; Z(j, i) = 0.0; }
        } else {
            for (size_t k = 0; k < i; ++k) { w[k] /= scale; h += w[k] * w[k]; }
            double f = w[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            w[i - 1] = f - g;
            for (size_t j = 0; j < i; ++j) e[j] = 0.0;
            for (size_t j = 0; j < i; ++j) {
                f = w[j];
                Z(j, i) = f;
                g = e[j] + Z(j, j) * f;
                for (size_t k = j + 1; k <= i - 1; ++k) { g += Z(k, j) * w[k]; e[k] += Z(k, j) * f; }
                e[j] = g;
            }
            f = 0.0;
            for (size_t j = 0; j < i; ++j) { e[j] /= h; f += e[j] * w[j]; }
            const double hh = f / (h + h);
            for (size_t j = 0; j < i; ++j) e[j] -= hh * w[j];
            for (size_t j = 0; j < i; ++j) {
                f = w[j]; g = e[j];
                for (size_t k = j; k <= i - 1; ++k) Z(k, j) -= (f * e[k] + g * w[k]);
                w[j] = Z(i - 1, j);
                Z(i, j) = 0.0;
            }
        }
        w[i] = h;
    }
    for (size_t i = 0; i + 1 < n; ++i) {                   // accumulate the transformations
        Z(n - 1, i) = Z(i, i);
        Z(i, i) = 1.0;
        const double h = w[i + 1];
        if (h != 0.0) {
            for (size_t k = 0; k <= i; ++k) w[k] = Z(k, i + 1) / h;
            for (size_t j = 0; j <= i; ++j) {
                double g = 0.0;
                for (size_t k = 0; k <= i; ++k) g += Z(k, i + 1) * Z(k, j);
                for (size_t k = 0; k <= i; ++k) Z(k, j) -= g * w[k];
            }
        }
        for (size_t k = 0; k <= i; ++k) Z(k, i + 1) = 0.0;
    }
    for (size_t j = 0; j < n; ++j) { w[j] = Z(n - 1, j); Z(n - 1, j) = 0.0; }
    Z(n - 1, n - 1) = 1.0;
    e[0] = 0.0;
    // ---- implicit QL
    for (size_t i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (size_t l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(w[l]) + std::fabs(e[l]));
        size_t m = l;
        while (m < n) { if (std::fabs(e[m]) <= eps * tst1) break; ++m; }
        if (m == n) m = n - 1;
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 200) return false;
                double g = w[l];
                double p = (w[l + 1] - g) / (2.0 * e[l]);
                double r = std::hypot(p, 1.0);
                if (p < 0) r = -r;
                w[l] = e[l] / (p + r);
                w[l + 1] = e[l] * (p + r);
                const double dl1 = w[l + 1];
                double h = g - w[l];
                for (size_t i = l + 2; i < n; ++i) w[i] -= h;
                f += h;
                p = w[m];
                double c = 1.0, c2 = c, c3 = c;
                const double el1 = e[l + 1];
                double s = 0.0, s2 = 0.0;
                for (size_t i = m; i-- > l;) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = std::hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * w[i] - s * g;
                    w[i + 1] = h + s * (c * g + s * w[i]);
                    for (size_t k = 0; k < n; ++k) {
                        h = Z(k, i + 1);
                        Z(k, i + 1) = s * Z(k, i) + c * h;
                        Z(k, i) = c * Z(k, i) - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                w[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1);
        }
        w[l] += f;
        e[l] = 0.0;
    }
    for (size_t i = 0; i + 1 < n; ++i) {                   // ascending order
        size_t k = i;
        double p = w[i];
        for (size_t j = i + 1; j < n; ++j) if (w[j] < p) { k = j; p = w[j]; }
        if (k != i) {
            w[k] = w[i]; w[i] = p;
            for (size_t j = 0; j < n; ++j) std::swap(Z(j, i), Z(j, k));
        }
    }
    return all_finite(w) && all_finite(Z.a);
}

// U f(diag(w)) U'
Mat sym_fn(const Mat& U, const V& fw) {
    Mat T = U;
    for (size_t j = 0; j < U.c; ++j) for (size_t i = 0; i < U.r; ++i) T(i, j) *= fw[j];
    return matmul_nt(T, U);
}

// spd_mean (adapt/low_rank.rs:262-290): G^-1/2 (G^1/2 D G^1/2)^1/2 G^-1/2
bool spd_mean(const Mat& cov_draws, const Mat& cov_grads, Mat& out) {
    V w; Mat U;
    if (!eigh(cov_grads, w, U)) return false;
    V sq(w.size()), isq(w.size());
    for (size_t i = 0; i < w.size(); ++i) { sq[i] = std::sqrt(w[i]); isq[i] = 1.0 / std::sqrt(w[i]); }
    const Mat g_sqrt = sym_fn(U, sq);
    const Mat m = matmul(matmul(g_sqrt, cov_draws), g_sqrt);
    Mat msym = m;                                          // the eigensolver reads the lower triangle, like faer's Side::Lower
    V mw; Mat MU;
    if (!eigh(msym, mw, MU)) return false;
    // Eigenvalues of G^1/2 D G^1/2 below eps x its norm are rounding noise of either sign (the product is formed
    // explicitly; with gamma = 1e-5 its norm reaches 1e15 while directions outside both windows have eigenvalue ~1).  A
    // negative one would turn the whole update into NaN; they are floored at the noise level, which leaves those
    // directions with a geometric-mean eigenvalue of order 1 that the cutoff filter discards.
    double mw_max = 0.0;
    for (double v : mw) mw_max = std::max(mw_max, v);
    const double floor_ = 2.220446049250313e-16 * mw_max;
    for (double& v : mw) v = std::sqrt(std::max(v, floor_));
    const Mat m_sqrt = sym_fn(MU, mw);
    const Mat g_inv_sqrt = sym_fn(U, isq);
    out = matmul(matmul(g_inv_sqrt, m_sqrt), g_inv_sqrt);
    return all_finite(out.a);
}

}  // namespace

// declared in include/nuts_amd.h.  draws / grads: [n_draws][dim] (row = one draw); vecs out: [n_eig][dim].
// This file is compiled TWICE (nuts_rs_amd/build.py): plain x86-64 as nm_lowrank_compute_update_base, and with -mavx2 -mfma as
// nm_lowrank_compute_update_avx2 (its dense loops vectorise; no contraction: -ffp-contract=off, so both give the same bits);
// lowrank_dispatch.cpp exports nm_lowrank_compute_update and picks one by what the CPU it runs on supports (a library built here
// must not die with SIGILL on a host without AVX2).
#ifndef NM_LR_IMPL_NAME
#define NM_LR_IMPL_NAME nm_lowrank_compute_update_base
#endif
extern "C" int NM_LR_IMPL_NAME(void*, uint64_t dim_, uint64_t n_, const double* draws_in, const double* grads_in,
                                         double gamma, double eigval_cutoff, double* stds, double* mean, uint64_t* n_eig,
                                         double* vals_out, double* vecs_out, double* mu_out) {
    const size_t dim = dim_, n = n_;
    if (n == 0 || dim == 0) return 1;
    Mat draws(dim, n), grads(dim, n);                      // the reference's Mat<f64> of shape (ndim, ndraws)
    for (size_t i = 0; i < n; ++i)
        for (size_t d = 0; d < dim; ++d) { draws(d, i) = draws_in[i * dim + d]; grads(d, i) = grads_in[i * dim + d]; }
    // ---- rescale_points (:161-226)
    V draw_mean(dim), grad_mean(dim);
    const double nf = (double)n;
    for (size_t row = 0; row < dim; ++row) {
        double ds = 0, gs = 0;
        for (size_t i = 0; i < n; ++i) { ds += draws(row, i); gs += grads(row, i); }
        const double dm = ds / nf, gm = gs / nf;
        double dv = 0, gv = 0;
        for (size_t i = 0; i < n; ++i) {
            dv += (draws(row, i) - dm) * (draws(row, i) - dm);
            gv += (grads(row, i) - gm) * (grads(row, i) - gm);
        }
        dv /= nf; gv /= nf;
        const double sigma = std::sqrt(std::sqrt(dv / gv));
        mean[row] = dm + sigma * sigma * gm;
        stds[row] = sigma;
        const double draw_scale = 1.0 / sigma;
        for (size_t i = 0; i < n; ++i) { draws(row, i) = (draws(row, i) - mean[row]) * draw_scale; grads(row, i) = grads(row, i) * sigma; }
        double ds2 = 0, gs2 = 0;
        for (size_t i = 0; i < n; ++i) { ds2 += draws(row, i); gs2 += grads(row, i); }
        draw_mean[row] = ds2 / nf; grad_mean[row] = gs2 / nf;
        for (size_t i = 0; i < n; ++i) { draws(row, i) -= draw_mean[row]; grads(row, i) -= grad_mean[row]; }
    }
    if (!all_finite(draws.a) || !all_finite(grads.a)) return 1;   // faer's thin_svd fails on non-finite input: `.ok()?`
    // ---- subspace of the two windows (:80-88)
    const Mat ud = thin_q(draws), ug = thin_q(grads);
    Mat subspace(dim, ud.c + ug.c);
    std::memcpy(subspace.col(0), ud.a.data(), ud.a.size() * sizeof(double));
    std::memcpy(subspace.col(ud.c), ug.a.data(), ug.a.size() * sizeof(double));
    const Mat basis = thin_q(subspace);
    const Mat dp = matmul_tn(basis, draws), gp = matmul_tn(basis, grads);
    // ---- estimate_mass_matrix (:228-260)
    Mat cov_d = matmul_nt(dp, dp), cov_g = matmul_nt(gp, gp);
    const double ig = 1.0 / gamma;
    for (double& v : cov_d.a) v *= ig;
    for (double& v : cov_g.a) v *= ig;
    for (size_t i = 0; i < cov_d.r; ++i) { cov_d(i, i) += 1.0; cov_g(i, i) += 1.0; }
    Mat gmean;
    if (!spd_mean(cov_d, cov_g, gmean)) return 1;
    V vals; Mat vecs;
    if (!eigh(gmean, vals, vecs)) return 1;
    // ---- filter (:92-109), back-project, translation (:111-139)
    const double lo = 1.0 / eigval_cutoff;
    std::vector<size_t> keep;
    for (size_t i = 0; i < vals.size(); ++i) if ((vals[i] > eigval_cutoff) | (vals[i] < lo)) keep.push_back(i);
    const size_t r = keep.size();
    Mat sel(basis.c, r);
    for (size_t j = 0; j < r; ++j) std::memcpy(sel.col(j), vecs.col(keep[j]), basis.c * sizeof(double));
    const Mat full = matmul(basis, sel);                   // dim x r
    V b(r, 0.0);
    for (size_t j = 0; j < r; ++j) {
        double s = 0;
        for (size_t d = 0; d < dim; ++d) s += full(d, j) * grad_mean[d];
        b[j] = (vals[keep[j]] - 1.0) * s;
    }
    for (size_t d = 0; d < dim; ++d) {
        double s = 0;
        for (size_t j = 0; j < r; ++j) s += full(d, j) * b[j];
        mu_out[d] = draw_mean[d] + grad_mean[d] + s;
    }
    *n_eig = r;
    for (size_t j = 0; j < r; ++j) {
        vals_out[j] = vals[keep[j]];
        for (size_t d = 0; d < dim; ++d) vecs_out[j * dim + d] = full(d, j);
    }
    return 0;
}
