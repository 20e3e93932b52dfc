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
//
// One translation unit, two builds of the algorithm (lowrank_impl.inc): namespace lr_base for the baseline ISA and lr_avx2 with
// target("avx2,fma") on every function of the namespace (the dense loops vectorise; no contraction: -ffp-contract=off, so both
// give the same bits).  The exported entry points pick one by what the CPU they run on supports — a library built here must
// not die with SIGILL on a host without AVX2.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#define NM_LR_NS lr_base
#include "lowrank_impl.inc"
#undef NM_LR_NS

#pragma clang attribute push(__attribute__((target("avx2,fma"))), apply_to = function)
#define NM_LR_NS lr_avx2
#include "lowrank_impl.inc"
#undef NM_LR_NS
#pragma clang attribute pop

static bool lr_wide() {
    static const bool wide = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    return wide;
}

// declared in include/nuts_amd.h
extern "C" int nm_lowrank_compute_update(void*, uint64_t dim, uint64_t n_draws, const double* draws, const double* grads, double gamma,
                                         double eigval_cutoff, double* stds, double* mean, uint64_t* n_eig, double* vals, double* vecs,
                                         double* mu_low_rank) {
    return (lr_wide() ? lr_avx2::compute_update : lr_base::compute_update)(dim, n_draws, draws, grads, gamma, eigval_cutoff, stds, mean, n_eig,
                                                                          vals, vecs, mu_low_rank);
}
// (lowrank_device.hip: the same two routines of the block form's twin; force_base == 2 selects them)
extern "C" int nm_lowrank_block_twin_spd_mean(uint64_t n, const double* cov_draws, const double* cov_grads, double* out);
extern "C" int nm_lowrank_block_twin_estimate_mass_matrix(uint64_t rows, uint64_t n_draws, const double* draws, const double* grads, double gamma,
                                                          double* vals, double* vecs);
extern "C" int nm_lowrank_test_spd_mean(uint64_t n, const double* cov_draws, const double* cov_grads, double* out, uint64_t force_base) {
    if (force_base == 2) return nm_lowrank_block_twin_spd_mean(n, cov_draws, cov_grads, out);
    return (lr_wide() && !force_base ? lr_avx2::hook_spd_mean : lr_base::hook_spd_mean)(n, cov_draws, cov_grads, out);
}
extern "C" int nm_lowrank_test_estimate_mass_matrix(uint64_t rows, uint64_t n_draws, const double* draws, const double* grads, double gamma,
                                                    double* vals, double* vecs, uint64_t force_base) {
    if (force_base == 2) return nm_lowrank_block_twin_estimate_mass_matrix(rows, n_draws, draws, grads, gamma, vals, vecs);
    return (lr_wide() && !force_base ? lr_avx2::hook_estimate_mass_matrix : lr_base::hook_estimate_mass_matrix)(rows, n_draws, draws, grads, gamma, vals, vecs);
}
