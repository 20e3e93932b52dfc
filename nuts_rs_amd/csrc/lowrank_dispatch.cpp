// lowrank_dispatch.cpp — nm_lowrank_compute_update (include/nuts_amd.h): the host estimator of the low-rank adaptation, in the build
// the running CPU supports.  lowrank_host.cpp holds the algorithm and is compiled twice (see there).
#include <cstdint>

extern "C" int nm_lowrank_compute_update_base(void*, uint64_t, uint64_t, const double*, const double*, double, double, double*, double*,
                                              uint64_t*, double*, double*, double*);
extern "C" int nm_lowrank_compute_update_avx2(void*, uint64_t, uint64_t, const double*, const double*, double, double, double*, double*,
                                              uint64_t*, double*, double*, double*);

extern "C" int nm_lowrank_compute_update(void* ctx, uint64_t dim, uint64_t n_draws, const double* draws, const double* grads, double gamma,
                                         double eigval_cutoff, double* stds, double* mean, uint64_t* n_eig, double* vals, double* vecs,
                                         double* mu_low_rank) {
    static const bool wide = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    return (wide ? nm_lowrank_compute_update_avx2 : nm_lowrank_compute_update_base)(ctx, dim, n_draws, draws, grads, gamma, eigval_cutoff,
                                                                                    stds, mean, n_eig, vals, vecs, mu_low_rank);
}
