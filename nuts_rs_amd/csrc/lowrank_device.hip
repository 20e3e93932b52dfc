// lowrank_device.hip — the low-rank estimator ON THE DEVICE (VERDICT r03 item 7 / "missing" 2: the reference runs
// `LowRankMassMatrixStrategy::compute_update`, src/transform/adapt/low_rank.rs:73-142, in the thread that runs the chain; here it
// runs on the GPU that runs the chains): one block of 256 threads per paused chain, the block algorithm of lowrank_block.hpp, the
// chain's window read where the draw kernel left it (lrwin), the answer written in the packed form `lr_scatter_kernel` already
// takes.  Nothing crosses PCIe except the list of paused chains and a 3-word verdict per chain.
//
// Two instantiations of the same template code live here: the kernel, and the host TWIN (nm_lowrank_block_twin) that walks the
// same steps in loops with the same reduction trees; they agree bit for bit (tests/test_gpu_lowrank.py), so the reference's KATs
// and the tolerance ladder the CPU tests run against the twin are statements about the kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <vector>

#include "nuts_kernels.hpp"
#include "lowrank_block.hpp"

using namespace nm;
using namespace nm::lrb;

namespace {

#ifdef NM_LRB_PROF
__device__ unsigned long long lrb_prof[16];
#endif
struct DevEx {
    int tid; double* red; unsigned long long* redi;       // (the block's reduction cells: SmallT::red / redi)
#ifdef NM_LRB_PROF
    unsigned long long last = 0;
    __device__ __forceinline__ void mark(int id) {
        if (tid == 0) { const unsigned long long t = wall_clock64(); if (last) atomicAdd(&lrb_prof[id], t - last); last = t; }
    }
#else
    __device__ __forceinline__ void mark(int) {}
#endif
    template <class F> __device__ __forceinline__ void par(size_t n, F f) {
        __syncthreads();
        for (size_t i = (size_t)tid; i < n; i += LRB_T) f(i);
        __syncthreads();
    }
    template <class F> __device__ __forceinline__ void one(F f) {
        __syncthreads();
        if (tid == 0) f();
        __syncthreads();
    }
    // thread 0 runs g while the threads of the OTHER wavefronts run f over n independent items (g and f touch disjoint data: the QL iteration
    // on the tridiagonal / the application of the previous sweeps to the eigenvector matrix)
    template <class G, class F> __device__ __forceinline__ void split(G g, size_t n, F f) {
        __syncthreads();
        if (tid == 0) g();
        else if (tid >= 64) for (size_t i = (size_t)tid - 64; i < n; i += LRB_T - 64) f(i);
        __syncthreads();
    }
    template <class F> __device__ __forceinline__ double sum(size_t n, F f) {
        __syncthreads();
        double p = 0.0;
        for (size_t i = (size_t)tid; i < n; i += LRB_T) p += f(i);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p += __shfl_xor(p, off);
        if ((tid & 63) == 0) red[tid >> 6] = p;
        __syncthreads();
        return ((red[0] + red[1]) + red[2]) + red[3];
    }
    // item i: *addr(i) = f(i, *addr(i)); RMW_N items in flight per thread (their loads before their stores)
    static constexpr int RMW_N = 4;          // (16 measured: no faster, profiles/r05ag_*)
    template <class FA, class F> __device__ __forceinline__ void rmw(size_t n, FA addr, F f) {
        __syncthreads();
        for (size_t i = (size_t)tid; i < n; i += RMW_N * LRB_T) {
            double* p[RMW_N]; double v[RMW_N];
#pragma unroll
            for (int c = 0; c < RMW_N; ++c) if (i + c * LRB_T < n) { p[c] = addr(i + c * LRB_T); v[c] = *p[c]; }
#pragma unroll
            for (int c = 0; c < RMW_N; ++c) if (i + c * LRB_T < n) *p[c] = f(i + c * LRB_T, v[c]);
        }
        __syncthreads();
    }
    // the same over a rows x cols grid of items, *addr(k, j) = f(k, j, *addr(k, j)): a wavefront takes a column at a time (its lanes rows k = lane,
    // lane + 64, ...: neighbouring addresses in a column-major matrix), four rows in flight per lane; no division by the grid's shape
    template <class FA, class F> __device__ __forceinline__ void rmw2(size_t rows, size_t cols, FA addr, F f) {
        __syncthreads();
        const size_t lane = (size_t)(tid & 63);
        for (size_t j = (size_t)(tid >> 6); j < cols; j += LRB_T / 64) {
            for (size_t k = lane; k < rows; k += 4 * 64) {
                double* p[4]; double v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) if (k + c * 64 < rows) { p[c] = addr(k + c * 64, j); v[c] = *p[c]; }
#pragma unroll
                for (int c = 0; c < 4; ++c) if (k + c * 64 < rows) *p[c] = f(k + c * 64, j, v[c]);
            }
        }
        __syncthreads();
    }
    // out(j, sum_k f(k, j)) for j < ncols, k < len: one wavefront per column (four columns at a time), lane l sums k = l, l + 64, ...
    // then the butterfly
    template <class F, class FO> __device__ __forceinline__ void col_dots(size_t ncols, size_t len, F f, FO out) {
        __syncthreads();
        const size_t lane = (size_t)(tid & 63);
        for (size_t j0 = (size_t)(tid >> 6) * 4; j0 < ncols; j0 += 16) {
            double p[4] = {0.0, 0.0, 0.0, 0.0};
            for (size_t k = lane; k < len; k += 64) {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (j0 + c < ncols) p[c] += f(k, j0 + c);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) p[c] += __shfl_xor(p[c], off);
            }
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (j0 + c < ncols) out(j0 + c, p[c]);
            }
        }
        __syncthreads();
    }
    // largest f(i); ties and the all-NaN case go to the smallest index (val = -inf, idx = n when nothing compares)
    template <class F> __device__ __forceinline__ void argmax(size_t n, F f, double& val, size_t& idx) {
        __syncthreads();
        double bv = -HUGE_VAL; unsigned long long bi = (unsigned long long)n;
        for (size_t i = (size_t)tid; i < n; i += LRB_T) { const double v = f(i); if (v > bv) { bv = v; bi = i; } }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(bv, off); const unsigned long long oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { red[4 + (tid >> 6)] = bv; redi[tid >> 6] = bi; }
        __syncthreads();
        bv = red[4]; bi = redi[0];
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) { const double ov = red[4 + wv]; const unsigned long long oi = redi[wv]; if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; } }
        val = bv; idx = (size_t)bi;
    }
};

// the same steps, one thread: items in a loop, the reductions through the device's tree
struct SimEx {
    void mark(int) {}
    template <class F> void par(size_t n, F f) { for (size_t i = 0; i < n; ++i) f(i); }
    template <class F> void one(F f) { f(); }
    template <class G, class F> void split(G g, size_t n, F f) { g(); for (size_t i = 0; i < n; ++i) f(i); }
    template <class FA, class F> void rmw2(size_t rows, size_t cols, FA addr, F f) {
        for (size_t j = 0; j < cols; ++j) for (size_t k = 0; k < rows; ++k) { double* p = addr(k, j); *p = f(k, j, *p); }
    }
    template <class F> double sum(size_t n, F f) {
        double p[LRB_T];
        for (int t = 0; t < LRB_T; ++t) p[t] = 0.0;
        for (size_t i = 0; i < n; ++i) p[i % LRB_T] += f(i);
        for (int off = 32; off >= 1; off >>= 1) {
            double q[LRB_T];
            for (int t = 0; t < LRB_T; ++t) q[t] = p[t] + p[t ^ off];
            memcpy(p, q, sizeof(p));
        }
        return ((p[0] + p[64]) + p[128]) + p[192];
    }
    template <class FA, class F> void rmw(size_t n, FA addr, F f) { for (size_t i = 0; i < n; ++i) { double* p = addr(i); *p = f(i, *p); } }
    template <class F, class FO> void col_dots(size_t ncols, size_t len, F f, FO out) {
        for (size_t j = 0; j < ncols; ++j) {
            double p[64];
            for (int t = 0; t < 64; ++t) p[t] = 0.0;
            for (size_t k = 0; k < len; ++k) p[k % 64] += f(k, j);
            for (int off = 32; off >= 1; off >>= 1) {
                double q[64];
                for (int t = 0; t < 64; ++t) q[t] = p[t] + p[t ^ off];
                memcpy(p, q, sizeof(p));
            }
            out(j, p[0]);
        }
    }
    template <class F> void argmax(size_t n, F f, double& val, size_t& idx) {
        double bv = -HUGE_VAL; size_t bi = n;             // the tree's winner is the first largest element: order-free
        for (size_t i = 0; i < n; ++i) { const double v = f(i); if (v > bv) { bv = v; bi = i; } }
        val = bv; idx = bi;
    }
};

struct LnDet { __host__ __device__ double operator()(double x) const { return dlog(x); } };

}  // namespace

namespace nm { namespace lrb {

// jobs[3 i] = {chain, first row of its window in lrwin, rows}.  Per job i: rows_out + i * rows * dp ((4 + rmax) rows of dp doubles),
// vals2 + i * 2 * rmax, meta[3 i] = {chain, n_eig, ok}; *too_wide counts answers of rank > rmax.  sc (may be null): the chains'
// scalars, where the answer's rank, log-determinant and LR_ANSWERED go.  vals_plain (may be null): [job][rmax] eigenvalues.
template <size_t KM>
__global__ __launch_bounds__(LRB_T) void lr_estimate_kernel(const uint64_t* jobs, uint64_t n_jobs, uint64_t dim, const double* lrwin, uint64_t lr_cap,
                                                           double gamma, double eigval_cutoff, double* scratch, uint64_t scratch_stride,
                                                           double* rows_out, uint64_t rows, uint64_t dp, double* vals2, uint64_t rmax, uint64_t* meta,
                                                           unsigned long long* too_wide, ChainScalars* sc, double* vals_plain) {
    __shared__ SmallT<KM> S;                               // KM = 256: dim <= 256 (25 KiB: several blocks per compute unit); 512 beyond
    dm_init_lds();                                         // dlog's tables (the log-determinant)
    DevEx ex{(int)threadIdx.x, S.red, S.redi};
    for (uint64_t job = blockIdx.x; job < n_jobs; job += gridDim.x) {
        const uint64_t c = jobs[3 * job], start = jobs[3 * job + 1], len = jobs[3 * job + 2];
        const double* win = lrwin + ((size_t)c * lr_cap + start) * 2 * dim;
        Out out{rows_out + (size_t)job * rows * dp, (size_t)dp, vals2 + (size_t)job * 2 * rmax, (size_t)rmax, vals_plain ? vals_plain + (size_t)job * rmax : nullptr};
        size_t n_eig = 0; double logdet = 0.0;
        const int rc = estimate(ex, S, (size_t)dim, (size_t)len, win, gamma, eigval_cutoff, scratch + (size_t)blockIdx.x * scratch_stride, out, &n_eig, &logdet, LnDet());
        __syncthreads();
        if (threadIdx.x == 0) {
            const bool ok = rc == LRB_OK;
            meta[3 * job] = c; meta[3 * job + 1] = ok ? n_eig : 0; meta[3 * job + 2] = ok ? 1 : 0;
            if (rc == LRB_TOO_WIDE) atomicAdd(too_wide, 1ull);
            if (sc) {
                ChainScalars& q = sc[c];
                q.lr_upd_ok = ok ? 1 : 0; q.lr_upd_rank = ok ? n_eig : 0; q.lr_upd_logdet = ok ? logdet : 0.0;
                q.lr_pending = LR_ANSWERED;
            }
        }
        __syncthreads();
    }
}

#ifdef NM_LRB_PROF
// tuning builds: wall-clock ticks (100 MHz) per phase, summed over blocks; out[16]
extern "C" int nm_debug_lrb_prof(unsigned long long* out, int reset) {
    hipError_t er = hipMemcpyFromSymbol(out, HIP_SYMBOL(lrb_prof), sizeof(lrb_prof));
    if (er == hipSuccess && reset) { unsigned long long z[16] = {0}; er = hipMemcpyToSymbol(HIP_SYMBOL(lrb_prof), z, sizeof(z)); }
    return (int)er;
}
#endif
bool block_estimator_takes(uint64_t dim, uint64_t n_max) { return dim >= 1 && dim <= LRB_KMAX && n_max >= 1 && n_max <= LRB_NMAX; }
size_t block_estimator_scratch_doubles(uint64_t dim, uint64_t n_max) { return scratch_doubles((size_t)dim, (size_t)n_max); }

hipError_t launch_estimate(unsigned grid, hipStream_t stream, const uint64_t* d_jobs, uint64_t n_jobs, uint64_t dim, const double* d_lrwin, uint64_t lr_cap,
                           double gamma, double eigval_cutoff, double* d_scratch, uint64_t scratch_stride, double* d_rows, uint64_t rows, uint64_t dp,
                           double* d_vals2, uint64_t rmax, uint64_t* d_meta, unsigned long long* d_too_wide, void* d_sc, double* d_vals_plain) {
    if (dim <= 256)
        hipLaunchKernelGGL(lr_estimate_kernel<256>, dim3(grid), dim3(LRB_T), 0, stream, d_jobs, n_jobs, dim, d_lrwin, lr_cap, gamma, eigval_cutoff, d_scratch,
                           scratch_stride, d_rows, rows, dp, d_vals2, rmax, d_meta, d_too_wide, (ChainScalars*)d_sc, d_vals_plain);
    else
        hipLaunchKernelGGL(lr_estimate_kernel<LRB_KMAX>, dim3(grid), dim3(LRB_T), 0, stream, d_jobs, n_jobs, dim, d_lrwin, lr_cap, gamma, eigval_cutoff, d_scratch,
                           scratch_stride, d_rows, rows, dp, d_vals2, rmax, d_meta, d_too_wide, (ChainScalars*)d_sc, d_vals_plain);
    return hipGetLastError();
}

}}  // namespace nm::lrb

// ---- the twin: nm_lowrank_estimator_fn's signature (include/nuts_amd.h), host only
extern "C" int nm_lowrank_block_twin(void*, uint64_t dim, uint64_t n, const double* draws, const double* grads, double gamma, double eigval_cutoff,
                                     double* stds, double* mean, uint64_t* n_eig, double* vals, double* vecs, double* mu_low_rank) {
    if (!block_estimator_takes(dim, n)) return 1;
    std::vector<double> win((size_t)n * 2 * dim);
    for (uint64_t r = 0; r < n; ++r) {
        memcpy(&win[(2 * r) * dim], &draws[r * dim], dim * sizeof(double));
        memcpy(&win[(2 * r + 1) * dim], &grads[r * dim], dim * sizeof(double));
    }
    const size_t rmax = (size_t)std::min<uint64_t>(dim, 2 * n);
    std::vector<double> scratch(scratch_doubles(dim, n), 0.0), rows((4 + rmax) * dim, 0.0), v2(2 * rmax, 0.0), vp(rmax, 0.0);
    static thread_local Small S;
    SimEx ex;
    Out out{rows.data(), (size_t)dim, v2.data(), rmax, vp.data()};
    size_t r = 0; double logdet = 0.0;
    const int rc = estimate(ex, S, (size_t)dim, (size_t)n, win.data(), gamma, eigval_cutoff, scratch.data(), out, &r, &logdet, LnDet());
    if (rc != LRB_OK) return 1;
    memcpy(stds, &rows[0], dim * sizeof(double));
    memcpy(mean, &rows[2 * dim], dim * sizeof(double));
    memcpy(mu_low_rank, &rows[3 * dim], dim * sizeof(double));
    *n_eig = r;
    for (size_t j = 0; j < r; ++j) { vals[j] = vp[j]; memcpy(&vecs[j * dim], &rows[(4 + j) * dim], dim * sizeof(double)); }
    return 0;
}

// ---- the twin's spd_mean / estimate_mass_matrix, for the reference's own estimator vectors (adapt/low_rank.rs:354-407): reached
// through nm_lowrank_test_spd_mean / nm_lowrank_test_estimate_mass_matrix with force_base = 2 (lowrank_host.cpp)
extern "C" int nm_lowrank_block_twin_spd_mean(uint64_t n, const double* cov_draws, const double* cov_grads, double* out) {
    if (n == 0 || n > LRB_KMAX) return 1;
    std::vector<double> buf(5 * n * n, 0.0);
    memcpy(&buf[0], cov_draws, n * n * sizeof(double));
    memcpy(&buf[n * n], cov_grads, n * n * sizeof(double));
    static thread_local Small S;
    SimEx ex;
    const size_t nn = (size_t)n;
    const M covd = colmajor(&buf[0], nn), covg = colmajor(&buf[nn * nn], nn), t1 = colmajor(&buf[2 * nn * nn], nn), t2 = colmajor(&buf[3 * nn * nn], nn), t3 = colmajor(&buf[4 * nn * nn], nn);
    if (!spd_mean(ex, S, covd, covg, t1, t2, t3, (size_t)n)) return 1;
    memcpy(out, &buf[0], n * n * sizeof(double));
    return 0;
}
extern "C" int nm_lowrank_block_twin_estimate_mass_matrix(uint64_t rows, uint64_t n_draws, const double* draws, const double* grads, double gamma,
                                                          double* vals, double* vecs) {
    if (rows == 0 || rows > LRB_KMAX) return 1;
    const size_t K = (size_t)rows, n = (size_t)n_draws;
    std::vector<double> buf(5 * K * K, 0.0);
    static thread_local Small S;
    SimEx ex;
    const M covd = colmajor(&buf[0], K), covg = colmajor(&buf[K * K], K), t1 = colmajor(&buf[2 * K * K], K), t2 = colmajor(&buf[3 * K * K], K), t3 = colmajor(&buf[4 * K * K], K);
    const M DP = colmajor(const_cast<double*>(draws), K), GP = colmajor(const_cast<double*>(grads), K);
    const double ig = 1.0 / gamma;
    ex.par(K * K, [&](size_t idx) {
        const size_t i = idx % K, j = idx / K;
        double sd = 0.0, sg = 0.0;
        for (size_t l = 0; l < n; ++l) { sd += DP(i, l) * DP(j, l); sg += GP(i, l) * GP(j, l); }
        sd *= ig; sg *= ig;
        if (i == j) { sd += 1.0; sg += 1.0; }
        covd(i, j) = sd; covg(i, j) = sg;
    });
    if (!spd_mean(ex, S, covd, covg, t1, t2, t3, K)) return 1;
    if (!eigh(ex, S, covd, K)) return 1;
    for (size_t t = 0; t < K; ++t) { vals[t] = S.w[S.order[t]]; memcpy(&vecs[t * K], &buf[(size_t)S.order[t] * K], K * sizeof(double)); }
    return 0;
}

// ---- the kernel on a batch of windows of one shape (test hook): draws / grads [n_windows][n][dim]; outputs with a leading
// [n_windows] axis, vals [kmax], vecs [kmax][dim] with kmax = min(dim, 2 n); status[w] = 0 (Some) / 1 (None).  Returns a hipError_t.
extern "C" int nm_lowrank_test_block_device(uint64_t dim, uint64_t n, uint64_t n_windows, const double* draws, const double* grads, double gamma,
                                            double eigval_cutoff, double* stds, double* mean, uint64_t* n_eig, double* vals, double* vecs,
                                            double* mu_low_rank, uint64_t* status, uint64_t* logdet_bits) {
    if (!block_estimator_takes(dim, n) || n_windows == 0) return (int)hipErrorInvalidValue;
    const size_t rmax = (size_t)std::min<uint64_t>(dim, 2 * n), rows = 4 + rmax, nw = (size_t)n_windows;
    std::vector<double> win(nw * n * 2 * dim);
    for (size_t w = 0; w < nw; ++w)
        for (uint64_t r = 0; r < n; ++r) {
            memcpy(&win[(w * n + r) * 2 * dim], &draws[(w * n + r) * dim], dim * sizeof(double));
            memcpy(&win[(w * n + r) * 2 * dim + dim], &grads[(w * n + r) * dim], dim * sizeof(double));
        }
    std::vector<uint64_t> jobs(3 * nw);
    for (size_t w = 0; w < nw; ++w) { jobs[3 * w] = w; jobs[3 * w + 1] = 0; jobs[3 * w + 2] = n; }
    const unsigned grid = (unsigned)std::min<size_t>(nw, 512);
    const size_t ss = scratch_doubles(dim, n);
    double *d_win = nullptr, *d_scr = nullptr, *d_rows = nullptr, *d_v2 = nullptr, *d_vp = nullptr;
    uint64_t *d_jobs = nullptr, *d_meta = nullptr; unsigned long long* d_tw = nullptr; ChainScalars* d_sc = nullptr;
    hipError_t er = hipSuccess;
    auto T = [&](hipError_t x) { if (er == hipSuccess) er = x; };
    hipStream_t st = nullptr;                              // (stream discipline: the hook's own stream, nothing on the null stream)
    T(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto up = [&](void* d, const void* h, size_t n) { T(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st)); };
    auto down = [&](void* h, const void* d, size_t n) { T(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st)); };
    auto zero = [&](void* d, size_t n) { T(hipMemsetAsync(d, 0, n, st)); };
    T(hipMalloc((void**)&d_win, win.size() * 8)); T(hipMalloc((void**)&d_scr, ss * grid * 8)); T(hipMalloc((void**)&d_rows, nw * rows * dim * 8));
    T(hipMalloc((void**)&d_v2, nw * 2 * rmax * 8)); T(hipMalloc((void**)&d_vp, nw * rmax * 8)); T(hipMalloc((void**)&d_jobs, jobs.size() * 8));
    T(hipMalloc((void**)&d_meta, jobs.size() * 8)); T(hipMalloc((void**)&d_tw, 8)); T(hipMalloc((void**)&d_sc, nw * sizeof(ChainScalars)));
    if (er == hipSuccess) {
        up(d_win, win.data(), win.size() * 8);
        up(d_jobs, jobs.data(), jobs.size() * 8);
        zero(d_tw, 8); zero(d_sc, nw * sizeof(ChainScalars)); zero(d_rows, nw * rows * dim * 8);
        T(launch_estimate(grid, st, d_jobs, nw, dim, d_win, n, gamma, eigval_cutoff, d_scr, ss, d_rows, rows, dim, d_v2, rmax, d_meta, d_tw, d_sc, d_vp));
        T(hipStreamSynchronize(st));
    }
    if (er == hipSuccess) {
        std::vector<double> h_rows(nw * rows * dim), h_vp(nw * rmax);
        std::vector<uint64_t> meta(3 * nw);
        std::vector<ChainScalars> sc(nw);
        down(h_rows.data(), d_rows, h_rows.size() * 8); down(h_vp.data(), d_vp, h_vp.size() * 8);
        down(meta.data(), d_meta, meta.size() * 8); down(sc.data(), d_sc, nw * sizeof(ChainScalars));
        T(hipStreamSynchronize(st));
        if (er == hipSuccess)
            for (size_t w = 0; w < nw; ++w) {
                const double* R = &h_rows[w * rows * dim];
                const bool ok = meta[3 * w + 2] != 0;
                status[w] = ok ? 0 : 1;
                n_eig[w] = ok ? meta[3 * w + 1] : 0;
                if (logdet_bits) memcpy(&logdet_bits[w], &sc[w].lr_upd_logdet, 8);
                if (!ok) continue;
                memcpy(&stds[w * dim], R, dim * 8); memcpy(&mean[w * dim], R + 2 * dim, dim * 8); memcpy(&mu_low_rank[w * dim], R + 3 * dim, dim * 8);
                for (size_t j = 0; j < meta[3 * w + 1]; ++j) { vals[w * rmax + j] = h_vp[w * rmax + j]; memcpy(&vecs[(w * rmax + j) * dim], R + (4 + j) * dim, dim * 8); }
            }
    }
    (void)hipFree(d_win); (void)hipFree(d_scr); (void)hipFree(d_rows); (void)hipFree(d_v2); (void)hipFree(d_vp); (void)hipFree(d_jobs); (void)hipFree(d_meta);
    (void)hipFree(d_tw); (void)hipFree(d_sc);
    if (st) (void)hipStreamDestroy(st);
    return (int)er;
}
