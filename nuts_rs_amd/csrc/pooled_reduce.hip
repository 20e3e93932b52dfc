// pooled_reduce.hip — the per-rank half of the OPT-IN pooled adaptation (north_star's "RCCL cross-chain Welford reduction",
// SURVEY §8(e); not reference behaviour: every reference chain adapts alone, src/adapt_strategy.rs:24-39).
// One window of recorded draws and gradients ([rows][dim], rows = window draws x local chains, as nm_engine_draw_ex leaves them
// in device buffers) is reduced to (count, mean[dim], M2[dim]) for the draws and for the gradients — the payload the ranks then
// exchange with ONE all_gather (nuts_rs_amd/pooled.py) and merge in rank order with Chan's formula.  Rows are filtered the way
// the reference's DrawGradCollector filters them (src/transform/adapt/diagonal.rs:57-84): the chain is healthy and the draw
// `is_good` (index_in_trajectory != 0, or |index| > 4 for a divergent draw); rows of chains that stopped are never written by the
// engine and carry all-zero statistics in a zero-initialised buffer, which this filter rejects.
// Deterministic: fixed chunks of rows, each reduced by one thread per dimension in row order (Welford), chunk partials merged in
// chunk order — the same rows give the same bits whatever else runs on the device.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <string>
#include "../../include/nuts_amd.h"

namespace {
constexpr int CHUNK_ROWS = 512;

__device__ inline bool row_ok(const nm_draw_stats* st, uint64_t row) {
    if (!st) return true;
    const nm_draw_stats& s = st[row];
    if (s.chain_status != NM_CHAIN_OK) return false;
    const int64_t idx = s.index_in_trajectory;
    return s.diverging ? ((idx < 0 ? -idx : idx) > 4) : (idx != 0);
}

// grid (chunks, ceil(dim / 128), 2): z = 0 draws, 1 gradients.  partial[(z * chunks + chunk) * (1 + 2 dim)] = count, mean, M2
__global__ __launch_bounds__(128) void pooled_chunk_kernel(uint64_t n_rows, uint64_t dim, const double* x, const double* g,
                                                           const nm_draw_stats* st, double* partial) {
    const uint64_t d = (uint64_t)blockIdx.y * 128u + threadIdx.x;
    const uint64_t chunk = blockIdx.x, chunks = gridDim.x;
    const double* src = blockIdx.z ? g : x;
    const uint64_t r0 = chunk * CHUNK_ROWS, r1 = r0 + CHUNK_ROWS < n_rows ? r0 + CHUNK_ROWS : n_rows;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (uint64_t r = r0; r < r1; ++r) {
        if (!row_ok(st, r)) continue;                 // (uniform over the block: one row at a time)
        if (d < dim) {
            const double v = src[r * dim + d];
            n += 1.0;
            const double delta = v - mean;
            mean += delta / n;
            m2 += delta * (v - mean);
        } else {
            n += 1.0;
        }
    }
    double* out = partial + ((uint64_t)blockIdx.z * chunks + chunk) * (1 + 2 * dim);
    if (d == 0) out[0] = n;
    if (d < dim) { out[1 + d] = mean; out[1 + dim + d] = m2; }
}
// one block per (z): thread per dimension, chunk partials merged in chunk order (Chan et al.)
__global__ __launch_bounds__(128) void pooled_merge_kernel(uint64_t chunks, uint64_t dim, const double* partial, double* out) {
    const uint64_t d = (uint64_t)blockIdx.y * 128u + threadIdx.x;
    const double* p = partial + (uint64_t)blockIdx.z * chunks * (1 + 2 * dim);
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (uint64_t c = 0; c < chunks; ++c) {
        const double* q = p + c * (1 + 2 * dim);
        const double nb = q[0];
        if (nb == 0.0) continue;
        if (d < dim) {
            const double mb = q[1 + d], sb = q[1 + dim + d], tot = n + nb, delta = mb - mean;
            mean = mean + delta * (nb / tot);
            m2 = m2 + sb + delta * delta * (n * nb / tot);
            n = tot;
        } else {
            n += nb;
        }
    }
    double* o = out + (uint64_t)blockIdx.z * (1 + 2 * dim);
    if (d == 0) o[0] = n;
    if (d < dim) { o[1 + d] = mean; o[1 + dim + d] = m2; }
}
// the ranks' gathered partials [world][2][1 + 2 dim] -> the pooled diagonal transformation (pooled.py's _merge_payloads + finish)
__global__ __launch_bounds__(128) void pooled_finish_kernel(uint64_t world, uint64_t dim, const double* gathered, double* sigma, double* mean_out, double* count) {
    const uint64_t d = (uint64_t)blockIdx.x * 128u + threadIdx.x;
    const uint64_t stride = 2 * (1 + 2 * dim);
    double n[2] = {0.0, 0.0}, mean[2] = {0.0, 0.0}, m2[2] = {0.0, 0.0};
    for (int z = 0; z < 2; ++z)
        for (uint64_t r = 0; r < world; ++r) {                    // rank order
            const double* q = gathered + r * stride + (uint64_t)z * (1 + 2 * dim);
            const double nb = q[0];
            if (nb == 0.0) continue;
            if (d < dim) {
                const double mb = q[1 + d], sb = q[1 + dim + d];
                if (n[z] == 0.0) { n[z] = nb; mean[z] = mb; m2[z] = sb; continue; }
                const double tot = n[z] + nb, delta = mb - mean[z];
                mean[z] = mean[z] + delta * (nb / tot);
                m2[z] = m2[z] + sb + delta * delta * (n[z] * nb / tot);
                n[z] = tot;
            } else {
                n[z] += nb;
            }
        }
    if (d == 0 && count) *count = n[0];
    if (d < dim) {
        double sg = sqrt(sqrt(m2[0] / m2[1]));
        const bool ok = (sg == sg) && sg > 0.0 && sg < __builtin_inf();
        sg = ok ? fmin(fmax(sg, 1e-10), 1e10) : 1.0;
        sigma[d] = sg;
        mean_out[d] = mean[0] + sg * sg * mean[1];
    }
}
typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
all_gather_fn rccl_all_gather() {
    static all_gather_fn fn = [] {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        return h ? (all_gather_fn)dlsym(h, "ncclAllGather") : (all_gather_fn) nullptr;
    }();
    return fn;
}
thread_local std::string g_perr;
}  // namespace

extern "C" nm_status nm_pooled_exchange(void* rccl_comm, uint64_t world, uint64_t dim, const double* d_payload, double* d_gathered, void* stream) {
    if (!d_payload || !d_gathered || dim == 0 || world == 0) { g_perr = "nm_pooled_exchange: null argument"; return NM_ERR_INVALID_ARG; }
    const size_t n = 2 * (1 + 2 * (size_t)dim);
    hipStream_t s = (hipStream_t)stream;
    if (world == 1 || !rccl_comm) {
        const hipError_t e = hipMemcpyAsync(d_gathered, d_payload, n * sizeof(double), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { g_perr = std::string("nm_pooled_exchange: ") + hipGetErrorString(e); return NM_ERR_HIP; }
        return NM_OK;
    }
    all_gather_fn ag = rccl_all_gather();
    if (!ag) { g_perr = "nm_pooled_exchange: librccl.so (ncclAllGather) could not be loaded"; return NM_ERR_UNSUPPORTED; }
    const int rc = ag(d_payload, d_gathered, n, /* ncclDouble */ 8, rccl_comm, s);
    if (rc != 0) { g_perr = "nm_pooled_exchange: ncclAllGather failed with code " + std::to_string(rc); return NM_ERR_HIP; }
    return NM_OK;
}
extern "C" nm_status nm_pooled_finish(uint64_t world, uint64_t dim, const double* d_gathered, double* d_sigma, double* d_mean, double* d_count, void* stream) {
    if (!d_gathered || !d_sigma || !d_mean || dim == 0 || world == 0) { g_perr = "nm_pooled_finish: null argument"; return NM_ERR_INVALID_ARG; }
    hipLaunchKernelGGL(pooled_finish_kernel, dim3((unsigned)((dim + 127) / 128)), dim3(128), 0, (hipStream_t)stream, world, dim, d_gathered, d_sigma, d_mean, d_count);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_perr = std::string("nm_pooled_finish: ") + hipGetErrorString(e); return NM_ERR_HIP; }
    return NM_OK;
}

extern "C" const char* nm_pooled_last_error(void) { return g_perr.c_str(); }

// declared in include/nuts_amd.h
extern "C" nm_status nm_pooled_partials(uint64_t n_rows, uint64_t dim, const double* d_positions, const double* d_gradients,
                                        const nm_draw_stats* d_stats, double* d_out, void* stream) {
    if (!d_positions || !d_gradients || !d_out || dim == 0) { g_perr = "nm_pooled_partials: null argument"; return NM_ERR_INVALID_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_perr = "no HIP device available; there is no CPU fallback"; return NM_ERR_NO_DEVICE; }
    hipStream_t s = (hipStream_t)stream;
    const uint64_t chunks = n_rows ? (n_rows + CHUNK_ROWS - 1) / CHUNK_ROWS : 1;
    double* partial = nullptr;
    hipError_t e = hipMallocAsync((void**)&partial, 2 * chunks * (1 + 2 * dim) * sizeof(double), s);
    if (e != hipSuccess) { g_perr = std::string("nm_pooled_partials: ") + hipGetErrorString(e); return NM_ERR_HIP; }
    const dim3 block(128), gridc((unsigned)chunks, (unsigned)((dim + 127) / 128), 2), gridm(1, (unsigned)((dim + 127) / 128), 2);
    hipLaunchKernelGGL(pooled_chunk_kernel, gridc, block, 0, s, n_rows, dim, d_positions, d_gradients, d_stats, partial);
    hipLaunchKernelGGL(pooled_merge_kernel, gridm, block, 0, s, chunks, dim, (const double*)partial, d_out);
    e = hipGetLastError();
    (void)hipFreeAsync(partial, s);
    if (e != hipSuccess) { g_perr = std::string("nm_pooled_partials: ") + hipGetErrorString(e); return NM_ERR_HIP; }
    return NM_OK;
}
