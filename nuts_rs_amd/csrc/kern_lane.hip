// kern_lane.hip — the one-chain-per-lane draw kernels (nuts_lane.hpp) for the built-in densities with dim <= 16.
#include <hip/hip_runtime.h>
#include "nuts_lane.hpp"
namespace nm {
namespace {
template <class Dens, int NP>
hipError_t launch_lane_t(int query, bool tune, const KParams& P, const lane::LaneParams& LP, unsigned grid, hipStream_t stream, int* occ) {
    if constexpr (std::is_void<typename lane::LaneDensity<Dens, NP>::type>::value) {
        return hipErrorInvalidValue;
    } else {
        if (query) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, lane::nuts_lane_draw_kernel<Dens, NP, true>, 64, 0);
        if (tune) hipLaunchKernelGGL((lane::nuts_lane_draw_kernel<Dens, NP, true>), dim3(grid), dim3(64), 0, stream, P, LP);
        else hipLaunchKernelGGL((lane::nuts_lane_draw_kernel<Dens, NP, false>), dim3(grid), dim3(64), 0, stream, P, LP);
        return hipGetLastError();
    }
}
template <class Dens>
hipError_t launch_lane_d(int query, bool tune, const KParams& P, const lane::LaneParams& LP, unsigned grid, hipStream_t stream, int* occ) {
    switch (lane::lane_pairs(P.dim)) {
    case 2: return launch_lane_t<Dens, 2>(query, tune, P, LP, grid, stream, occ);
    case 4: return launch_lane_t<Dens, 4>(query, tune, P, LP, grid, stream, occ);
    case 5: return launch_lane_t<Dens, 5>(query, tune, P, LP, grid, stream, occ);
    case 8: return launch_lane_t<Dens, 8>(query, tune, P, LP, grid, stream, occ);
    }
    return hipErrorInvalidValue;
}
// ---- chains into wavefronts by step size (a counting sort over 2048 logarithmic buckets; the order inside a bucket is whatever the
// atomics make it: no result depends on it) ----
__device__ inline int lane_sort_bucket(double step) {
    const uint64_t b = d2u(step);
    if ((b >> 63) || !(step == step)) return lane::LANE_SORT_BUCKETS - 1;
    const int e = (int)(b >> 52) - (1023 - 40);                 // 2^-40 .. 2^23 in 64 octaves of 32 steps
    if (e < 0) return 0;
    if (e > 63) return lane::LANE_SORT_BUCKETS - 1;
    return e * 32 + (int)((b >> 47) & 31);
}
__global__ void lane_sort_hist(const ChainScalars* sc, uint64_t n, uint32_t* hist) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) atomicAdd(&hist[lane_sort_bucket(sc[c].step_size)], 1u);
}
__global__ void lane_sort_scan(uint32_t* hist) {                // one block of 1024 threads, two buckets each: exclusive prefix sums in place
    __shared__ uint32_t part[1024];
    const int t = (int)threadIdx.x;
    const uint32_t a = hist[2 * t], b = hist[2 * t + 1];
    part[t] = a + b;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const uint32_t before = part[t] - (a + b);
    hist[2 * t] = before; hist[2 * t + 1] = before + a;
}
__global__ void lane_sort_scatter(const ChainScalars* sc, uint64_t n, uint32_t* offs, uint32_t* perm) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) perm[atomicAdd(&offs[lane_sort_bucket(sc[c].step_size)], 1u)] = (uint32_t)c;
}
}  // namespace
hipError_t lane_sort_chains(const ChainScalars* sc, uint64_t n, uint32_t* hist, uint32_t* perm, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(hist, 0, lane::LANE_SORT_BUCKETS * sizeof(uint32_t), stream);
    if (err != hipSuccess) return err;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(lane_sort_hist, dim3(blocks), dim3(256), 0, stream, sc, n, hist);
    hipLaunchKernelGGL(lane_sort_scan, dim3(1), dim3(1024), 0, stream, hist);
    hipLaunchKernelGGL(lane_sort_scatter, dim3(blocks), dim3(256), 0, stream, sc, n, hist, perm);
    return hipGetLastError();
}
// query = 1: *occ = resident blocks (wavefronts) per CU
hipError_t launch_lane(uint64_t logp_kind, int query, bool tune, const KParams& P, const lane::LaneParams& LP, unsigned grid, hipStream_t stream, int* occ) {
    switch (logp_kind) {
    case NM_LOGP_IID_NORMAL: return launch_lane_d<IidNormal>(query, tune, P, LP, grid, stream, occ);
    case NM_LOGP_DIAG_NORMAL: return launch_lane_d<DiagNormal>(query, tune, P, LP, grid, stream, occ);
    case NM_LOGP_FUNNEL: return launch_lane_d<Funnel>(query, tune, P, LP, grid, stream, occ);
    case NM_LOGP_EIGHT_SCHOOLS: return P.dim == 10 ? launch_lane_t<EightSchools, 5>(query, tune, P, LP, grid, stream, occ) : hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}
}  // namespace nm
