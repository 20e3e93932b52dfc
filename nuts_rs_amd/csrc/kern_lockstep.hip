// kern_lockstep.hip — the lockstep matrix-core kernel (nuts_lockstep.hpp): 16 chains per block advance together, one density
// evaluation per round; the full-precision normal through a shared, frozen low-rank transformation (BASELINE config 5).
// Own translation unit, compiled in tile mode (tid() = lane).
#include <hip/hip_runtime.h>
#include "nuts_lockstep.hpp"
namespace nm {
// kind 0: the draw launch (the LR_SET_TRANSFORM commit first, on the same stream); 1: resident blocks per CU
hipError_t launch_lockstep(int query, const KParams& P, const tile::TileMats& M, unsigned grid, hipStream_t stream, int* occ) {
    if (query) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, lock::nuts_lockstep_kernel, 64 * lock::LWV, 0);
    hipLaunchKernelGGL(lock::lock_commit_kernel, dim3((unsigned)P.n_chains), dim3(64), 0, stream, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lock::nuts_lockstep_kernel, dim3(grid), dim3(64 * lock::LWV), 0, stream, P, M);
    return hipGetLastError();
}
}  // namespace nm
