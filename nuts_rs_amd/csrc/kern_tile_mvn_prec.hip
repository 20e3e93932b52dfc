// kern_tile_mvn_prec.hip — the 16-chains-per-block matrix-core kernel (nuts_tile.hpp) for the full-precision normal with a
// shared low-rank transformation.  Own translation unit: it is compiled in tile mode (tid() = lane).
#include <hip/hip_runtime.h>
#include "nuts_tile.hpp"
namespace nm {
// kind 0: launch; 1: occupancy query (*occ = resident blocks per CU)
hipError_t launch_tile_mvn_prec(int dpl, int query, const KParams& P, const tile::TileMats& M, unsigned grid, hipStream_t stream, int* occ) {
    typedef LrWrap<tile::TileMvnPrec> D;
    if (query) {
        if (dpl == 4) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, tile::nuts_tile_draw_kernel<4, D>, 64 * tile::TC, 0);
        return hipErrorInvalidValue;
    }
    if (dpl == 4) hipLaunchKernelGGL((tile::nuts_tile_draw_kernel<4, D>), dim3(grid), dim3(64 * tile::TC), 0, stream, P, M);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
}  // namespace nm
