// kern_tile_mvn_diag.hip — the 16-chains-per-block matrix-core kernel (nuts_tile.hpp) for DiagNutsSettings on the full-precision
// normal: every chain keeps its own adapting diagonal mass matrix, the density's P x is the block's rendezvous product.  Own
// translation unit (tile mode: tid() = lane); apart from kern_tile_mvn_prec.hip so that the two compile side by side.
#include <hip/hip_runtime.h>
#include "nuts_tile.hpp"
namespace nm {
// the same block shape for chains with their own (adapting) diagonal transformation: one product per density evaluation
hipError_t launch_tile_mvn_diag(int dpl, const KParams& P, const tile::TileMats& M, unsigned grid, hipStream_t stream) {
    typedef tile::TileMvnDiag D;
    if (dpl == 4) hipLaunchKernelGGL((tile::nuts_tile_diag_kernel<4, D>), dim3(grid), dim3(64 * tile::TC), 0, stream, P, M);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
}  // namespace nm
