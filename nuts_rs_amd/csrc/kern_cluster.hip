// kern_cluster.hip — chains wider than one block (4096 < dim <= 131072): the same draw and init kernels compiled in
// NM_CLUSTER_MODE, where ceil(dim / 4096) co-resident blocks of 4 wavefronts each own a 4096-element slice of ONE chain and
// exchange every block sum through the chain's mailbox (dev_math.hpp "chains wider than one block").  Element-wise densities
// (IidNormal, DiagNormal) and host-callback densities (any density), the (16 doubles, 4 waves) tiling only.  Own TU: the mode is a macro, like NM_TILE_MODE.
#define NM_CLUSTER_MODE 1
#include "nuts_launch.hpp"
namespace nm {
template <class Dens>
static hipError_t launch_cluster_t(KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    if (kind == K_QUERY) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, nuts_draw_kernel<16, 4, Dens>, 256, 0);
    if (kind == K_INIT) hipLaunchKernelGGL((nuts_init_kernel<16, 4, Dens>), dim3(grid), dim3(256), 0, stream, P);
    else if (kind == K_DRAW) hipLaunchKernelGGL((nuts_draw_kernel<16, 4, Dens>), dim3(grid), dim3(256), 0, stream, P);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
// the Cl* names keep these instantiations apart from the single-block ones of the other translation units
struct ClIidNormal : IidNormal {};
struct ClDiagNormal : DiagNormal {};
struct ClHostCb : HostCb {};          // any host density: the members share the chain's mailbox (nuts_kernels.hpp HostCb)
hipError_t launch_cluster(uint64_t logp_kind, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    switch (logp_kind) {
    case NM_LOGP_IID_NORMAL: return launch_cluster_t<ClIidNormal>(kind, P, grid, stream, occ);
    case NM_LOGP_DIAG_NORMAL: return launch_cluster_t<ClDiagNormal>(kind, P, grid, stream, occ);
    case NM_LOGP_HOST_CALLBACK: return launch_cluster_t<ClHostCb>(kind, P, grid, stream, occ);
    }
    return hipErrorInvalidValue;
}
}  // namespace nm
