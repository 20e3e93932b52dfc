// kern_cluster_kin.hip — the wide-chain kernels (kern_cluster.hip: NM_CLUSTER_MODE, ceil(dim / 4096) blocks per chain) with the
// non-Euclidean KineticEnergyKinds and the MCLMC sampler compiled in (KinWrap<Density>): every sum of the geodesic / ESH
// leapfrogs and of the partial momentum refresh goes through the same exchange.  Own TU: parallel build.
#define NM_CLUSTER_MODE 1
#include "nuts_launch.hpp"
namespace nm {
template <class Dens>
static hipError_t launch_cluster_kin_t(KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    if (kind == K_QUERY) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, nuts_draw_kernel<16, 4, Dens>, 256, 0);
    if (kind == K_INIT) hipLaunchKernelGGL((nuts_init_kernel<16, 4, Dens>), dim3(grid), dim3(256), 0, stream, P);
    else if (kind == K_DRAW) hipLaunchKernelGGL((nuts_draw_kernel<16, 4, Dens>), dim3(grid), dim3(256), 0, stream, P);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
struct ClkIidNormal : IidNormal {};
struct ClkDiagNormal : DiagNormal {};
struct ClkHostCb : HostCb {};
hipError_t launch_cluster_kin(uint64_t logp_kind, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    switch (logp_kind) {
    case NM_LOGP_IID_NORMAL: return launch_cluster_kin_t<KinWrap<ClkIidNormal>>(kind, P, grid, stream, occ);
    case NM_LOGP_DIAG_NORMAL: return launch_cluster_kin_t<KinWrap<ClkDiagNormal>>(kind, P, grid, stream, occ);
    case NM_LOGP_HOST_CALLBACK: return launch_cluster_kin_t<KinWrap<ClkHostCb>>(kind, P, grid, stream, occ);
    }
    return hipErrorInvalidValue;
}
}  // namespace nm
