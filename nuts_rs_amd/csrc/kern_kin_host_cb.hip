// kern_kin_host_cb.hip — the HostCb kernels with the non-Euclidean KineticEnergyKinds compiled in (KinWrap<HostCb>:
// nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_host_cb_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<KinWrap<HostCb>>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
