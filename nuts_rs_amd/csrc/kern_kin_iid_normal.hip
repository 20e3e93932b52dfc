// kern_kin_iid_normal.hip — the IidNormal kernels with the non-Euclidean KineticEnergyKinds compiled in (KinWrap<IidNormal>:
// nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_iid_normal_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<KinWrap<IidNormal>>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
