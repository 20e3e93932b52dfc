// kern_kin_diag_normal.hip — the DiagNormal kernels with the non-Euclidean KineticEnergyKinds compiled in (KinWrap<DiagNormal>:
// nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_diag_normal_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<KinWrap<DiagNormal>>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
