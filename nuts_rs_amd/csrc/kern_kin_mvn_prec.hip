// kern_kin_mvn_prec.hip — the MvnPrec kernels with the non-Euclidean KineticEnergyKinds compiled in (KinWrap<MvnPrec>:
// nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_mvn_prec_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<KinWrap<MvnPrec>>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
