// kern_kin_diag_normal.hip — the DiagNormal kernels with the non-Euclidean KineticEnergyKinds compiled in (KinWrap<DiagNormal>:
// nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_diag_normal_kin, KinWrap<DiagNormal>)
}  // namespace nm
