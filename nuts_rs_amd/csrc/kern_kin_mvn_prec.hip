// kern_kin_mvn_prec.hip — the MvnPrec kernels with the non-Euclidean KineticEnergyKinds compiled in (KinWrap<MvnPrec>:
// nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_mvn_prec_kin, KinWrap<MvnPrec>)
}  // namespace nm
