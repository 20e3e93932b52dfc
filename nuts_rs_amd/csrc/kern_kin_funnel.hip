// kern_kin_funnel.hip — the Funnel kernels with the non-Euclidean KineticEnergyKinds compiled in (KinWrap<Funnel>:
// nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_funnel_kin, KinWrap<Funnel>)
}  // namespace nm
