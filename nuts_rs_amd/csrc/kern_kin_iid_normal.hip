// kern_kin_iid_normal.hip — the IidNormal kernels with the non-Euclidean KineticEnergyKinds compiled in (KinWrap<IidNormal>:
// nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_iid_normal_kin, KinWrap<IidNormal>)
}  // namespace nm
