// kern_kin_host_cb.hip — the HostCb kernels with the non-Euclidean KineticEnergyKinds compiled in (KinWrap<HostCb>:
// nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_host_cb_kin, KinWrap<HostCb>)
}  // namespace nm
