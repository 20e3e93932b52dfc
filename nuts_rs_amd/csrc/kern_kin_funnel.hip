// kern_kin_funnel.hip — the Funnel kernels with the non-Euclidean KineticEnergyKinds compiled in (KinWrap<Funnel>:
// nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_funnel_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<KinWrap<Funnel>>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
