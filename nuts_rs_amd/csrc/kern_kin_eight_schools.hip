// kern_kin_eight_schools.hip — the EightSchools kernels with the non-Euclidean KineticEnergyKinds compiled in
// (KinWrap<EightSchools>); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_eight_schools_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return (dpl == 2 && w == 1) ? launch_t<2, 1, KinWrap<EightSchools>>(kind, P, grid, stream, occ) : hipErrorInvalidValue;
}
}  // namespace nm
