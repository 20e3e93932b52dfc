// kern_eight_schools.hip — nuts_draw_kernel / nuts_init_kernel instantiations for the EightSchools density (own TU: parallel build)
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_eight_schools(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return (dpl == 2 && w == 1) ? launch_t<2, 1, EightSchools>(kind, P, grid, stream, occ) : hipErrorInvalidValue;
}
}  // namespace nm
