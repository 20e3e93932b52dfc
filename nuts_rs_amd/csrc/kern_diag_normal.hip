// kern_diag_normal.hip — nuts_draw_kernel / nuts_init_kernel instantiations for the DiagNormal density (own TU: parallel build)
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_diag_normal(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<DiagNormal>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
