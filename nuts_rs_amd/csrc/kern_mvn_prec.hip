// kern_mvn_prec.hip — nuts_draw_kernel / nuts_init_kernel instantiations for the MvnPrec density (own TU: parallel build)
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_mvn_prec(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<MvnPrec>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
