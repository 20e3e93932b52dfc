// kern_lr_funnel.hip — the Funnel kernels with the low-rank transformation (LrWrap<Funnel>); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_funnel_lr(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<LrWrap<Funnel>>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
