// kern_lr_diag_normal.hip — the DiagNormal kernels with the low-rank transformation (LrWrap<DiagNormal>); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_diag_normal_lr(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<LrWrap<DiagNormal>>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
