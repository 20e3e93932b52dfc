// kern_lr_mvn_prec.hip — the MvnPrec kernels with the low-rank transformation (LrWrap<MvnPrec>); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_mvn_prec_lr(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<LrWrap<MvnPrec>>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
