// kern_lr_iid_normal.hip — the IidNormal kernels with the low-rank transformation (LrWrap<IidNormal>); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_iid_normal_lr(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<LrWrap<IidNormal>>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
