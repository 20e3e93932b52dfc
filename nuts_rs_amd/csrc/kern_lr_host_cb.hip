// kern_lr_host_cb.hip — HostCb with the low-rank transformation (LrWrap<HostCb>)
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_host_cb_lr(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<LrWrap<HostCb>>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
