// kern_host_cb.hip — the kernels of the host-callback density (NM_LOGP_HOST_CALLBACK, struct HostCb): diagonal and low-rank forms
#include "nuts_launch.hpp"
namespace nm {
hipError_t launch_host_cb(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    return launch_d<HostCb>(dpl, w, kind, P, grid, stream, occ);
}
}  // namespace nm
