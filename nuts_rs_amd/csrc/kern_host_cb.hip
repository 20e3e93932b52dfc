// kern_host_cb.hip — the kernels of the host-callback density (NM_LOGP_HOST_CALLBACK, struct HostCb): diagonal and low-rank forms
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_host_cb, HostCb)
}  // namespace nm
