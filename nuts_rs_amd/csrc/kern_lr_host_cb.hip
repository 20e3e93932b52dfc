// kern_lr_host_cb.hip — HostCb with the low-rank transformation (LrWrap<HostCb>)
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_host_cb_lr, LrWrap<HostCb>)
}  // namespace nm
