// kern_lr_iid_normal.hip — the IidNormal kernels with the low-rank transformation (LrWrap<IidNormal>); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_iid_normal_lr, LrWrap<IidNormal>)
}  // namespace nm
