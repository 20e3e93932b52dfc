// kern_lr_diag_normal.hip — the DiagNormal kernels with the low-rank transformation (LrWrap<DiagNormal>); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_diag_normal_lr, LrWrap<DiagNormal>)
}  // namespace nm
