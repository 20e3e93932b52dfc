// kern_lr_mvn_prec.hip — the MvnPrec kernels with the low-rank transformation (LrWrap<MvnPrec>); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_mvn_prec_lr, LrWrap<MvnPrec>)
}  // namespace nm
