// kern_lr_funnel.hip — the Funnel kernels with the low-rank transformation (LrWrap<Funnel>); own TU: parallel build
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_funnel_lr, LrWrap<Funnel>)
}  // namespace nm
