// kern_diag_normal.hip — nuts_draw_kernel / nuts_init_kernel instantiations for the DiagNormal density (own TU: parallel build)
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_diag_normal, DiagNormal)
}  // namespace nm
