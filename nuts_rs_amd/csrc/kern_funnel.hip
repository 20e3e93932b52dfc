// kern_funnel.hip — nuts_draw_kernel / nuts_init_kernel instantiations for the Funnel density (own TU: parallel build)
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_funnel, Funnel)
}  // namespace nm
