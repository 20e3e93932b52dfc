// kern_iid_normal.hip — nuts_draw_kernel / nuts_init_kernel instantiations for the IidNormal density (own TU: parallel build)
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_iid_normal, IidNormal)
}  // namespace nm
