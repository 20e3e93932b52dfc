// kern_mvn_prec.hip — nuts_draw_kernel / nuts_init_kernel instantiations for the MvnPrec density (own TU: parallel build)
#include "nuts_launch.hpp"
namespace nm {
NM_DEFINE_LAUNCH(launch_mvn_prec, MvnPrec)
}  // namespace nm
