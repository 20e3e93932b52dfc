// density_module.hip — translation unit of a USER density module (include/nuts_amd.h, "User densities").
// Compiled by the user (or nuts_rs_amd.build.build_density_module) with
//   -DNM_MODULE_DENSITY=<struct name> -DNM_MODULE_HEADER='"<header that defines it>"' -DNM_MODULE_DPL=<d> -DNM_MODULE_W=<w>
// into a shared object that the engine dlopen()s for nm_logp_spec.kind == NM_LOGP_MODULE.
#include "nuts_launch.hpp"
#include NM_MODULE_HEADER

#ifndef NM_MODULE_DPL
#error "define NM_MODULE_DPL (doubles per lane: 2, 4, 8 or 16)"
#endif
#ifndef NM_MODULE_W
#error "define NM_MODULE_W (waves per chain: 1, 2 or 4)"
#endif

extern "C" {
// {sizeof(KParams), NM_ABI_VERSION, DPL, W}: the engine refuses a module built against another layout
void nm_module_info(uint64_t out[4]) {
    out[0] = sizeof(nm::KParams); out[1] = NM_ABI_VERSION; out[2] = NM_MODULE_DPL; out[3] = NM_MODULE_W;
}
// kind: 0 init kernel, 1 draw kernel, 2 occupancy query (nm::KernelKind)
int nm_module_launch(int kind, const void* kparams, unsigned grid_blocks, void* stream, int* occ) {
    return (int)nm::launch_t<NM_MODULE_DPL, NM_MODULE_W, NM_MODULE_DENSITY>((nm::KernelKind)kind, *static_cast<const nm::KParams*>(kparams),
                                                                            grid_blocks, static_cast<hipStream_t>(stream), occ);
}
}
