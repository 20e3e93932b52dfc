// density_module.hip — translation unit of a USER density module (include/nuts_amd.h, "User densities").
// Compiled by the user (or nuts_rs_amd.build.build_density_module) with
//   -DNM_MODULE_DENSITY=<struct name> -DNM_MODULE_HEADER='"<header that defines it>"' -DNM_MODULE_DPL=<d> -DNM_MODULE_W=<w>
// into a shared object that the engine dlopen()s for nm_logp_spec.kind == NM_LOGP_MODULE.
// For dim > 4096 (several blocks per chain): -DNM_CLUSTER_MODE=1 with the (16, 4) tiling; the density then brings
// `init_slice(params, dim, gdim, goff, reducer)` — its block holds elements [goff, goff + dim) of a chain of gdim — and every
// reducer sum spans the whole chain.
// Optional, for dim <= 64: -DNM_MODULE_GROUP_DENSITY=<template name> -DNM_MODULE_GS=<8|16|32> adds the density's group form
// (several chains per wavefront, nuts_group.hpp): `template <class L> struct Name` written against L = Lanes.
#include "nuts_launch.hpp"
#include NM_MODULE_HEADER

#ifndef NM_MODULE_DPL
#error "define NM_MODULE_DPL (doubles per lane: 2, 4, 8 or 16)"
#endif
#ifndef NM_MODULE_W
#error "define NM_MODULE_W (waves per chain: 1, 2 or 4)"
#endif

#ifdef NM_MODULE_GROUP_DENSITY
#ifndef NM_MODULE_GS
#error "define NM_MODULE_GS (lanes per chain: 8 for dim <= 16, 16 for dim <= 32, 32 for dim <= 64)"
#endif
#define NM_MODULE_CAT2(a, b) a##b
#define NM_MODULE_CAT(a, b) NM_MODULE_CAT2(a, b)
namespace nm { namespace NM_MODULE_CAT(grp, NM_MODULE_GS) {
template <> struct GroupDensity<NM_MODULE_DENSITY> { using type = NM_MODULE_GROUP_DENSITY<Lanes>; };
} }
#define NM_MODULE_GS_VALUE NM_MODULE_GS
#else
#define NM_MODULE_GS_VALUE 0
#endif

extern "C" {
// {sizeof(KParams), NM_ABI_VERSION, DPL, W, lanes per chain of the group form or 0, 1 if built for chains wider than one block}:
// the engine refuses a module built against another layout
void nm_module_info(uint64_t out[6]) {
    out[0] = sizeof(nm::KParams); out[1] = NM_ABI_VERSION; out[2] = NM_MODULE_DPL; out[3] = NM_MODULE_W; out[4] = NM_MODULE_GS_VALUE;
    out[5] = NM_CLUSTER_MODE;
}
// kind: nm::KernelKind (init, draw, occupancy query; the group form's draw / warm-up / query)
int nm_module_launch(int kind, const void* kparams, unsigned grid_blocks, void* stream, int* occ) {
    return (int)nm::launch_t<NM_MODULE_DPL, NM_MODULE_W, NM_MODULE_DENSITY>((nm::KernelKind)kind, *static_cast<const nm::KParams*>(kparams),
                                                                            grid_blocks, static_cast<hipStream_t>(stream), occ);
}
}
