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
#ifdef NM_MODULE_LANE_DENSITY
#include "nuts_lane.hpp"      // (before the user's header: its lane form is written against nm::lane)
#endif
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

// Optional, for dim <= 16: -DNM_MODULE_LANE_DENSITY=<template name> adds the density's LANE form (one chain per lane, nuts_lane.hpp):
// `template <int NP> struct Name { void init(const double* params, int dim); double eval(const double (&x)[2 NP], double (&gx)[2 NP], int dim) const; }`
// — the whole chain in one lane; sums over dim through nm::lane::pair_tree (the engine's order for <= 16 elements).
#ifdef NM_MODULE_LANE_DENSITY
namespace nm { namespace lane {
template <int NP> struct LaneDensity<NM_MODULE_DENSITY, NP> { using type = NM_MODULE_LANE_DENSITY<NP>; };
template <int NP>
static hipError_t module_lane_t(int query, bool tune, const KParams& P, const LaneParams& LP, unsigned grid, hipStream_t stream, int* occ) {
    if (query == 1) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, nuts_lane_draw_kernel<NM_MODULE_DENSITY, NP, true>, 64, 0);
    if (tune) hipLaunchKernelGGL((nuts_lane_draw_kernel<NM_MODULE_DENSITY, NP, true>), dim3(grid), dim3(64), 0, stream, P, LP);
    else hipLaunchKernelGGL((nuts_lane_draw_kernel<NM_MODULE_DENSITY, NP, false>), dim3(grid), dim3(64), 0, stream, P, LP);
    return hipGetLastError();
}
} }
#define NM_MODULE_HAS_LANE 1
#else
#define NM_MODULE_HAS_LANE 0
#endif

#ifndef NM_MODULE_VARIANTS
#define NM_MODULE_VARIANTS 0          // bit 0: the kernels with the low-rank transformation (LowRankNutsSettings), bit 1: the non-Euclidean
#endif                                // trajectory kinds / MCLMC — the same functor inside LrWrap / KinWrap, as for the built-in densities

extern "C" {
// {sizeof(KParams), NM_ABI_VERSION, DPL, W, lanes per chain of the group form or 0, 1 if built for chains wider than one block,
//  1 if the module carries a lane form, NM_MODULE_VARIANTS}:
// the engine refuses a module built against another layout
void nm_module_info(uint64_t out[8]) {
    out[0] = sizeof(nm::KParams); out[1] = NM_ABI_VERSION; out[2] = NM_MODULE_DPL; out[3] = NM_MODULE_W; out[4] = NM_MODULE_GS_VALUE;
    out[5] = NM_CLUSTER_MODE; out[6] = NM_MODULE_HAS_LANE; out[7] = NM_MODULE_VARIANTS;
}
#if NM_MODULE_VARIANTS
// variant 1: LrWrap<density>, 2: KinWrap<density> (kern_lr_*.hip / kern_kin_*.hip for the built-in ones)
int nm_module_launch_variant(int variant, int kind, const void* kparams, unsigned grid_blocks, void* stream, int* occ) {
    const nm::KParams& P = *static_cast<const nm::KParams*>(kparams);
    hipStream_t s = static_cast<hipStream_t>(stream);
#if NM_MODULE_VARIANTS & 1
    if (variant == 1) return (int)nm::launch_t<NM_MODULE_DPL, NM_MODULE_W, nm::LrWrap<NM_MODULE_DENSITY>>((nm::KernelKind)kind, P, grid_blocks, s, occ);
#endif
#if NM_MODULE_VARIANTS & 2
    if (variant == 2) return (int)nm::launch_t<NM_MODULE_DPL, NM_MODULE_W, nm::KinWrap<NM_MODULE_DENSITY>>((nm::KernelKind)kind, P, grid_blocks, s, occ);
#endif
    return (int)hipErrorInvalidValue;
}
#endif
// kind: nm::KernelKind (init, draw, occupancy query; the group form's draw / warm-up / query)
int nm_module_launch(int kind, const void* kparams, unsigned grid_blocks, void* stream, int* occ) {
    return (int)nm::launch_t<NM_MODULE_DPL, NM_MODULE_W, NM_MODULE_DENSITY>((nm::KernelKind)kind, *static_cast<const nm::KParams*>(kparams),
                                                                            grid_blocks, static_cast<hipStream_t>(stream), occ);
}
#if NM_MODULE_HAS_LANE
// the one-chain-per-lane kernels for this density (query 1: resident wavefronts per CU; 0: launch; tune: the warm-up kernel)
int nm_module_launch_lane(int query, int tune, const void* kparams, const void* lane_params, unsigned grid, void* stream, int* occ) {
    const nm::KParams& P = *static_cast<const nm::KParams*>(kparams);
    const nm::lane::LaneParams& LP = *static_cast<const nm::lane::LaneParams*>(lane_params);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (nm::lane::lane_pairs(P.dim)) {
    case 2: return (int)nm::lane::module_lane_t<2>(query, tune != 0, P, LP, grid, s, occ);
    case 4: return (int)nm::lane::module_lane_t<4>(query, tune != 0, P, LP, grid, s, occ);
    case 5: return (int)nm::lane::module_lane_t<5>(query, tune != 0, P, LP, grid, s, occ);
    }
    return (int)hipErrorInvalidValue;
}
#endif
}
