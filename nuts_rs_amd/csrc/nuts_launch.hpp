// nuts_launch.hpp — kernel launch plumbing shared by the per-density translation units (each density's kernels
// are compiled in their own .hip file so the build parallelises).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "nuts_kernels.hpp"
#include "nuts_group.hpp"

// A density's kernels are compiled as TWO translation units from one source (nuts_rs_amd/build.py, units "x.hip@small" / "x.hip@large"):
//   NM_TU_PART 1: the tilings of <= 4 doubles per lane and the small-chain kernels, special functions inlined (NM_DETMATH_INLINE = 1); it also
//                 holds the entry point, which hands the larger tilings to part 2;
//   NM_TU_PART 2: the tilings of 8 and 16 doubles per lane, special functions out of line;
//   NM_TU_PART 0: everything in one unit (tools/build_unit_variant.sh, user modules: one tiling each).
#ifndef NM_TU_PART
#define NM_TU_PART 0
#endif
namespace nm {
enum KernelKind { K_INIT, K_DRAW, K_QUERY,      // K_QUERY: resident blocks per CU of the draw kernel
                  K_GROUP_DRAW, K_GROUP_TUNE, K_GROUP_QUERY,    // the small-chain kernels of nuts_group.hpp (dim <= 64): sampling / warm-up
                  K_GROUP_DRAW_ROOMY, K_GROUP_TUNE_ROOMY };     // ... compiled for one wavefront per SIMD (grids of at most 4 x CUs blocks; built-in densities)

// the small-chain kernels exist for the densities that have a group form (nuts_group.hpp); the group size follows P.dim
#define NM_LAUNCH_GROUP_NS(NS)                                                                                            \
    if constexpr (!std::is_void<typename NS::GroupDensity<Dens>::type>::value) {                                          \
        if (kind == K_GROUP_QUERY) {   /* both kernels share the block's tree scratch: the grid is sized for the roomier one */ \
            int a = 0, b = 0;                                                                                             \
            hipError_t st = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, NS::nuts_group_draw_kernel<Dens, false>, 64, 0); \
            if (st != hipSuccess) return st;                                                                              \
            st = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, NS::nuts_group_draw_kernel<Dens, true>, 64, 0);         \
            *occ = a > b ? a : b;                                                                                         \
            return st;                                                                                                    \
        }                                                                                                                 \
        if (kind == K_GROUP_TUNE_ROOMY) { hipLaunchKernelGGL((NS::nuts_group_draw_kernel<Dens, true, true>), dim3(grid_blocks), dim3(64), 0, stream, P); return hipGetLastError(); } \
        if (kind == K_GROUP_DRAW_ROOMY) { hipLaunchKernelGGL((NS::nuts_group_draw_kernel<Dens, false, true>), dim3(grid_blocks), dim3(64), 0, stream, P); return hipGetLastError(); } \
        if (kind == K_GROUP_TUNE) hipLaunchKernelGGL((NS::nuts_group_draw_kernel<Dens, true>), dim3(grid_blocks), dim3(64), 0, stream, P); \
        else hipLaunchKernelGGL((NS::nuts_group_draw_kernel<Dens, false>), dim3(grid_blocks), dim3(64), 0, stream, P);    \
        return hipGetLastError();                                                                                         \
    } else {                                                                                                              \
        return hipErrorInvalidValue;                                                                                      \
    }
template <class Dens>
inline hipError_t launch_group(KernelKind kind, const KParams& P, unsigned grid_blocks, hipStream_t stream, int* occ) {
    switch (grp::group_size(P.dim)) {
    case 8: NM_LAUNCH_GROUP_NS(grp8)
    case 16: NM_LAUNCH_GROUP_NS(grp16)
    case 32: NM_LAUNCH_GROUP_NS(grp32)
    }
    return hipErrorInvalidValue;
}
#undef NM_LAUNCH_GROUP_NS
// grid = number of blocks (one block of 64*W threads = one resident chain); for K_QUERY *occ receives
// hipOccupancyMaxActiveBlocksPerMultiprocessor of the draw kernel
inline bool is_group_kind(KernelKind kind) { return kind == K_GROUP_DRAW || kind == K_GROUP_TUNE || kind == K_GROUP_QUERY || kind == K_GROUP_DRAW_ROOMY || kind == K_GROUP_TUNE_ROOMY; }
template <int DPL, int W, class Dens>
inline hipError_t launch_t(KernelKind kind, const KParams& P, unsigned grid_blocks, hipStream_t stream, int* occ) {
#if NM_TU_PART != 2
    if (is_group_kind(kind)) return launch_group<Dens>(kind, P, grid_blocks, stream, occ);
#else
    if (is_group_kind(kind)) return hipErrorInvalidValue;      // (the small-chain kernels live in part 1)
#endif
    if (kind == K_QUERY) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, nuts_draw_kernel<DPL, W, Dens>, 64 * W, 0);
    dim3 grid(grid_blocks), block(64 * W);
    if (kind == K_INIT) hipLaunchKernelGGL((nuts_init_kernel<DPL, W, Dens>), grid, block, 0, stream, P);
    else hipLaunchKernelGGL((nuts_draw_kernel<DPL, W, Dens>), grid, block, 0, stream, P);
    return hipGetLastError();
}
// supported tilings: W = 1: DPL 2,4,8,16 (dim <= 1024); W = 2: DPL 8,16 (dim <= 2048); W = 4: DPL 4,16 (dim <= 4096)
template <class Dens, int PART = NM_TU_PART>          // (PART: the two units' instantiations are different functions)
inline hipError_t launch_d(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ) {
    switch (w * 100 + dpl) {
#ifdef NM_DEV_ONLY_K2
    case 116: return launch_t<16, 1, Dens>(kind, P, grid, stream, occ);
    default: return hipErrorInvalidValue;
#elif defined(NM_DEV_ONLY_41)      // (bisecting builds: one tiling per unit compiles in a minute)
    case 104: return launch_t<4, 1, Dens>(kind, P, grid, stream, occ);
    default: return hipErrorInvalidValue;
#else
#if NM_TU_PART != 2
    case 102: return launch_t<2, 1, Dens>(kind, P, grid, stream, occ);
    case 104: return launch_t<4, 1, Dens>(kind, P, grid, stream, occ);
    case 404: return launch_t<4, 4, Dens>(kind, P, grid, stream, occ);
#endif
#if NM_TU_PART != 1
    case 108: return launch_t<8, 1, Dens>(kind, P, grid, stream, occ);
    case 116: return launch_t<16, 1, Dens>(kind, P, grid, stream, occ);
    case 208: return launch_t<8, 2, Dens>(kind, P, grid, stream, occ);
    case 216: return launch_t<16, 2, Dens>(kind, P, grid, stream, occ);
    case 416:   // 138 KiB of LDS already: no room for a density that keeps a block-visible vector there
        if constexpr (!Dens::kNeedsLdsVector) return launch_t<16, 4, Dens>(kind, P, grid, stream, occ);
        else return hipErrorInvalidValue;
#endif
#endif
    }
    return hipErrorInvalidValue;
}

// the entry point of a density's kernels, in kern_<density>.hip: NM_DEFINE_LAUNCH(launch_x, Density)
#define NM_LAUNCH_ARGS int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ
#if NM_TU_PART == 1
#define NM_DEFINE_LAUNCH(name, ...)                                                                                       \
    hipError_t name##_large(NM_LAUNCH_ARGS);                                                                              \
    hipError_t name(NM_LAUNCH_ARGS) {                                                                                     \
        if (dpl <= 4 || is_group_kind(kind)) return launch_d<__VA_ARGS__>(dpl, w, kind, P, grid, stream, occ);            \
        return name##_large(dpl, w, kind, P, grid, stream, occ);                                                          \
    }
#elif NM_TU_PART == 2
#define NM_DEFINE_LAUNCH(name, ...) hipError_t name##_large(NM_LAUNCH_ARGS) { return launch_d<__VA_ARGS__>(dpl, w, kind, P, grid, stream, occ); }
#else
#define NM_DEFINE_LAUNCH(name, ...) hipError_t name(NM_LAUNCH_ARGS) { return launch_d<__VA_ARGS__>(dpl, w, kind, P, grid, stream, occ); }
#endif
// one definition per density, in kern_<density>.hip
hipError_t launch_iid_normal(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_diag_normal(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_funnel(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_eight_schools(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_mvn_prec(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
// the same kernels with the low-rank transformation compiled in (LrWrap<Density>), in kern_lr_<density>.hip
hipError_t launch_iid_normal_lr(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_diag_normal_lr(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_funnel_lr(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_eight_schools_lr(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_mvn_prec_lr(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
// the same kernels with the non-Euclidean KineticEnergyKinds compiled in (KinWrap<Density>), in kern_kin_<density>.hip
hipError_t launch_iid_normal_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_diag_normal_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_funnel_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_eight_schools_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_mvn_prec_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_host_cb_kin(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
// chains wider than one block (kern_cluster.hip): dim > 4096, element-wise densities
hipError_t launch_cluster(uint64_t logp_kind, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
// ... with the non-Euclidean trajectory kinds / MCLMC (kern_cluster_kin.hip)
hipError_t launch_cluster_kin(uint64_t logp_kind, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
// NM_LOGP_HOST_CALLBACK (kern_host_cb.hip)
hipError_t launch_host_cb(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
hipError_t launch_host_cb_lr(int dpl, int w, KernelKind kind, const KParams& P, unsigned grid, hipStream_t stream, int* occ);
}  // namespace nm
