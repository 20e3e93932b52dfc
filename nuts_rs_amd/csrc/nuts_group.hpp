// nuts_group.hpp — the draw kernels for SMALL chains: GS = 8, 16 or 32 lanes per chain, 64 / GS chains per wavefront.
//
// The wave-per-chain kernel (nuts_kernels.hpp) spends ~900 vector instructions per leaf on per-chain scalar work
// (merge arithmetic, reductions, bookkeeping) that 64 lanes execute for ONE chain; at dim 10 five lanes carry data and
// the kernel is VALU-issue bound.  Here a chain with dim <= 2 GS lives in a group of GS lanes (lane l
// holds elements 2l, 2l+1 — the same elements thread l of the wave kernel holds), and the groups of a wave execute the
// same instruction stream with per-lane "scalars".  Everything a group computes is what the wave kernel computes for
// that chain, bit for bit:
//   * sums over dim: the first log2(GS) steps of the wave kernel's butterfly; its further steps add the +0.0 partial
//     sums of the lanes beyond the chain, which changes nothing (partial sums start at +0.0 and are never -0.0);
//   * the same exp / ln, the same ChaCha stream and ziggurat, the same tree (ported from nuts_transition), the same
//     adaptation, the same memory layout (pvec slots per chain; a block's tree scratch holds its chains side by side).
// The doubling loop is naturally lockstep (every group is at depth d in iteration d); groups whose tree ended wait for
// the wave's longest tree.  Scope: warm-up and sampling draws with all their statistics, every built-in density (and user
// modules that bring a group form), maxdepth <= 10; the engine uses the wave kernel otherwise (and for set_position).
// The implementation (nuts_group_impl.hpp) is compiled once per group size into namespaces grp8 / grp16 / grp32.
#pragma once
#include <type_traits>
#include "nuts_kernels.hpp"

namespace nm {
namespace grp {
constexpr int GMAXDEPTH = 10;
#ifndef NM_GROUP_OCC_TUNE
#define NM_GROUP_OCC_TUNE 2  // ... and the warm-up kernel's (1 measured on K4's 8192-chain shard, where one wavefront per SIMD is resident anyway: see DESIGN §8)
#endif
#ifndef NM_GROUP_OCC
#define NM_GROUP_OCC 2       // waves per SIMD the sampling kernel's register allocation leaves room for (3: spills, slower)
#endif
// lanes per chain for a dim (0: no group form)
__host__ __device__ inline int group_size(uint64_t dim) { return dim <= 16 ? 8 : dim <= 32 ? 16 : dim <= 64 ? 32 : 0; }
}  // namespace grp
}  // namespace nm

#define NM_GS 8
#define NM_GNS grp8
#include "nuts_group_impl.hpp"
#undef NM_GS
#undef NM_GNS
#define NM_GS 16
#define NM_GNS grp16
#include "nuts_group_impl.hpp"
#undef NM_GS
#undef NM_GNS
#define NM_GS 32
#define NM_GNS grp32
#include "nuts_group_impl.hpp"
#undef NM_GS
#undef NM_GNS
