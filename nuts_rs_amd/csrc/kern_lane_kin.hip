// kern_lane_kin.hip — the one-chain-per-lane draw kernels (nuts_lane.hpp) with the non-Euclidean KineticEnergyKinds compiled in
// (KinWrap<Density>: nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL of NUTS); own TU: parallel build.
// The one-chain-per-lane kernels keep the general-purpose exp / ln / ln_1p of rounds 1-4 (NM_BRANCH_FREE_MATH = 0).  Measured in round 5
// with the branch-free forms the other kernel families use: K4 on 65536 chains 4.90e9 -> 3.88e9 leapfrogs/s (64 unrelated chains per wavefront
// rarely take a special-case branch, and these kernels sit at their register cap: the straight-line forms' constants spill), and ONE parity
// case of the 8-pair kernel (microcanonical, funnel dim 11: profiles/r05h_*) stopped matching the oracle at draw 87 although the two forms
// agree on every operand on the device (tools/probes/sl_math_device_check.hip) — a code-generation-sensitive failure of the kind DESIGN §22
// describes, not root-caused; the form that has passed every suite since round 3 stays.
#define NM_BRANCH_FREE_MATH 0
#include <hip/hip_runtime.h>
#include "nuts_lane.hpp"
namespace nm {
namespace {
template <class Dens, int NP>
hipError_t launch_lane_kin_t(int query, bool tune, const KParams& P, const lane::LaneParams& LP, unsigned grid, hipStream_t stream, int* occ) {
    if constexpr (std::is_void<typename lane::LaneDensity<Dens, NP>::type>::value) {
        return hipErrorInvalidValue;
    } else {
        if (query == 1) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, lane::nuts_lane_draw_kernel<KinWrap<Dens>, NP, true>, 64, 0);
        if (query != 0) return hipErrorInvalidValue;          // (the unsynchronised form stays Euclidean)
        if (tune) hipLaunchKernelGGL((lane::nuts_lane_draw_kernel<KinWrap<Dens>, NP, true>), dim3(grid), dim3(64), 0, stream, P, LP);
        else hipLaunchKernelGGL((lane::nuts_lane_draw_kernel<KinWrap<Dens>, NP, false>), dim3(grid), dim3(64), 0, stream, P, LP);
        return hipGetLastError();
    }
}
template <class Dens>
hipError_t launch_lane_kin_d(int query, bool tune, const KParams& P, const lane::LaneParams& LP, unsigned grid, hipStream_t stream, int* occ) {
    switch (lane::lane_pairs(P.dim)) {
    case 2: return launch_lane_kin_t<Dens, 2>(query, tune, P, LP, grid, stream, occ);
    case 4: return launch_lane_kin_t<Dens, 4>(query, tune, P, LP, grid, stream, occ);
    case 5: return launch_lane_kin_t<Dens, 5>(query, tune, P, LP, grid, stream, occ);
    }
    return hipErrorInvalidValue;
}
}  // namespace
hipError_t launch_lane_kin(uint64_t logp_kind, int query, bool tune, const KParams& P, const lane::LaneParams& LP, unsigned grid, hipStream_t stream, int* occ) {
    switch (logp_kind) {
    case NM_LOGP_IID_NORMAL: return launch_lane_kin_d<IidNormal>(query, tune, P, LP, grid, stream, occ);
    case NM_LOGP_DIAG_NORMAL: return launch_lane_kin_d<DiagNormal>(query, tune, P, LP, grid, stream, occ);
    case NM_LOGP_FUNNEL: return launch_lane_kin_d<Funnel>(query, tune, P, LP, grid, stream, occ);
    case NM_LOGP_EIGHT_SCHOOLS: return P.dim == 10 ? launch_lane_kin_t<EightSchools, 5>(query, tune, P, LP, grid, stream, occ) : hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}
}  // namespace nm
