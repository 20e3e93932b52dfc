// kern_lane_kin.hip — the one-chain-per-lane draw kernels (nuts_lane.hpp) with the non-Euclidean KineticEnergyKinds compiled in
// (KinWrap<Density>: nm_settings.trajectory_kind = NM_TRAJ_EXACT_NORMAL / NM_TRAJ_MICROCANONICAL of NUTS); own TU: parallel build.
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
    case 8: return launch_lane_kin_t<Dens, 8>(query, tune, P, LP, grid, stream, occ);
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
