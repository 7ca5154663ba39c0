// Register-resident rollout kernels (sx_rollout_rw.hpp, sx_rollout_rh.hpp) for state dimension 4: every n_pad / 16 that fits.
#include "sx_rw_impl.hpp"

namespace sx {
template int launch_rollout_rw<4, 1>(const GpConst<4, 5>&, const ReachConst<4, 1>&,
                                      const CostConst<SX_MAX_M, 4, 1>&, const RolloutPtrs&, hipStream_t);
template int launch_rollout_rh<4, 1>(const GpConst<4, 5>&, const ReachConst<4, 1>&,
                                      const CostConst<SX_MAX_M, 4, 1>&, const RolloutPtrs&, hipStream_t);
template int launch_rollout_rw<4, 2>(const GpConst<4, 6>&, const ReachConst<4, 2>&,
                                      const CostConst<SX_MAX_M, 4, 2>&, const RolloutPtrs&, hipStream_t);
template int launch_rollout_rh<4, 2>(const GpConst<4, 6>&, const ReachConst<4, 2>&,
                                      const CostConst<SX_MAX_M, 4, 2>&, const RolloutPtrs&, hipStream_t);
template bool rollout_rh_applies<4, 1>(int, int, int);
template bool rollout_rw_applies<4, 1>(int, int, int);
template bool rollout_rh_applies<4, 2>(int, int, int);
template bool rollout_rw_applies<4, 2>(int, int, int);
}  // namespace sx
