// Register-resident rollout kernels (sx_rollout_rw.hpp, sx_rollout_rh.hpp) for state dimension 2: every n_pad / 16 that fits.
#include "sx_rw_impl.hpp"

namespace sx {
template int launch_rollout_rw<2, 1>(const GpConst<2, 3>&, const ReachConst<2, 1>&,
                                      const CostConst<SX_MAX_M, 2, 1>&, const RolloutPtrs&, hipStream_t);
template int launch_rollout_rh<2, 1>(const GpConst<2, 3>&, const ReachConst<2, 1>&,
                                      const CostConst<SX_MAX_M, 2, 1>&, const RolloutPtrs&, hipStream_t);
template int launch_rollout_rw<2, 2>(const GpConst<2, 4>&, const ReachConst<2, 2>&,
                                      const CostConst<SX_MAX_M, 2, 2>&, const RolloutPtrs&, hipStream_t);
template int launch_rollout_rh<2, 2>(const GpConst<2, 4>&, const ReachConst<2, 2>&,
                                      const CostConst<SX_MAX_M, 2, 2>&, const RolloutPtrs&, hipStream_t);
template bool rollout_rh_applies<2, 1>(int, int, int);
template bool rollout_rw_applies<2, 1>(int, int, int);
template bool rollout_rh_applies<2, 2>(int, int, int);
template bool rollout_rw_applies<2, 2>(int, int, int);
}  // namespace sx
