// Register-resident rollout kernels (sx_rollout_rw.hpp, sx_rollout_rh.hpp) for state dimension 3: every n_pad / 16 that fits.
#include "sx_rw_impl.hpp"

namespace sx {
template int launch_rollout_rw<3, 1>(const GpConst<3, 4>&, const ReachConst<3, 1>&,
                                      const CostConst<SX_MAX_M, 3, 1>&, const RolloutPtrs&, hipStream_t);
template int launch_rollout_rh<3, 1>(const GpConst<3, 4>&, const ReachConst<3, 1>&,
                                      const CostConst<SX_MAX_M, 3, 1>&, const RolloutPtrs&, hipStream_t);
template bool rollout_rh_applies<3, 1>(int, int, int);
template bool rollout_rw_applies<3, 1>(int, int, int);
}  // namespace sx
