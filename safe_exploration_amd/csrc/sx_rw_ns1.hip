// Register-resident rollout kernels (sx_rollout_rw.hpp, sx_rollout_rh.hpp) for state dimension 1: every n_pad / 16 that fits.
#include "sx_rw_impl.hpp"

namespace sx {
template int launch_rollout_rw<1, 1>(const GpConst<1, 2>&, const ReachConst<1, 1>&,
                                      const CostConst<SX_MAX_M, 1, 1>&, const RolloutPtrs&, hipStream_t);
template int launch_rollout_rh<1, 1>(const GpConst<1, 2>&, const ReachConst<1, 1>&,
                                      const CostConst<SX_MAX_M, 1, 1>&, const RolloutPtrs&, hipStream_t);
template bool rollout_rh_applies<1, 1>(int, int, int);
template bool rollout_rw_applies<1, 1>(int, int, int);
}  // namespace sx
