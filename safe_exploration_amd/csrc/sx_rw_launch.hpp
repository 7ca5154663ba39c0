// Launcher of the register-resident rollout kernels (sx_rollout_rw.hpp).  Its instantiations are compiled in translation
// units of their own (sx_rw_ns{1..4}.hip: one per state dimension, built in parallel); sx_kernels.hip sees the declaration.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sx_amd.h"
#include "sx_gp.hpp"
#include "sx_reach.hpp"
#include "sx_rollout.hpp"

namespace sx {

constexpr int kRwWaves = 4;   // one wave per SIMD, 512 registers each

// largest n_pad / 16 whose W fits the register budget next to the step's working set, i.e. that compiles without a
// scratch spill (checked over all instantiations with tools/kernel_resources.py after every change of the kernel)
constexpr int rw_max_nrb(int ns, int nu) {
    return ns == 1 ? 18 : ns == 2 ? (nu == 1 ? 13 : 12) : ns == 3 ? 8 : (nu == 1 ? 6 : 5);
}

// Launches cem_rollout_rw_kernel<NS, NU, n_pad / 16> on `stream`; SX_ERR_UNSUPPORTED when the model is too large for the
// register-resident form (the caller then takes cem_rollout_kernel).
template <int NS, int NU>
int launch_rollout_rw(const GpConst<NS, NS + NU>& gc, const ReachConst<NS, NU>& rc, const CostConst<SX_MAX_M, NS, NU>& cc,
                      const RolloutPtrs& rp, hipStream_t stream);

// cem_rollout_rh_kernel (eight waves of 256 registers): instantiated where finish() and the Kstar phase leave room for the
// resident pairs without a scratch spill -- n_s <= 2; at n_s >= 3 their working sets (Jacobi sweeps on 3 x 3 / 4 x 4
// matrices, 2 n_s exponential chains) do not (tools/kernel_resources.py: (2, 2, 10) and every n_s >= 3 instantiation spill)
constexpr int rh_max_nrb(int ns, int nu) { return ns == 1 ? 18 : ns == 2 ? (nu == 1 ? 13 : 8) : 0; }

// The same for cem_rollout_rh_kernel (sx_rollout_rh.hpp: eight waves, W partly resident).
template <int NS, int NU>
int launch_rollout_rh(const GpConst<NS, NS + NU>& gc, const ReachConst<NS, NU>& rc, const CostConst<SX_MAX_M, NS, NU>& cc,
                      const RolloutPtrs& rp, hipStream_t stream);

// Would launch_rollout_rh / launch_rollout_rw take this model (an instantiation exists and its LDS fits)?  No launch.
template <int NS, int NU>
bool rollout_rh_applies(int n_train, int n_pad, int H);
template <int NS, int NU>
bool rollout_rw_applies(int n_train, int n_pad, int H);

}  // namespace sx
