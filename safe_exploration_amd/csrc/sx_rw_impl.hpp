// Body of launch_rollout_rw<NS, NU>; included by sx_rw_ns*.hip, which instantiate it.
#pragma once
#include "sx_launch.hpp"
#include "sx_rollout_rh.hpp"
#include "sx_rollout_rw.hpp"
#include "sx_rw_launch.hpp"

namespace sx {

template <int NS, int NU, int NRB>
static int rw_try(int nrb, const GpConst<NS, NS + NU>& gc, const ReachConst<NS, NU>& rc,
                  const CostConst<SX_MAX_M, NS, NU>& cc, const RolloutPtrs& rp, hipStream_t stream) {
    if constexpr (NRB == 0) {
        return SX_ERR_UNSUPPORTED;
    } else {
        if (nrb != NRB) return rw_try<NS, NU, NRB - 1>(nrb, gc, rc, cc, rp, stream);
        static_assert(rw_fits<NS, NRB>(), "rw_max_nrb promises more than the register budget holds");
        const size_t lds = (gp_tile_lds_doubles(NS, NS + NU, gc.n_train, gc.n_pad, kRwWaves, NS) +
                            (((size_t)SX_TILE * rp.H * NU + 1) & ~(size_t)1) + (((size_t)SX_TILE * (NS + NS * NS + 3) + 1) & ~(size_t)1) + RwKstarLds<NS, NS + NU>::doubles(gc.n_pad)) *
                               sizeof(double) +
                           sizeof(RwConst<NS, NU>);
        if (int r = allow_lds(cem_rollout_rw_kernel<NS, NU, NRB>, lds)) return r;
        const int tiles = rp.E * ((rp.P + SX_TILE - 1) / SX_TILE);
        // one workgroup (4 waves x 512 registers) fills a compute unit: a persistent grid, W loaded once per workgroup
        const int grid = tiles < device_cus() ? tiles : device_cus();
        launch(SX_PROF_ROLLOUT_FUSED, cem_rollout_rw_kernel<NS, NU, NRB>, dim3(grid), dim3(kRwThreads), lds, stream, gc, rc, cc,
               rp);
        return check_launch();
    }
}

template <int NS, int NU>
int launch_rollout_rw(const GpConst<NS, NS + NU>& gc, const ReachConst<NS, NU>& rc, const CostConst<SX_MAX_M, NS, NU>& cc,
                      const RolloutPtrs& rp, hipStream_t stream) {
    const int nrb = gc.n_pad >> 4;
    if (nrb < 1 || nrb > rw_max_nrb(NS, NU)) return SX_ERR_UNSUPPORTED;
#if SX_RW_DIET
    // rw_kstar_phase's table is 2^(j/2048): the exponent constants in units of ln 2 / 2048 (a factor of 8: exact)
    GpConst<NS, NS + NU> g8 = gc;
    for (int i = 0; i < NS * (NS + NU); ++i) g8.k_nh_ils2[i] *= 8.0;
    for (int d = 0; d < NS; ++d) g8.k_log_os[d] *= 8.0;
    return rw_try<NS, NU, rw_max_nrb(NS, NU)>(nrb, g8, rc, cc, rp, stream);
#else
    return rw_try<NS, NU, rw_max_nrb(NS, NU)>(nrb, gc, rc, cc, rp, stream);
#endif
}

template <int NS, int NU, int NRB>
static int rh_try(int nrb, const GpConst<NS, NS + NU>& gc, const ReachConst<NS, NU>& rc,
                  const CostConst<SX_MAX_M, NS, NU>& cc, const RolloutPtrs& rp, hipStream_t stream) {
    if constexpr (NRB == 0) {
        return SX_ERR_UNSUPPORTED;
    } else {
        if (nrb != NRB) return rh_try<NS, NU, NRB - 1>(nrb, gc, rc, cc, rp, stream);
        const size_t lds = rh_lds_doubles<NS, NU, NRB>(gc.n_train, gc.n_pad, rp.H) * sizeof(double) + sizeof(RwConst<NS, NU>);
        if (int r = allow_lds(cem_rollout_rh_kernel<NS, NU, NRB>, lds)) return r;
        const int tiles = rp.E * ((rp.P + SX_TILE - 1) / SX_TILE);
        const int grid = tiles < device_cus() ? tiles : device_cus();
        launch(SX_PROF_ROLLOUT_FUSED, cem_rollout_rh_kernel<NS, NU, NRB>, dim3(grid), dim3(kRhThreads), lds, stream, gc, rc, cc,
               rp);
        return check_launch();
    }
}

template <int NS, int NU>
int launch_rollout_rh(const GpConst<NS, NS + NU>& gc, const ReachConst<NS, NU>& rc, const CostConst<SX_MAX_M, NS, NU>& cc,
                      const RolloutPtrs& rp, hipStream_t stream) {
    const int nrb = gc.n_pad >> 4;
    if (nrb < 1 || nrb > rh_max_nrb(NS, NU)) return SX_ERR_UNSUPPORTED;
    // rw_kstar_phase's table is 2^(j/2048): the exponent constants in units of ln 2 / 2048 (a factor of 8: exact)
    GpConst<NS, NS + NU> g8 = gc;
    for (int i = 0; i < NS * (NS + NU); ++i) g8.k_nh_ils2[i] *= 8.0;
    for (int d = 0; d < NS; ++d) g8.k_log_os[d] *= 8.0;
    return rh_try<NS, NU, rh_max_nrb(NS, NU)>(nrb, g8, rc, cc, rp, stream);
}

template <int NS, int NU, int NRB>
static size_t rh_lds_bytes_of(int nrb, int n_train, int n_pad, int H) {
    if constexpr (NRB == 0) {
        return ~(size_t)0;
    } else {
        if (nrb != NRB) return rh_lds_bytes_of<NS, NU, NRB - 1>(nrb, n_train, n_pad, H);
        return rh_lds_doubles<NS, NU, NRB>(n_train, n_pad, H) * sizeof(double) + sizeof(RwConst<NS, NU>);
    }
}

template <int NS, int NU>
bool rollout_rh_applies(int n_train, int n_pad, int H) {
    const int nrb = n_pad >> 4;
    if (nrb < 1 || nrb > rh_max_nrb(NS, NU)) return false;
    return rh_lds_bytes_of<NS, NU, rh_max_nrb(NS, NU)>(nrb, n_train, n_pad, H) <= kMaxLdsBytes;
}

template <int NS, int NU>
bool rollout_rw_applies(int n_train, int n_pad, int H) {
    const int nrb = n_pad >> 4;
    if (nrb < 1 || nrb > rw_max_nrb(NS, NU)) return false;
    const size_t lds = (gp_tile_lds_doubles(NS, NS + NU, n_train, n_pad, kRwWaves, NS) + (((size_t)SX_TILE * H * NU + 1) & ~(size_t)1) +
                        (((size_t)SX_TILE * (NS + NS * NS + 3) + 1) & ~(size_t)1) + RwKstarLds<NS, NS + NU>::doubles(n_pad)) *
                           sizeof(double) + sizeof(RwConst<NS, NU>);
    return lds <= kMaxLdsBytes;
}

}  // namespace sx
