// MC-dropout state-space models (reference ssm_cem/dropout_ssm_cem.py, gal_concrete_dropout.py) as an ENSEMBLE of
// S thinned networks: a fully connected ReLU network whose dropout masks are drawn once per (re)training and then held fixed,
// so that prediction is a deterministic function -- mean and unbiased variance over the S members, mean Jacobian by a
// reverse sweep per member and output:
//     a_0 = m_0^s * z        a_l = relu(W_l a_{l-1} + b_l) * m_l^s   (l = 1 .. L)        out^s = W_{L+1} a_L + b_{L+1}
//     mean = 1/S sum_s out^s      var = 1/(S-1) sum_s (out^s - mean)^2      jac = 1/S sum_s d out^s / dz
// (the reference draws its masks with torch's RNG -- bnn's fixed eval masks, or fresh concrete-dropout noise on every forward
// pass in gal_concrete_dropout.py:61-75 -- and differentiates with autograd; bnn is absent: values parity-unpinned, and
// freezing the concrete noise is a documented deviation)
//
// THIS FILE: one particle per LANE, members in sequence: the weights and masks are wave-uniform (scalar loads), the member's
// hidden pre-activations sit in LDS as [unit][lane].  It serves the shapes the matrix-core kernels of sx_mlp_mfma.hpp do not
// (no hidden layer, three or four hidden layers) and is their A/B reference (SX_MLP_PATH=valu); the reference's default
// network (64 x 64, 30 members; ~0.8 MFLOP per particle-step) runs 350x faster there.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sx_amd.h"
#include "sx_reach.hpp"

namespace sx {

constexpr int kMlpLanes = 64;

struct MlpConst {
    int n_hidden, n_samples, d_in, n_out, wmax, predict_std;
    int width[SX_MLP_MAX_HIDDEN + 1];   // width[0] = D, width[l] = hidden layer l
    const double* net;                  // W_1 [w1 x D], b_1, ..., W_L, b_L, W_out [n_out x w_L], b_out
    const double* masks;                // [S][sum of width[0 .. L]]: m_0 (input), m_1, ..., m_L
};

// LDS doubles per workgroup: pre-activations of every hidden layer + two gradient vectors, [unit][lane]
inline size_t mlp_lds_doubles(int n_hidden, int wmax) { return (size_t)(n_hidden + 2) * wmax * kMlpLanes; }

// mean, unbiased variance over the members and (WITH_JAC) the mean Jacobian of this lane's point
template <int NS, int D, bool WITH_JAC>
__device__ __forceinline__ void mlp_ensemble_predict(const MlpConst& mc, const double (&z)[D], double* lds, int lane,
                                                     double (&mean)[NS], double (&var)[NS], double (&jac)[NS][D]) {
    const int L = mc.n_hidden, W = mc.wmax;
    int msum = 0;
    for (int l = 0; l <= L; ++l) msum += mc.width[l];
    double m2[NS], alea[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        mean[d] = 0.0;
        m2[d] = 0.0;
        alea[d] = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) jac[d][j] = 0.0;
    }
    for (int s = 0; s < mc.n_samples; ++s) {
        const double* mk = mc.masks + (size_t)s * msum;
        // ---- forward ----
        const double* w = mc.net;
        const double* mk_in = mk;                      // mask of this layer's INPUT units
        for (int l = 1; l <= L; ++l) {
            const int win = mc.width[l - 1], wout = mc.width[l];
            double* pre = lds + (size_t)(l - 1) * W * kMlpLanes;
            const double* prev = lds + (size_t)(l - 2) * W * kMlpLanes;
            const double* bias = w + (size_t)wout * win;
            // four output units at a time: one LDS read of the input unit feeds four independent fma chains (a dependent
            // f64 fma issues every ~9 cycles, an independent one every ~4: tools/valu_probe.hip)
            for (int k0 = 0; k0 < wout; k0 += 4) {
                double acc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = (k0 + u < wout) ? bias[k0 + u] : 0.0;
                if (l == 1) {
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        const double m = mk_in[i] * z[i];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (k0 + u < wout) acc[u] = fma(w[(k0 + u) * D + i], m, acc[u]);
                    }
                } else {
                    for (int i = 0; i < win; ++i) {
                        const double a = prev[i * kMlpLanes + lane];
                        const double m = mk_in[i] * (a > 0.0 ? a : 0.0);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (k0 + u < wout) acc[u] = fma(w[(k0 + u) * win + i], m, acc[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k0 + u < wout) pre[(k0 + u) * kMlpLanes + lane] = acc[u];
            }
            w = bias + wout;
            mk_in += win;
        }
        // output layer (mk_in now points at m_L)
        const int wl = mc.width[L];
        const double* w_out = w;
        const double* b_out = w + (size_t)mc.n_out * wl;
        const double* last = lds + (size_t)(L - 1) * W * kMlpLanes;
        double out[NS];
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            double acc = b_out[d];
            if (L == 0) {
#pragma unroll
                for (int i = 0; i < D; ++i) acc = fma(w_out[d * D + i] * mk_in[i], z[i], acc);
            } else {
                for (int i = 0; i < wl; ++i) {
                    const double a = last[i * kMlpLanes + lane];
                    acc = fma(w_out[d * wl + i] * mk_in[i], a > 0.0 ? a : 0.0, acc);
                }
            }
            out[d] = acc;
            // Welford over the members
            const double delta = acc - mean[d];
            mean[d] += delta / (double)(s + 1);
            m2[d] = fma(delta, acc - mean[d], m2[d]);
            if (mc.predict_std) {   // log standard deviation of output d: row n_s + d of the output layer
                double ls = b_out[NS + d];
                if (L == 0) {
#pragma unroll
                    for (int i = 0; i < D; ++i) ls = fma(w_out[(NS + d) * D + i] * mk_in[i], z[i], ls);
                } else {
                    for (int i = 0; i < wl; ++i) {
                        const double a = last[i * kMlpLanes + lane];
                        ls = fma(w_out[(NS + d) * wl + i] * mk_in[i], a > 0.0 ? a : 0.0, ls);
                    }
                }
                alea[d] += exp(2.0 * ls);
            }
        }
        // ---- reverse sweep per output ----
        if constexpr (WITH_JAC) {
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                if (L == 0) {
#pragma unroll
                    for (int i = 0; i < D; ++i) jac[d][i] += w_out[d * D + i] * mk_in[i];
                    continue;
                }
                double* g = lds + (size_t)L * W * kMlpLanes;          // gradient w.r.t. the current layer's pre-activations
                double* gn = lds + (size_t)(L + 1) * W * kMlpLanes;
                for (int i = 0; i < wl; ++i)
                    g[i * kMlpLanes + lane] = (last[i * kMlpLanes + lane] > 0.0) ? w_out[d * wl + i] * mk_in[i] : 0.0;
                // layers L .. 1: weights of layer l start at woff(l), its input mask at moff(l - 1)
                for (int l = L; l >= 1; --l) {
                    int woff = 0, moff = 0;
                    for (int q = 1; q < l; ++q) {
                        woff += mc.width[q] * mc.width[q - 1] + mc.width[q];
                        moff += mc.width[q - 1];
                    }
                    const int win = mc.width[l - 1], wout = mc.width[l];
                    const double* wl_ = mc.net + woff;
                    const double* mkl = mk + moff;
                    if (l == 1) {
#pragma unroll
                        for (int i = 0; i < D; ++i) {
                            double acc = 0.0;
                            for (int k = 0; k < wout; ++k) acc = fma(wl_[k * D + i], g[k * kMlpLanes + lane], acc);
                            jac[d][i] += acc * mkl[i];
                        }
                    } else {
                        const double* pre_in = lds + (size_t)(l - 2) * W * kMlpLanes;
                        for (int i0 = 0; i0 < win; i0 += 4) {
                            double acc[4] = {0.0, 0.0, 0.0, 0.0};
                            for (int k = 0; k < wout; ++k) {
                                const double gk = g[k * kMlpLanes + lane];
#pragma unroll
                                for (int u = 0; u < 4; ++u)
                                    if (i0 + u < win) acc[u] = fma(wl_[k * win + i0 + u], gk, acc[u]);
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (i0 + u < win)
                                    gn[(i0 + u) * kMlpLanes + lane] =
                                        (pre_in[(i0 + u) * kMlpLanes + lane] > 0.0) ? acc[u] * mkl[i0 + u] : 0.0;
                        }
                        double* t = g;
                        g = gn;
                        gn = t;
                    }
                }
            }
        }
    }
    const double inv_s = 1.0 / (double)mc.n_samples;
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        var[d] = (mc.n_samples > 1 ? m2[d] / (double)(mc.n_samples - 1) : 0.0) + alea[d] * inv_s;
        if constexpr (WITH_JAC) {
#pragma unroll
            for (int j = 0; j < D; ++j) jac[d][j] *= inv_s;
        }
    }
}

inline MlpConst make_mlp_const(const sx_mlp_model* m) {
    MlpConst mc;
    mc.n_hidden = m->n_hidden;
    mc.n_samples = m->n_samples;
    mc.d_in = m->n_s + m->n_u;
    mc.n_out = m->n_out;
    mc.predict_std = m->predict_std;
    mc.wmax = 1;
    for (int l = 0; l <= SX_MLP_MAX_HIDDEN; ++l) {
        mc.width[l] = m->width[l];
        if (l >= 1 && l <= m->n_hidden && m->width[l] > mc.wmax) mc.wmax = m->width[l];
    }
    mc.net = m->net;
    mc.masks = m->masks;
    return mc;
}

template <int NS, int NU>
__global__ __launch_bounds__(kMlpLanes) void mlp_predict_kernel(MlpConst mc, const double* __restrict__ zin, int P,
                                                                double* __restrict__ mean, double* __restrict__ var,
                                                                double* __restrict__ jac) {
    constexpr int D = NS + NU;
    extern __shared__ __attribute__((aligned(16))) double mlp_smem[];
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x * (int64_t)kMlpLanes + lane;
    double z[D], m[NS], v[NS], jc[NS][D];
#pragma unroll
    for (int j = 0; j < D; ++j) z[j] = (g < P) ? zin[g * D + j] : 0.0;
    if (jac)
        mlp_ensemble_predict<NS, D, true>(mc, z, mlp_smem, lane, m, v, jc);
    else
        mlp_ensemble_predict<NS, D, false>(mc, z, mlp_smem, lane, m, v, jc);
    if (g >= P) return;
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        mean[g * NS + d] = m[d];
        var[g * NS + d] = v[d];
        if (jac) {
#pragma unroll
            for (int j = 0; j < D; ++j) jac[(g * NS + d) * D + j] = jc[d][j];
        }
    }
}

// the CEM particle rollout over the ensemble: one particle per lane for all H steps (arguments as FeatRolloutPtrs)
template <int NS, int NU>
__global__ __launch_bounds__(kMlpLanes) void cem_rollout_mlp_kernel(MlpConst mc, ReachConst<NS, NU> rc,
                                                                    CostConst<SX_MAX_M, NS, NU> cc, FeatRolloutPtrs rp) {
    constexpr int D = NS + NU;
    constexpr int S = NS + NS * NS;
    extern __shared__ __attribute__((aligned(16))) double mlp_smem[];
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x * (int64_t)kMlpLanes + lane;
    const int64_t total = (int64_t)rp.E * rp.P;
    const bool valid = g < total;
    const int64_t gg = valid ? g : 0;
    const int e = (int)(gg / rp.P);
    const int H = rp.H;
    double p[NS], Q[NS][NS];
    bool have_q = rp.q0 != nullptr;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        p[i] = rp.x0[(int64_t)e * NS + i];
#pragma unroll
        for (int j = 0; j < NS; ++j) Q[i][j] = have_q ? rp.q0[((int64_t)e * NS + i) * NS + j] : 0.0;
    }
    double obj = 0.0, con = 0.0;
    int st = 0;
    for (int t = 0; t < H; ++t) {
        double z[D], u[NU], mean[NS], var[NS], jac[NS][D], p1[NS], Q1[NS][NS];
#pragma unroll
        for (int c = 0; c < NU; ++c) {
            const int64_t gi = (gg * H + t) * NU + c;
            double a;
            if (rp.noise) {
                a = rp.mean[((int64_t)e * H + t) * NU + c] + rp.std[((int64_t)e * H + t) * NU + c] * rp.noise[gi];
                if (valid) rp.actions[gi] = a;
            } else {
                a = rp.actions[gi];
            }
            u[c] = a;
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) z[j] = p[j];
#pragma unroll
        for (int c = 0; c < NU; ++c) z[NS + c] = u[c];
        if (have_q) {
            mlp_ensemble_predict<NS, D, true>(mc, z, mlp_smem, lane, mean, var, jac);
            reach_ellipsoid<NS, NU>(rc, p, Q, u, mean, var, jac, p1, Q1, st);
        } else {
            mlp_ensemble_predict<NS, D, false>(mc, z, mlp_smem, lane, mean, var, jac);
            reach_point<NS, NU>(rc, p, u, mean, var, p1, Q1, st);
        }
        have_q = true;
        obj += objective_cost<SX_MAX_M, NS, NU>(cc, p1, var);
        bool uviol = false;
#pragma unroll
        for (int c = 0; c < NU; ++c) uviol = uviol || (u[c] < cc.u_min[c]) || (u[c] > cc.u_max[c]);
        if (uviol) con += SX_ACTION_VIOLATION_COST;
        if (cc.con_mode == SX_CON_ALL_STATES || t == H - 1) {
            if (polytope_violated<SX_MAX_M, NS>(cc.h_mat, cc.h_vec, cc.m, 1.0, p1, Q1, nullptr)) con += SX_STATE_VIOLATION_COST;
        }
        if (valid && rp.traj) {
            double* tr = rp.traj + (g * H + t) * S;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                tr[i] = p1[i];
#pragma unroll
                for (int j = 0; j < NS; ++j) tr[NS + i * NS + j] = Q1[i][j];
            }
        }
        if (valid && rp.sigma) {
#pragma unroll
            for (int i = 0; i < NS; ++i) rp.sigma[(g * H + t) * NS + i] = var[i];
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            p[i] = p1[i];
#pragma unroll
            for (int j = 0; j < NS; ++j) Q[i][j] = Q1[i][j];
        }
    }
    if (valid) {
        rp.obj_cost[g] = obj;
        rp.con_cost[g] = con;
        if (st) atomicOr(rp.status, st);
    }
}

}  // namespace sx
