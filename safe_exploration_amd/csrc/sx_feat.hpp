// Exact GPs with a DEGENERATE kernel  k_d(x, x') = c_d phi(x) . phi(x')  -- the reference's 'linear' kernel (phi = identity)
// and 'nn' kernel (phi = a small fully connected network, min/max-normalised per point), ssm_cem/gp_ssm_cem.py:45-57,140-185.
//
// MI355X-first form: such a GP is Bayesian linear regression on F features, so nothing about it scales with N_train at
// prediction time.  With  A_d = Phi^T Phi + (noise_d / c_d) I  (F x F),  M_d = chol(A_d)^-1,  wbar_d = A_d^-1 Phi^T y_d:
//     mean_d(z) = wbar_d . phi(z)          var_d(z) = noise_d (|M_d phi(z)|^2 + 1)      (likelihood noise included)
//     d mean_d / dz = (d phi / dz)^T wbar_d                                          (reverse sweep through the network)
// -- identical to the kernel-space posterior (Woodbury), which is how the oracle computes it.  A particle-step is a few
// thousand flops on F <= 32 features: one particle per LANE for the whole rollout, weights and M_d read through the scalar
// cache (their indices are wave-uniform), per-lane activations in LDS ([unit][lane]: conflict-free).  No matrix cores: there
// is no N x N operand to contract.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sx_amd.h"
#include "sx_reach.hpp"

namespace sx {

constexpr int kFeatWave = 64;                            // one wave per workgroup, one particle per lane
// LDS doubles per workgroup: pre-activations of every layer + features + two scratch vectors of the reverse sweep, [unit][lane]
constexpr int kFeatLdsDoubles = (SX_FEAT_MAX_LAYERS + 3) * SX_FEAT_MAX_WIDTH * kFeatWave;

struct FeatConst {
    int n_layers, n_feat, normalise, d_in;
    int width[SX_FEAT_MAX_LAYERS + 1];
    double prelu;
    double noise[SX_MAX_NS];
    const double* net;
    const double* wbar;
    const double* minv;
};

// phi(z) for this lane's point.  pre[l] (LDS, [unit][lane]) keeps layer l's pre-activations for the reverse sweep;
// `feat` receives phi.  Returns argmin / argmax / min / max of the un-normalised features through the references.
template <int D>
__device__ __forceinline__ void feat_forward(const FeatConst& fc, const double (&z)[D], double* lds, int lane, int& amin,
                                             int& amax, double& fmin, double& fmax) {
    double* feat = lds + (SX_FEAT_MAX_LAYERS) * SX_FEAT_MAX_WIDTH * kFeatWave;
    const int F = fc.n_feat;
    if (fc.n_layers == 0) {   // linear kernel: phi = z
#pragma unroll
        for (int j = 0; j < D; ++j) feat[j * kFeatWave + lane] = z[j];
        amin = amax = 0;
        fmin = fmax = 0.0;
        return;
    }
    const double* w = fc.net;
    for (int l = 1; l <= fc.n_layers; ++l) {
        const int win = fc.width[l - 1], wout = fc.width[l];
        double* out = lds + (l - 1) * SX_FEAT_MAX_WIDTH * kFeatWave;
        const double* in = lds + (l - 2) * SX_FEAT_MAX_WIDTH * kFeatWave;   // (l == 1 reads z instead)
        const double* bias = w + (size_t)wout * win;
        for (int k = 0; k < wout; ++k) {
            double s = bias[k];
            if (l == 1) {
#pragma unroll
                for (int i = 0; i < D; ++i) s = fma(w[k * D + i], z[i], s);
            } else {
                for (int i = 0; i < win; ++i) {
                    const double a = in[i * kFeatWave + lane];
                    s = fma(w[k * win + i], a > 0.0 ? a : 0.0, s);           // ReLU between the layers
                }
            }
            out[k * kFeatWave + lane] = s;
        }
        w = bias + wout;
    }
    // PReLU after the last layer, then the reference's per-point normalisation (gp_ssm_cem.py:176-181):
    //   phi = 2 (f - min f) / max f - 1        (max of the UN-shifted features, as the reference writes it)
    const double* last = lds + (fc.n_layers - 1) * SX_FEAT_MAX_WIDTH * kFeatWave;
    fmin = 0.0;
    fmax = 0.0;
    amin = amax = 0;
    for (int k = 0; k < F; ++k) {
        const double p = last[k * kFeatWave + lane];
        const double f = p > 0.0 ? p : fc.prelu * p;
        feat[k * kFeatWave + lane] = f;
        if (k == 0 || f < fmin) { fmin = f; amin = k; }
        if (k == 0 || f > fmax) { fmax = f; amax = k; }
    }
    if (fc.normalise) {
        for (int k = 0; k < F; ++k) feat[k * kFeatWave + lane] = 2.0 * ((feat[k * kFeatWave + lane] - fmin) / fmax) - 1.0;
    }
}

// mean, variance (noise included) and, with WITH_JAC, the mean Jacobian [NS][D] of this lane's point.
template <int NS, int D, bool WITH_JAC>
__device__ __forceinline__ void feat_gp_predict(const FeatConst& fc, const double (&z)[D], double* lds, int lane,
                                                double (&mean)[NS], double (&var)[NS], double (&jac)[NS][D]) {
    int amin, amax;
    double fmin, fmax;
    feat_forward<D>(fc, z, lds, lane, amin, amax, fmin, fmax);
    const int F = fc.n_feat;
    const double* feat = lds + (SX_FEAT_MAX_LAYERS) * SX_FEAT_MAX_WIDTH * kFeatWave;
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        double* grad = lds + (SX_FEAT_MAX_LAYERS + 1) * SX_FEAT_MAX_WIDTH * kFeatWave;   // scratch of the reverse sweep
        const double* wb = fc.wbar + (size_t)d * F;
        const double* M = fc.minv + (size_t)d * F * F;
        double m = 0.0, q = 0.0;
        for (int r = 0; r < F; ++r) {
            const double ph = feat[r * kFeatWave + lane];
            m = fma(wb[r], ph, m);
            double t = 0.0;
            for (int k = 0; k <= r; ++k) t = fma(M[r * F + k], feat[k * kFeatWave + lane], t);
            q = fma(t, t, q);
        }
        mean[d] = m;
        var[d] = fc.noise[d] * q + fc.noise[d];
        if constexpr (WITH_JAC) {
            if (fc.n_layers == 0) {
#pragma unroll
                for (int j = 0; j < D; ++j) jac[d][j] = wb[j];
                continue;
            }
            // reverse sweep: g_phi = wbar_d -> normalisation -> PReLU -> layers
            double gsum = 0.0, gdot = 0.0;
            const double* last = lds + (fc.n_layers - 1) * SX_FEAT_MAX_WIDTH * kFeatWave;
            if (fc.normalise) {
                for (int k = 0; k < F; ++k) {
                    gsum += wb[k];
                    // f_k - min f recovered from phi: (phi + 1) max f / 2
                    gdot = fma(wb[k], (feat[k * kFeatWave + lane] + 1.0) * 0.5 * fmax, gdot);
                }
            }
            for (int k = 0; k < F; ++k) {
                double g = wb[k];
                if (fc.normalise) {
                    g = 2.0 / fmax * g;
                    if (k == amin) g -= 2.0 / fmax * gsum;
                    if (k == amax) g -= 2.0 / (fmax * fmax) * gdot;
                }
                const double p = last[k * kFeatWave + lane];
                grad[k * kFeatWave + lane] = g * (p > 0.0 ? 1.0 : fc.prelu);
            }
            // layers L .. 1; the weights of layer l start at offset off[l]
            int off[SX_FEAT_MAX_LAYERS + 1];
            off[1] = 0;
            for (int l = 1; l < fc.n_layers; ++l) off[l + 1] = off[l] + fc.width[l] * fc.width[l - 1] + fc.width[l];
            for (int l = fc.n_layers; l >= 1; --l) {
                const int win = fc.width[l - 1], wout = fc.width[l];
                const double* w = fc.net + off[l];
                if (l == 1) {
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        double s = 0.0;
                        for (int k = 0; k < wout; ++k) s = fma(w[k * D + i], grad[k * kFeatWave + lane], s);
                        jac[d][i] = s;
                    }
                } else {
                    const double* pre = lds + (l - 2) * SX_FEAT_MAX_WIDTH * kFeatWave;
                    // (every input unit reads ALL of the layer's output gradients: the new gradient goes to a second
                    // scratch vector and the two swap roles)
                    double* next = lds + (SX_FEAT_MAX_LAYERS + 1 + ((fc.n_layers - l) & 1 ? 0 : 1)) * SX_FEAT_MAX_WIDTH * kFeatWave;
                    for (int i = 0; i < win; ++i) {
                        double s = 0.0;
                        for (int k = 0; k < wout; ++k) s = fma(w[k * win + i], grad[k * kFeatWave + lane], s);
                        next[i * kFeatWave + lane] = (pre[i * kFeatWave + lane] > 0.0) ? s : 0.0;
                    }
                    grad = next;
                }
            }
        }
    }
}

inline FeatConst make_feat_const(const sx_feat_model* m) {
    FeatConst fc;
    fc.n_layers = m->n_layers;
    fc.n_feat = m->n_feat;
    fc.normalise = m->normalise;
    fc.d_in = m->n_s + m->n_u;
    for (int l = 0; l <= SX_FEAT_MAX_LAYERS; ++l) fc.width[l] = m->width[l];
    fc.prelu = m->prelu;
    for (int d = 0; d < SX_MAX_NS; ++d) fc.noise[d] = m->noise[d];
    fc.net = m->net;
    fc.wbar = m->wbar;
    fc.minv = m->minv;
    return fc;
}

// ---- sx_feat_features: Phi = phi(X) for N points ---------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kFeatWave) void feat_features_kernel(FeatConst fc, const double* __restrict__ x, int n,
                                                                  double* __restrict__ phi) {
    extern __shared__ __attribute__((aligned(16))) double feat_smem[];
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x * (int64_t)kFeatWave + lane;
    double z[D];
#pragma unroll
    for (int j = 0; j < D; ++j) z[j] = (g < n) ? x[g * D + j] : 0.0;
    int amin, amax;
    double fmin, fmax;
    feat_forward<D>(fc, z, feat_smem, lane, amin, amax, fmin, fmax);
    const double* feat = feat_smem + (SX_FEAT_MAX_LAYERS) * SX_FEAT_MAX_WIDTH * kFeatWave;
    if (g < n)
        for (int k = 0; k < fc.n_feat; ++k) phi[g * fc.n_feat + k] = feat[k * kFeatWave + lane];
}

// ---- sx_feat_predict ---------------------------------------------------------------------------------------------------
template <int NS, int NU>
__global__ __launch_bounds__(kFeatWave) void feat_predict_kernel(FeatConst fc, const double* __restrict__ zin, int P,
                                                                 double* __restrict__ mean, double* __restrict__ var,
                                                                 double* __restrict__ jac) {
    constexpr int D = NS + NU;
    extern __shared__ __attribute__((aligned(16))) double feat_smem[];
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x * (int64_t)kFeatWave + lane;
    double z[D], m[NS], v[NS], jc[NS][D];
#pragma unroll
    for (int j = 0; j < D; ++j) z[j] = (g < P) ? zin[g * D + j] : 0.0;
    if (jac)
        feat_gp_predict<NS, D, true>(fc, z, feat_smem, lane, m, v, jc);
    else
        feat_gp_predict<NS, D, false>(fc, z, feat_smem, lane, m, v, jc);
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

// ---- sx_feat_fit: A_d = Phi^T Phi + lambda_d I, M_d = chol(A_d)^-1, wbar_d = M_d^T M_d Phi^T y_d ------------------------
// One workgroup of 1024 threads per output; F <= 32, so A (F x F) has one thread per entry and lives in LDS.
// stats[d] = { y_d^T y_d, |M_d Phi^T y_d|^2, sum log diag chol(A_d) }: what the exact marginal likelihood needs.
struct FeatFitArgs {
    const double* phi;   // [N x F]
    const double* y;     // [N x n_s]
    double lambda[SX_MAX_NS];
    double* wbar;        // [n_s x F]
    double* minv;        // [n_s x F x F]
    double* stats;       // [n_s x 3]
    int* status;
    int n, F, n_s;
};

__global__ __launch_bounds__(1024) void feat_fit_kernel(FeatFitArgs a) {
    __shared__ double A[SX_FEAT_MAX_WIDTH][SX_FEAT_MAX_WIDTH + 1];
    __shared__ double Li[SX_FEAT_MAX_WIDTH][SX_FEAT_MAX_WIDTH + 1];
    __shared__ double b[SX_FEAT_MAX_WIDTH], t[SX_FEAT_MAX_WIDTH];
    __shared__ double yy_part[16];
    const int d = blockIdx.x, tid = threadIdx.x;
    const int F = a.F, n = a.n;
    const int r = tid / SX_FEAT_MAX_WIDTH, c = tid % SX_FEAT_MAX_WIDTH;
    if (r < F && c <= r) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s = fma(a.phi[(size_t)i * F + r], a.phi[(size_t)i * F + c], s);
        if (r == c) s += a.lambda[d];
        A[r][c] = s;
        A[c][r] = s;
    }
    if (tid < F) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s = fma(a.phi[(size_t)i * F + tid], a.y[(size_t)i * a.n_s + d], s);
        b[tid] = s;
    }
    {
        double s = 0.0;
        for (int i = tid; i < n; i += 1024) {
            const double v = a.y[(size_t)i * a.n_s + d];
            s = fma(v, v, s);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if ((tid & 63) == 0) yy_part[tid >> 6] = s;
    }
    __syncthreads();
    // Cholesky A = L L^T in place (lower), column by column; F <= 32: one wave's worth of rows
    bool bad = false;
    for (int j = 0; j < F; ++j) {
        if (tid == 0) {
            const double p = A[j][j];
            if (!(p > 0.0)) bad = true;
            A[j][j] = sqrt(p);
        }
        __syncthreads();
        if (tid > j && tid < F) A[tid][j] /= A[j][j];
        __syncthreads();
        if (r > j && r < F && c > j && c <= r) A[r][c] -= A[r][j] * A[c][j];
        __syncthreads();
    }
    // M = L^-1 by forward substitution, one column per thread
    if (tid < F) {
        const int col = tid;
        for (int i = 0; i < F; ++i) {
            double s = (i == col) ? 1.0 : 0.0;
            for (int k = col; k < i; ++k) s -= A[i][k] * Li[k][col];
            Li[i][col] = (i >= col) ? s / A[i][i] : 0.0;
        }
    }
    __syncthreads();
    if (tid < F) {   // t = M b
        double s = 0.0;
        for (int k = 0; k <= tid; ++k) s = fma(Li[tid][k], b[k], s);
        t[tid] = s;
    }
    __syncthreads();
    if (tid < F) {   // wbar = M^T t
        double s = 0.0;
        for (int k = tid; k < F; ++k) s = fma(Li[k][tid], t[k], s);
        a.wbar[(size_t)d * F + tid] = s;
    }
    if (r < F && c < F) a.minv[((size_t)d * F + r) * F + c] = (c <= r) ? Li[r][c] : 0.0;
    if (tid == 0) {
        double yy = 0.0, tt = 0.0, ld = 0.0;
        for (int w = 0; w < 16; ++w) yy += yy_part[w];
        for (int k = 0; k < F; ++k) {
            tt = fma(t[k], t[k], tt);
            ld += log(A[k][k]);
        }
        a.stats[d * 3 + 0] = yy;
        a.stats[d * 3 + 1] = tt;
        a.stats[d * 3 + 2] = ld;
        if (bad) atomicOr(a.status, SX_STATUS_NOT_PD);
    }
}

// ---- sx_cem_rollout_feat: the CEM particle rollout over a feature-space GP, one particle per lane for all H steps ----------
struct FeatRolloutPtrs {
    const double* x0;
    const double* q0;
    const double* mean;
    const double* std;
    const double* noise;
    double* actions;
    double* traj;
    double* sigma;
    double* obj_cost;
    double* con_cost;
    int* status;
    int E, P, H;
};

template <int NS, int NU>
__global__ __launch_bounds__(kFeatWave) void cem_rollout_feat_kernel(FeatConst fc, ReachConst<NS, NU> rc,
                                                                     CostConst<SX_MAX_M, NS, NU> cc, FeatRolloutPtrs rp) {
    constexpr int D = NS + NU;
    constexpr int S = NS + NS * NS;
    extern __shared__ __attribute__((aligned(16))) double feat_smem[];
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x * (int64_t)kFeatWave + lane;
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
            feat_gp_predict<NS, D, true>(fc, z, feat_smem, lane, mean, var, jac);
            reach_ellipsoid<NS, NU>(rc, p, Q, u, mean, var, jac, p1, Q1, st);
        } else {
            feat_gp_predict<NS, D, false>(fc, z, feat_smem, lane, mean, var, jac);
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
