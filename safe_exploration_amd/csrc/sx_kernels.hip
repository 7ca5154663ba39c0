// libsxamd: kernels + C ABI (include/sx_amd.h).  gfx950 only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/sx_amd.h"
#include "sx_gp.hpp"
#include "sx_reach.hpp"
#include "sx_big.hpp"

namespace sx {

constexpr size_t kMaxLdsBytes = 160 * 1024;

#ifndef SX_ROLLOUT_THREADS
#define SX_ROLLOUT_THREADS (64 * SX_WAVES)
#endif
constexpr int kRolloutThreads = SX_ROLLOUT_THREADS;  // waves of the CU that owns the 16-particle tile
constexpr int kPredictThreads = 64 * SX_WAVES;

// ---------------------------------------------------------------------------------------------------------------
// sx_gp_pack: W_d / alpha_d -> fragment order
// ---------------------------------------------------------------------------------------------------------------
template <int MAXNS, int MAXD>
struct PackArgs {
    double inv_ls2[MAXNS * MAXD];
};

// rows < N: W_d (lower triangular);  rows N .. N + D: alpha_d, alpha_d * X_j / l_dj^2;  above: zero
__global__ void pack_a_kernel(const double* __restrict__ linv, const double* __restrict__ alpha,
                              const double* __restrict__ x_train, PackArgs<SX_MAX_NS, SX_MAX_D> args, int n_s, int D, int n,
                              int n_pad, double* __restrict__ a_pack) {
    const int nrb = n_pad >> 4;
    const int64_t wpo = w_pairs_per_output(nrb);
    const int64_t total = (int64_t)n_s * wpo * 128;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(i & 1);
        const int lane = (int)((i >> 1) & 63);
        int64_t pair = i >> 7;
        const int d = (int)(pair / wpo);
        pair -= (int64_t)d * wpo;
        // row-block rb owns pairs [rb (rb + 1), (rb + 1)(rb + 2))
        int rb = (int)((sqrt(4.0 * (double)pair + 1.0) - 1.0) * 0.5);
        while ((int64_t)rb * (rb + 1) > pair) --rb;
        while ((int64_t)(rb + 1) * (rb + 2) <= pair) ++rb;
        const int q = (int)(pair - (int64_t)rb * (rb + 1));
        const int row = rb * 16 + (lane & 15);
        const int k = 8 * q + 4 * slot + (lane >> 4);
        double v = 0.0;
        if (row < n) {
            if (k <= row) v = linv[((int64_t)d * n + row) * n + k];
        } else if (row - n <= D && k < n) {
            const int r = row - n;
            const double al = alpha[(int64_t)d * n + k];
            v = (r == 0) ? al : al * x_train[(int64_t)k * D + r - 1] * args.inv_ls2[d * D + r - 1];
        }
        a_pack[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sx_gp_fit: K_d + noise_d I = L_d L_d^T, W_d = L_d^-1, alpha_d = W_d^T W_d y_d, log det L_d.   Warm path: once per
// update_model.  One 1024-thread workgroup per output, everything in place in global memory (a workgroup lives on one
// CU, so its own stores are visible to it after a barrier); the active column / row is staged in LDS.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFitThreads = 1024;
constexpr int kFitMaxN = 4096;

struct FitArgs {
    double inv_ls2[SX_MAX_NS * SX_MAX_D];
    double outputscale[SX_MAX_NS];
    double noise[SX_MAX_NS];
    const double* x;   // [N x D]
    const double* y;   // [N x n_s]
    double* lmat;      // [n_s x N x N]  K then L (lower triangle)
    double* linv;      // [n_s x N x N]  W = L^-1
    double* alpha;     // [n_s x N]
    double* logdet;    // [n_s]  sum log diag L
    int* status;
    int n, D, n_s, panel_cols;
};

__global__ __launch_bounds__(kFitThreads) void gp_fit_kernel(FitArgs fa) {
    __shared__ double vec[kFitMaxN];
    __shared__ double red[kFitThreads];
    extern __shared__ __attribute__((aligned(16))) double panel[];   // [n x panel_cols]
    const int d = blockIdx.x, tid = threadIdx.x, n = fa.n, D = fa.D;
    double* A = fa.lmat + (size_t)d * n * n;
    double* W = fa.linv + (size_t)d * n * n;
    // 1. kernel matrix (lower triangle)
    for (int64_t idx = tid; idx < (int64_t)n * n; idx += kFitThreads) {
        const int i = (int)(idx / n), j = (int)(idx - (int64_t)i * n);
        if (j <= i) {
            double q = 0.0;
            for (int c = 0; c < D; ++c) {
                const double df = fa.x[(size_t)i * D + c] - fa.x[(size_t)j * D + c];
                q += df * df * fa.inv_ls2[d * D + c];
            }
            A[idx] = fa.outputscale[d] * exp(-0.5 * q) + (i == j ? fa.noise[d] : 0.0);
        }
    }
    __syncthreads();
    // 2. blocked right-looking Cholesky: a panel of nb columns is factored in LDS (its barriers wait on LDS only),
    //    then the trailing matrix gets ONE rank-nb update in HBM per panel instead of one rank-1 update per column
    const int ty = tid >> 6, tx = tid & 63;
    bool bad = false;
    const int nb_max = fa.panel_cols;
    for (int j0 = 0; j0 < n; j0 += nb_max) {
        const int nb = (n - j0 < nb_max) ? n - j0 : nb_max;
        const int rows = n - j0;
        // panel[r][c] = A[j0 + r][j0 + c], r >= c
        for (int idx = tid; idx < rows * nb; idx += kFitThreads) {
            const int r = idx / nb, c = idx - r * nb;
            panel[idx] = (c <= r) ? A[(size_t)(j0 + r) * n + j0 + c] : 0.0;
        }
        __syncthreads();
        for (int jj = 0; jj < nb; ++jj) {
            const double ajj = panel[jj * nb + jj];
            if (!(ajj > 0.0)) bad = true;
            const double piv = sqrt(ajj);
            __syncthreads();   // everyone has read the pivot before it is overwritten
            for (int r = jj + tid; r < rows; r += kFitThreads) panel[r * nb + jj] = (r == jj) ? piv : panel[r * nb + jj] / piv;
            __syncthreads();
            // update the rest of the panel: columns jj+1 .. nb-1, rows >= column
            const int ncols = nb - jj - 1;
            for (int idx = tid; idx < (rows - jj - 1) * ncols; idx += kFitThreads) {
                const int r = jj + 1 + idx / ncols, c = jj + 1 + idx % ncols;
                if (c <= r) panel[r * nb + c] -= panel[r * nb + jj] * panel[c * nb + jj];
            }
            __syncthreads();
        }
        // write the factored panel back
        for (int idx = tid; idx < rows * nb; idx += kFitThreads) {
            const int r = idx / nb, c = idx - r * nb;
            if (c <= r) A[(size_t)(j0 + r) * n + j0 + c] = panel[idx];
        }
        // trailing update: A[i][k] -= sum_c P[i][c] P[k][c]  for j0 + nb <= k <= i
        for (int i = j0 + nb + ty; i < n; i += kFitThreads / 64) {
            double* row = A + (size_t)i * n;
            const double* pi = panel + (size_t)(i - j0) * nb;
            for (int c = j0 + nb + tx; c <= i; c += 64) {
                const double* pc = panel + (size_t)(c - j0) * nb;
                double s = 0.0;
                for (int q = 0; q < nb; ++q) s += pi[q] * pc[q];
                row[c] -= s;
            }
        }
        __syncthreads();
    }
    if (bad && tid == 0) atomicOr(fa.status, 8);
    // 3. W = L^-1, row by row: W[i][c] = (delta_ic - sum_{k=c}^{i-1} L[i][k] W[k][c]) / L[i][i]
    for (int i = 0; i < n; ++i) {
        for (int c = tid; c <= i; c += kFitThreads) vec[c] = A[(size_t)i * n + c];
        __syncthreads();
        const double inv = 1.0 / vec[i];
        for (int c = tid; c < n; c += kFitThreads) {
            double w = 0.0;
            if (c <= i) {
                double s = (c == i) ? 1.0 : 0.0;
                for (int kk = c; kk < i; ++kk) s -= vec[kk] * W[(size_t)kk * n + c];
                w = s * inv;
            }
            W[(size_t)i * n + c] = w;
        }
        __syncthreads();
    }
    // 4. alpha = W^T (W y)
    for (int i = tid; i < n; i += kFitThreads) {
        double s = 0.0;
        for (int c = 0; c <= i; ++c) s += W[(size_t)i * n + c] * fa.y[(size_t)c * fa.n_s + d];
        vec[i] = s;
    }
    __syncthreads();
    for (int c = tid; c < n; c += kFitThreads) {
        double s = 0.0;
        for (int i = c; i < n; ++i) s += W[(size_t)i * n + c] * vec[i];
        fa.alpha[(size_t)d * n + c] = s;
    }
    // 5. sum log diag L
    double ld = 0.0;
    for (int i = tid; i < n; i += kFitThreads) ld += log(A[(size_t)i * n + i]);
    red[tid] = ld;
    __syncthreads();
    for (int off = kFitThreads / 2; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) fa.logdet[d] = red[0];
}

// ---------------------------------------------------------------------------------------------------------------
// sx_gp_mll_grad: exact marginal log likelihood of output d and its gradient w.r.t. (lengthscale_d[0..D), outputscale_d,
// noise_d), from the factorisation sx_gp_fit left behind:
//   mll = -1/2 y.alpha - sum log diag L - N/2 log 2 pi,      d mll / d theta = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta),
//   K^-1 = W^T W.   One workgroup per output; pair (i, j <= i) is handled by thread j (W rows are read coalesced).
// ---------------------------------------------------------------------------------------------------------------
struct MllArgs {
    double inv_ls2[SX_MAX_NS * SX_MAX_D];
    double outputscale[SX_MAX_NS];
    double noise[SX_MAX_NS];
    const double* x;
    const double* y;
    const double* linv;
    const double* alpha;
    const double* logdet;
    double* mll;    // [n_s]
    double* grad;   // [n_s x (D + 2)]
    int n, D, n_s;
};

__global__ __launch_bounds__(kFitThreads) void gp_mll_grad_kernel(MllArgs ma) {
    __shared__ double red[kFitThreads];
    const int d = blockIdx.x, tid = threadIdx.x, n = ma.n, D = ma.D;
    const double* W = ma.linv + (size_t)d * n * n;
    const double* al = ma.alpha + (size_t)d * n;
    double acc[SX_MAX_D + 2];
#pragma unroll
    for (int c = 0; c < SX_MAX_D + 2; ++c) acc[c] = 0.0;
    double ya = 0.0;
    for (int i = 0; i < n; ++i) {
        const double ai = al[i];
        for (int j = tid; j <= i; j += kFitThreads) {
            double kinv = 0.0;
            for (int r = i; r < n; ++r) kinv += W[(size_t)r * n + i] * W[(size_t)r * n + j];
            const double g = ai * al[j] - kinv;
            double q = 0.0;
            double dq[SX_MAX_D];
            for (int c = 0; c < D; ++c) {
                const double df = ma.x[(size_t)i * D + c] - ma.x[(size_t)j * D + c];
                dq[c] = df * df * ma.inv_ls2[d * D + c];   // (x_ic - x_jc)^2 / l_c^2
                q += dq[c];
            }
            const double kij = ma.outputscale[d] * exp(-0.5 * q);
            const double w = (i == j) ? 0.5 : 1.0;         // 1/2 tr(...) over the symmetric pair
            for (int c = 0; c < D; ++c) acc[c] += w * g * kij * dq[c];          // * 1 / l_c applied below
            acc[D] += w * g * kij;                                              // * 1 / s applied below
            if (i == j) acc[D + 1] += 0.5 * g;
        }
        if (i % kFitThreads == tid) ya += ma.y[(size_t)i * ma.n_s + d] * ai;
    }
    for (int c = 0; c < D + 3; ++c) {
        double v = (c < D + 2) ? acc[c] : ya;
        red[tid] = v;
        __syncthreads();
        for (int off = kFitThreads / 2; off > 0; off >>= 1) {
            if (tid < off) red[tid] += red[tid + off];
            __syncthreads();
        }
        if (tid == 0) {
            const double tot = red[0];
            if (c < D)
                ma.grad[d * (D + 2) + c] = tot * sqrt(ma.inv_ls2[d * D + c]);   // dK/dl_c = K (x_i - x_j)^2 / l_c^3
            else if (c == D)
                ma.grad[d * (D + 2) + D] = tot / ma.outputscale[d];
            else if (c == D + 1)
                ma.grad[d * (D + 2) + D + 1] = tot;
            else
                ma.mll[d] = -0.5 * tot - ma.logdet[d] - 0.5 * n * 1.8378770664093453;   // log(2 pi)
        }
        __syncthreads();
    }
}

__global__ void build_stage_tab_kernel(int4* tab, int ns, int n_train, int n_pad, int nw, int stage_cap) {
    if ((int)threadIdx.x < nw) gp_build_stage_tab(tab, ns, n_train, n_pad, nw, stage_cap, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// sx_gp_predict: one 16-point tile per workgroup
// ---------------------------------------------------------------------------------------------------------------
template <int NS, int NU>
__global__ __launch_bounds__(kPredictThreads) void gp_predict_kernel(GpConst<NS, NS + NU> gc,
                                                                     const int4* __restrict__ stage_tab,
                                                                     const double* __restrict__ z, int P,
                                                                     double* __restrict__ mean, double* __restrict__ var,
                                                                     double* __restrict__ jac) {
    constexpr int D = NS + NU;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    GpTileLds<NS, D> lds;
    const int nw = blockDim.x >> 6;
    lds.carve(smem, gc.n_train, gc.n_pad, nw);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    gp_load_xs(gc, lds);
    for (int tile = blockIdx.x; tile * SX_TILE < P; tile += gridDim.x) {
        const int g0 = tile * SX_TILE;
        if (tid < SX_TILE * D) {
            const int c = tid / D, j = tid - c * D;
            lds.zs[tid] = (g0 + c < P) ? z[(int64_t)(g0 + c) * D + j] : 0.0;
        }
        __syncthreads();
        gp_kstar_phase(gc, lds);
        __syncthreads();
        gp_mfma_phase(gc, stage_tab, lds, wave, nw, lane);
        __syncthreads();
        if (tid < SX_TILE && g0 + tid < P) {
            double zz[D], m[NS], v[NS], jc[NS][D];
#pragma unroll
            for (int j = 0; j < D; ++j) zz[j] = lds.zs[tid * D + j];
            if (jac) {
                gp_collect<NS, D, true>(gc, lds, nw, tid, zz, m, v, jc);
            } else {
                gp_collect<NS, D, false>(gc, lds, nw, tid, zz, m, v, jc);
            }
            const int64_t g = g0 + tid;
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
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sx_onestep_reach / sx_polytope_distance: one particle per lane
// ---------------------------------------------------------------------------------------------------------------
template <int NS, int NU>
__global__ void onestep_reach_kernel(ReachConst<NS, NU> rc, int P, const double* __restrict__ p_in,
                                     const double* __restrict__ q_in, const double* __restrict__ u_in,
                                     const double* __restrict__ mean_in, const double* __restrict__ var_in,
                                     const double* __restrict__ jac_in, double* __restrict__ p_out,
                                     double* __restrict__ q_out, double* __restrict__ sig_out, int* __restrict__ status) {
    constexpr int D = NS + NU;
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= P) return;
    double p[NS], u[NU], mean[NS], var[NS], p1[NS], Q1[NS][NS];
    int st = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        p[i] = p_in[g * NS + i];
        mean[i] = mean_in[g * NS + i];
        var[i] = var_in[g * NS + i];
    }
#pragma unroll
    for (int c = 0; c < NU; ++c) u[c] = u_in[g * NU + c];
    if (q_in == nullptr) {
        reach_point<NS, NU>(rc, p, u, mean, var, p1, Q1, st);
    } else {
        double Q[NS][NS], jac[NS][D];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
#pragma unroll
            for (int j = 0; j < NS; ++j) Q[i][j] = q_in[(g * NS + i) * NS + j];
#pragma unroll
            for (int j = 0; j < D; ++j) jac[i][j] = jac_in[(g * NS + i) * D + j];
        }
        reach_ellipsoid<NS, NU>(rc, p, Q, u, mean, var, jac, p1, Q1, st);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        p_out[g * NS + i] = p1[i];
        sig_out[g * NS + i] = var[i];
#pragma unroll
        for (int j = 0; j < NS; ++j) q_out[(g * NS + i) * NS + j] = Q1[i][j];
    }
    if (st) atomicOr(status, st);
}

template <int NS>
struct PolyArgs {
    double h_mat[SX_MAX_M * NS];
    double h_vec[SX_MAX_M];
    int m;
};

template <int NS>
__global__ void polytope_kernel(PolyArgs<NS> pa, int P, double c_safety, const double* __restrict__ p_in,
                                const double* __restrict__ q_in, double* __restrict__ d_out,
                                uint8_t* __restrict__ inside) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= P) return;
    double p[NS], Q[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        p[i] = p_in[g * NS + i];
#pragma unroll
        for (int j = 0; j < NS; ++j) Q[i][j] = q_in[(g * NS + i) * NS + j];
    }
    double d[SX_MAX_M];
    const bool viol = polytope_violated<SX_MAX_M, NS>(pa.h_mat, pa.h_vec, pa.m, c_safety, p, Q, d);
    for (int r = 0; r < pa.m; ++r) d_out[g * pa.m + r] = d[r];
    if (inside) inside[g] = viol ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------------------------
// sx_cem_rollout: the fused H-step particle rollout.  One workgroup = 16 particles of one problem for all H steps.
// ---------------------------------------------------------------------------------------------------------------
#ifdef SX_STAMPS
// Diagnostic build only (tools/phase_stamps.py): per-workgroup cycle sums of the three phases of a step.
__device__ unsigned long long* g_stamp_buf = nullptr;
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#endif

struct RolloutPtrs {
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
__global__ __launch_bounds__(kRolloutThreads) void cem_rollout_kernel(GpConst<NS, NS + NU> gc,
                                                                      const int4* __restrict__ stage_tab,
                                                                      ReachConst<NS, NU> rc,
                                                                      CostConst<SX_MAX_M, NS, NU> cc, RolloutPtrs rp) {
    constexpr int D = NS + NU;
    constexpr int S = NS + NS * NS;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    GpTileLds<NS, D> lds;
    const int nw = blockDim.x >> 6;
    double* acts = lds.carve(smem, gc.n_train, gc.n_pad, nw);  // [16][H][NU]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int H = rp.H;
    const int tiles_per_problem = (rp.P + SX_TILE - 1) / SX_TILE;
    const int e = blockIdx.x / tiles_per_problem;
    const int c0 = (blockIdx.x - e * tiles_per_problem) * SX_TILE;  // first particle of the tile within problem e

    gp_load_xs(gc, lds);
    // sample (or load) this tile's action sequences: a = mean + std * eps
    for (int i = tid; i < SX_TILE * H * NU; i += blockDim.x) {
        const int c = i / (H * NU);
        const int r = i - c * (H * NU);
        double a = 0.0;
        if (c0 + c < rp.P) {
            const int64_t gi = ((int64_t)e * rp.P + c0 + c) * (H * NU) + r;
            if (rp.noise) {
                a = rp.mean[(int64_t)e * H * NU + r] + rp.std[(int64_t)e * H * NU + r] * rp.noise[gi];
                rp.actions[gi] = a;
            } else {
                a = rp.actions[gi];
            }
        }
        acts[i] = a;
    }
    // per-particle state lives in the registers of thread c (tid < 16) for the whole rollout
    const bool owner = tid < SX_TILE;
    const bool valid = owner && (c0 + tid < rp.P);
    double p[NS], Q[NS][NS];
    bool have_q = rp.q0 != nullptr;
    double obj = 0.0, con = 0.0;
    int st = 0;
    if (owner) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            p[i] = rp.x0[(int64_t)e * NS + i];
#pragma unroll
            for (int j = 0; j < NS; ++j) Q[i][j] = have_q ? rp.q0[((int64_t)e * NS + i) * NS + j] : 0.0;
        }
    }
    __syncthreads();
    if (owner) {
#pragma unroll
        for (int i = 0; i < NS; ++i) lds.zs[tid * D + i] = p[i];
#pragma unroll
        for (int cidx = 0; cidx < NU; ++cidx) lds.zs[tid * D + NS + cidx] = acts[(tid * H + 0) * NU + cidx];
    }
    __syncthreads();

    // Step t:   Kstar(t)  |sync|  MFMA(t)  |sync|  quick(t): p_{t+1} = mean + a p + b u  ->  zs  |sync|
    // The rest of step t (variance, Jacobian, ellipsoid algebra, costs: ~4.5k cycles on 16 lanes) does not feed
    // Kstar(t+1), so wave 0 runs it DURING Kstar(t+1) while waves 1..7 compute the kernel rows.
    double pn[NS];  // p_{t+1} from quick(t), kept for finish(t)
    auto quick = [&](int t) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            double s = lds.mj[i * 256 + tid];  // posterior mean of output i
#pragma unroll
            for (int j = 0; j < NS; ++j) s += rc.a[i * NS + j] * p[j];
#pragma unroll
            for (int cidx = 0; cidx < NU; ++cidx) s += rc.b[i * NU + cidx] * acts[(tid * H + t) * NU + cidx];
            pn[i] = s;
        }
        if (t + 1 < H) {
#pragma unroll
            for (int i = 0; i < NS; ++i) lds.zs[tid * D + i] = pn[i];
#pragma unroll
            for (int cidx = 0; cidx < NU; ++cidx) lds.zs[tid * D + NS + cidx] = acts[(tid * H + t + 1) * NU + cidx];
        }
    };
    auto finish = [&](int t) {
        double z[D], u[NU], mean[NS], var[NS], jac[NS][D], p1[NS], Q1[NS][NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) z[j] = p[j];
#pragma unroll
        for (int cidx = 0; cidx < NU; ++cidx) {
            u[cidx] = acts[(tid * H + t) * NU + cidx];
            z[NS + cidx] = u[cidx];
        }
        int st_step = 0;
        if (have_q) {
            gp_collect<NS, D, true>(gc, lds, nw, tid, z, mean, var, jac);
            reach_ellipsoid<NS, NU>(rc, p, Q, u, mean, var, jac, p1, Q1, st_step);
        } else {
            gp_collect<NS, D, false>(gc, lds, nw, tid, z, mean, var, jac);
            reach_point<NS, NU>(rc, p, u, mean, var, p1, Q1, st_step);
        }
        have_q = true;
#pragma unroll
        for (int i = 0; i < NS; ++i) p1[i] = pn[i];  // exactly the centre the next GP query used
        if (valid) st |= st_step;
        // costs (safempc_cem.py:102-132,304-312; action constraint: test_safempc_cem.py:59-71)
        obj += objective_cost<SX_MAX_M, NS, NU>(cc, p1, var);
        bool uviol = false;
#pragma unroll
        for (int cidx = 0; cidx < NU; ++cidx) uviol = uviol || (u[cidx] < cc.u_min[cidx]) || (u[cidx] > cc.u_max[cidx]);
        if (uviol) con += SX_ACTION_VIOLATION_COST;
        if (cc.con_mode == SX_CON_ALL_STATES || t == H - 1) {
            if (polytope_violated<SX_MAX_M, NS>(cc.h_mat, cc.h_vec, cc.m, 1.0, p1, Q1, nullptr))
                con += SX_STATE_VIOLATION_COST;
        }
        const int64_t g = (int64_t)e * rp.P + c0 + tid;
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
    };

#ifdef SX_STAMPS
    unsigned long long c_k = 0, c_kb = 0, c_m = 0, c_mb = 0, c_e = 0, c_eb = 0;
#endif
    for (int t = 0; t < H; ++t) {
#ifdef SX_STAMPS
        const unsigned long long t0 = stamp();
#endif
        if (t == 0) {
            gp_kstar_phase(gc, lds);
        } else if (wave == 0) {
            if (owner) finish(t - 1);
        } else {
            gp_kstar_phase(gc, lds, 64);
        }
#ifdef SX_STAMPS
        const unsigned long long t1 = stamp();
#endif
        __syncthreads();
#ifdef SX_STAMPS
        const unsigned long long t2 = stamp();
#endif
        gp_mfma_phase(gc, stage_tab, lds, wave, nw, lane);
#ifdef SX_STAMPS
        const unsigned long long t3 = stamp();
#endif
        __syncthreads();
#ifdef SX_STAMPS
        const unsigned long long t4 = stamp();
#endif
        if (owner) quick(t);
#ifdef SX_STAMPS
        const unsigned long long t5 = stamp();
#endif
        __syncthreads();
#ifdef SX_STAMPS
        const unsigned long long t6 = stamp();
        c_k += t1 - t0; c_kb += t2 - t1; c_m += t3 - t2; c_mb += t4 - t3; c_e += t5 - t4; c_eb += t6 - t5;
#endif
    }
    if (owner) finish(H - 1);
#ifdef SX_STAMPS
    if (g_stamp_buf && lane == 0) {
        unsigned long long* o = g_stamp_buf + ((size_t)blockIdx.x * nw + wave) * 8;
        o[0] = c_k; o[1] = c_kb; o[2] = c_m; o[3] = c_mb; o[4] = c_e; o[5] = c_eb;
    }
#endif
    if (valid) {
        const int64_t g = (int64_t)e * rp.P + c0 + tid;
        rp.obj_cost[g] = obj;
        rp.con_cost[g] = con;
        if (st) atomicOr(rp.status, st);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sx_cem_rank_refit: one workgroup (16 waves) per problem.
//   1. every thread keeps its candidates' 128-bit keys (con, obj) in registers: element i lives in slot i / 1024 of
//      thread i % 1024, so (slot, thread) order is index order;
//   2. MSB-first radix select of the k-th key, 8 bits per pass: wave-aggregated LDS histogram (one atomic per wave when
//      all lanes agree -- the common case in the high bytes), bin scan by one wave, early exit as soon as the bin
//      holding the k-th key is wholly selected;
//   3. ballot compaction in index order (ties broken by the lower index); the best survivor is moved to the front,
//      the others stay where the compaction put them (nothing downstream needs them sorted);
//   4. refit: mean / unbiased std over the elites, rows spread over the whole workgroup.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRankThreads = 1024;
constexpr int kRankWaves = kRankThreads / 64;
constexpr int kRankMaxK = 2048;
constexpr int kRankSlots = 16;  // candidates per thread held in registers: P <= 16384

__device__ __forceinline__ unsigned long long sortable_key(double x) {
    if (x != x) return ~0ull;  // NaN last
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

struct RankArgs {
    int P, k, row_len;
    const double* con;
    const double* obj;
    long long cost_stride;
    const double* actions;
    long long act_stride;
    int* elite_idx;
    double* elite_rows;
    double* mean;
    double* std;
    double* best;
    int* best_ok;
};

template <int SLOTS>
__global__ __launch_bounds__(kRankThreads) void cem_rank_kernel(RankArgs ra) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long sel_hi[kRankMaxK], sel_lo[kRankMaxK];
    __shared__ int sel_idx[kRankMaxK];
    __shared__ double red[kRankThreads];
    __shared__ double col_mean[256];
    __shared__ int wave_cnt[kRankWaves][2];
    __shared__ unsigned long long red_u64[kRankWaves], red_lo[kRankWaves];
    __shared__ int sh_digit, sh_need, sh_done;

    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int P = ra.P, k = ra.k;
    const double* con = ra.con + (long long)e * P * ra.cost_stride;
    const double* obj = ra.obj + (long long)e * P * ra.cost_stride;
    const double* act = ra.actions + (long long)e * P * ra.act_stride;

    unsigned long long kh[SLOTS], kl[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = s * kRankThreads + tid;
        kh[s] = ~0ull;
        kl[s] = ~0ull;
        if (i < P) {
            kh[s] = sortable_key(con[(long long)i * ra.cost_stride]);
            kl[s] = sortable_key(obj[(long long)i * ra.cost_stride]);
        }
    }

#ifdef SX_STAMPS
    const unsigned long long ts0 = stamp();
#endif
    // ---- radix select ----
    unsigned long long ph = 0, pl = 0;   // prefix of the k-th key found so far (uniform)
    unsigned long long mh = 0, ml = 0;   // mask of the prefix bits
    int need = k;                        // rank of the k-th key among the candidates matching the prefix
    bool done = false;
    int first_pass = 0;
    {
        // The constraint word takes few distinct values (0 for every feasible particle, then 3 a + 10 b), so its
        // k-th smallest value is found by walking up the distinct values: one (min, multiplicity) reduction per value,
        // at most 8 of them, instead of eight radix passes.  (Beyond 8 the general passes below take over.)
        unsigned long long floor_key = 0;  // only keys >= floor_key are still in play
        int acc = 0;                       // candidates below floor_key
        for (int it = 0; it < 8; ++it) {
            unsigned long long mn = ~0ull;
            int cnt = 0;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const bool in_play = (s * kRankThreads + tid < P) && kh[s] >= floor_key;
                if (in_play) {
                    if (kh[s] < mn) { mn = kh[s]; cnt = 1; } else if (kh[s] == mn) { ++cnt; }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long om = __shfl_xor(mn, off);
                const int oc = __shfl_xor(cnt, off);
                if (om < mn) { mn = om; cnt = oc; } else if (om == mn) { cnt += oc; }
            }
            if (lane == 0) { red_u64[wave] = mn; wave_cnt[wave][0] = cnt; }
            __syncthreads();
            mn = red_u64[0];
            cnt = wave_cnt[0][0];
#pragma unroll
            for (int w = 1; w < kRankWaves; ++w) {
                const unsigned long long om = red_u64[w];
                const int oc = wave_cnt[w][0];
                if (om < mn) { mn = om; cnt = oc; } else if (om == mn) { cnt += oc; }
            }
            __syncthreads();
            if (acc + cnt >= k) {   // the k-th key has this constraint word
                ph = mn;
                mh = ~0ull;
                need = k - acc;
                first_pass = 8;
                break;
            }
            acc += cnt;
            floor_key = mn + 1;
        }
    }
    for (int pass = first_pass; pass < 16 && !done; ++pass) {
        const int shift = 56 - 8 * (pass & 7);
        const bool in_hi = pass < 8;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int i = s * kRankThreads + tid;
            const bool match = (i < P) && ((kh[s] & mh) == ph) && ((kl[s] & ml) == pl);
            const unsigned int digit = match ? (unsigned int)(((in_hi ? kh[s] : kl[s]) >> shift) & 255ull) : 0xffffffffu;
            const unsigned int first = __builtin_amdgcn_readfirstlane(digit);
            if (__all(digit == first)) {
                if (first != 0xffffffffu && lane == 0) atomicAdd(&hist[first], 64u);
            } else if (match) {
                atomicAdd(&hist[digit], 1u);
            }
        }
        __syncthreads();
        if (wave == 0) {
            // lane l owns bins 4l .. 4l+3
            const unsigned int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
            const int mine = (int)(c0 + c1 + c2 + c3);
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            const int before = incl - mine;
            if (need > before && need <= incl) {
                int rem = need - before;
                int dsel = 4 * lane;
                unsigned int cnt = c0;
                if (rem > (int)c0) { rem -= c0; dsel++; cnt = c1;
                    if (rem > (int)c1) { rem -= c1; dsel++; cnt = c2;
                        if (rem > (int)c2) { rem -= c2; dsel++; cnt = c3; } } }
                sh_digit = dsel;
                sh_need = rem;
                sh_done = (rem == (int)cnt) ? 1 : 0;  // the whole bin is selected: no need to look at lower digits
            }
        }
        __syncthreads();
        const unsigned long long dg = (unsigned long long)sh_digit << shift, mk = 255ull << shift;
        if (in_hi) { ph |= dg; mh |= mk; } else { pl |= dg; ml |= mk; }
        need = sh_need;
        done = sh_done != 0;
    }
#ifdef SX_STAMPS
    const unsigned long long ts1 = stamp();
#endif
    // Candidates whose masked key is below the prefix are selected; of those equal to it, the first `need` in index
    // order (all of them after an early exit).
    const int n_less_total = k - need;
    int base_less = 0, base_tie = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        if (s * kRankThreads >= P) break;
        const int i = s * kRankThreads + tid;
        const unsigned long long a_h = kh[s] & mh, a_l = kl[s] & ml;
        const bool valid = i < P;
        const bool less = valid && (a_h < ph || (a_h == ph && a_l < pl));
        const bool tie = valid && a_h == ph && a_l == pl;
        const unsigned long long bl = __ballot(less), bt = __ballot(tie);
        if (lane == 0) {
            wave_cnt[wave][0] = __popcll(bl);
            wave_cnt[wave][1] = __popcll(bt);
        }
        __syncthreads();
        int off_less = base_less, off_tie = base_tie, tot_less = 0, tot_tie = 0;
#pragma unroll
        for (int w = 0; w < kRankWaves; ++w) {
            const int cl = wave_cnt[w][0], ct = wave_cnt[w][1];
            if (w < wave) { off_less += cl; off_tie += ct; }
            tot_less += cl;
            tot_tie += ct;
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (less) {
            const int slot = off_less + __popcll(bl & below);
            sel_hi[slot] = kh[s]; sel_lo[slot] = kl[s]; sel_idx[slot] = i;
        } else if (tie) {
            const int r = off_tie + __popcll(bt & below);
            if (r < need) {
                const int slot = n_less_total + r;
                sel_hi[slot] = kh[s]; sel_lo[slot] = kl[s]; sel_idx[slot] = i;
            }
        }
        base_less += tot_less;
        base_tie += tot_tie;
        __syncthreads();
    }
#ifdef SX_STAMPS
    const unsigned long long ts2 = stamp();
#endif
    // ---- the elites stay where the compaction put them; only the best one is moved to the front ----
    {
        unsigned long long bh = ~0ull, bl = ~0ull;
        int bi = 0x7fffffff, bslot = 0;
        for (int i = tid; i < k; i += kRankThreads) {
            const unsigned long long h = sel_hi[i], l = sel_lo[i];
            const int ix = sel_idx[i];
            if (h < bh || (h == bh && (l < bl || (l == bl && ix < bi)))) { bh = h; bl = l; bi = ix; bslot = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long oh = __shfl_xor(bh, off), ol = __shfl_xor(bl, off);
            const int oi = __shfl_xor(bi, off), os = __shfl_xor(bslot, off);
            if (oh < bh || (oh == bh && (ol < bl || (ol == bl && oi < bi)))) { bh = oh; bl = ol; bi = oi; bslot = os; }
        }
        if (lane == 0) { red_u64[wave] = bh; red_lo[wave] = bl; wave_cnt[wave][0] = bi; wave_cnt[wave][1] = bslot; }
        __syncthreads();
        if (tid == 0) {
            int best_w = 0;
            for (int w = 1; w < kRankWaves; ++w) {
                const unsigned long long oh = red_u64[w], ol = red_lo[w], ch = red_u64[best_w], cl = red_lo[best_w];
                if (oh < ch || (oh == ch && (ol < cl || (ol == cl && wave_cnt[w][0] < wave_cnt[best_w][0])))) best_w = w;
            }
            const int s = wave_cnt[best_w][1];
            const unsigned long long th = sel_hi[0], tl = sel_lo[0];
            const int ti = sel_idx[0];
            sel_hi[0] = sel_hi[s]; sel_lo[0] = sel_lo[s]; sel_idx[0] = sel_idx[s];
            sel_hi[s] = th; sel_lo[s] = tl; sel_idx[s] = ti;
        }
        __syncthreads();
    }
#ifdef SX_STAMPS
    const unsigned long long ts3 = stamp();
#endif
    // ---- outputs ----
    const int L = ra.row_len;
    if (ra.elite_idx)
        for (int i = tid; i < k; i += kRankThreads) ra.elite_idx[(long long)e * k + i] = sel_idx[i];
    if (ra.elite_rows) {
        const int W = 2 + L;
        for (int i = tid; i < k * W; i += kRankThreads) {
            const int r = i / W, c = i - r * W;
            const int src = sel_idx[r];
            double v;
            if (c == 0)
                v = con[(long long)src * ra.cost_stride];
            else if (c == 1)
                v = obj[(long long)src * ra.cost_stride];
            else
                v = act[(long long)src * ra.act_stride + (c - 2)];
            ra.elite_rows[((long long)e * k + r) * W + c] = v;
        }
    }
    if (ra.best)
        for (int c = tid; c < L; c += kRankThreads) ra.best[(long long)e * L + c] = act[(long long)sel_idx[0] * ra.act_stride + c];
    if (ra.best_ok && tid == 0) ra.best_ok[e] = (con[(long long)sel_idx[0] * ra.cost_stride] == 0.0) ? 1 : 0;
    if (ra.mean) {
        // columns in chunks of up to 256; thread t sums rows t / Lc, t / Lc + R, ... of column t % Lc
        for (int c0 = 0; c0 < L; c0 += 256) {
            const int Lc = (L - c0) < 256 ? (L - c0) : 256;
            const int R = kRankThreads / Lc;  // row groups
            const int c = tid % Lc, r0 = tid / Lc;
            const bool active = r0 < R;
            double s = 0.0;
            if (active)
                for (int r = r0; r < k; r += R) s += act[(long long)sel_idx[r] * ra.act_stride + c0 + c];
            red[tid] = s;
            __syncthreads();
            if (tid < Lc) {
                double t = 0.0;
                for (int g = 0; g < R; ++g) t += red[g * Lc + tid];
                col_mean[tid] = t / k;
            }
            __syncthreads();
            const double mu = col_mean[c];
            double ss = 0.0;
            if (active)
                for (int r = r0; r < k; r += R) {
                    const double dv = act[(long long)sel_idx[r] * ra.act_stride + c0 + c] - mu;
                    ss += dv * dv;
                }
            red[tid] = ss;
            __syncthreads();
            if (tid < Lc) {
                double t = 0.0;
                for (int g = 0; g < R; ++g) t += red[g * Lc + tid];
                ra.mean[(long long)e * L + c0 + tid] = col_mean[tid];
                if (ra.std) ra.std[(long long)e * L + c0 + tid] = (k > 1) ? sqrt(t / (k - 1)) : 0.0;
            }
            __syncthreads();
        }
    }
#ifdef SX_STAMPS
    const unsigned long long ts4 = stamp();
    if (g_stamp_buf && tid == 0 && e == 0) {
        g_stamp_buf[0] = ts1 - ts0; g_stamp_buf[1] = ts2 - ts1; g_stamp_buf[2] = ts3 - ts2; g_stamp_buf[3] = ts4 - ts3;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------------------------
template <int NS, int NU>
static GpConst<NS, NS + NU> make_gp_const(const sx_gp_model* m, int nw) {
    constexpr int D = NS + NU;
    GpConst<NS, D> gc;
    for (int d = 0; d < NS; ++d) {
        for (int j = 0; j < D; ++j) {
            gc.inv_ls2[d * D + j] = m->inv_ls2[d * D + j];
            gc.nh_ils2[d * D + j] = -0.5 * m->inv_ls2[d * D + j];
        }
        gc.log_os[d] = std::log(m->outputscale[d]);
        gc.outputscale[d] = m->outputscale[d];
        gc.noise[d] = m->noise[d];
    }
    gc.x_train = m->x_train;
    gc.a_pack = m->a_pack;
    gc.stage_tab = reinterpret_cast<const int4*>(m->stage_tab);
    gc.n_train = m->n_train;
    gc.n_pad = m->n_pad;
    gc.stage_cap = gp_stage_cap(NS, m->n_pad, nw);
    return gc;
}

template <int NS, int NU>
static bool make_reach_const(const sx_env* env, ReachConst<NS, NU>& rc) {
    for (int i = 0; i < NS * NS; ++i) rc.a[i] = env->a[i];
    for (int i = 0; i < NS * NU; ++i) rc.b[i] = env->b[i];
    for (int i = 0; i < NU * NS; ++i) rc.kfb[i] = env->k_fb[i];
    for (int i = 0; i < NS; ++i) {
        rc.l_mu[i] = env->l_mu[i];
        rc.l_sigma[i] = env->l_sigma[i];
    }
    rc.beta = env->beta;
    // B = I + kfb^T kfb is SPD; lower Cholesky on the host
    double B[NS][NS];
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int c = 0; c < NU; ++c) s += env->k_fb[c * NS + i] * env->k_fb[c * NS + j];
            B[i][j] = s;
        }
    for (int i = 0; i < NS * NS; ++i) rc.cholB[i] = 0.0;
    for (int j = 0; j < NS; ++j) {
        double s = B[j][j];
        for (int k = 0; k < j; ++k) s -= rc.cholB[j * NS + k] * rc.cholB[j * NS + k];
        if (!(s > 0.0)) return false;
        const double ljj = std::sqrt(s);
        rc.cholB[j * NS + j] = ljj;
        for (int i = j + 1; i < NS; ++i) {
            double t = B[i][j];
            for (int k = 0; k < j; ++k) t -= rc.cholB[i * NS + k] * rc.cholB[j * NS + k];
            rc.cholB[i * NS + j] = t / ljj;
        }
    }
    return true;
}

template <int NS, int NU>
static void make_cost_const(const sx_env* env, CostConst<SX_MAX_M, NS, NU>& cc) {
    std::memset(&cc, 0, sizeof(cc));
    for (int r = 0; r < env->m; ++r) {
        for (int i = 0; i < NS; ++i) cc.h_mat[r * NS + i] = env->h_mat[r * NS + i];
        cc.h_vec[r] = env->h_vec[r];
    }
    for (int c = 0; c < NU; ++c) {
        cc.u_min[c] = env->u_min[c];
        cc.u_max[c] = env->u_max[c];
    }
    for (int i = 0; i < NS; ++i) {
        cc.w_abs[i] = env->obj_w_abs[i];
        cc.target[i] = env->obj_target[i];
        cc.w_lin[i] = env->obj_w_lin[i];
    }
    cc.m = env->m;
    cc.obj_mode = env->obj_mode;
    cc.con_mode = env->con_mode;
}

static int check_launch() {
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        std::fprintf(stderr, "libsxamd: HIP launch error: %s\n", hipGetErrorString(err));
        return SX_ERR_LAUNCH;
    }
    return SX_OK;
}


template <typename K>
static int allow_lds(K kernel, size_t bytes) {
    if (bytes > kMaxLdsBytes) return SX_ERR_UNSUPPORTED;
    if (bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            return SX_ERR_UNSUPPORTED;
        }
    }
    return SX_OK;
}

static bool predict_fits(int ns, int nu, int n_train, int n_pad) {
    const int nw = kPredictThreads / 64;
    return gp_tile_lds_doubles(ns, ns + nu, n_train, n_pad, nw) * sizeof(double) <= kMaxLdsBytes && n_pad <= 1024;
}

template <int NS, int NU>
static int launch_predict_big(const sx_gp_model* m, const double* z, int P, double* mean, double* var, double* jac,
                              double* workspace, int64_t workspace_bytes, hipStream_t stream);

template <int NS, int NU>
static int launch_predict(const sx_gp_model* m, const double* z, int P, double* mean, double* var, double* jac,
                          double* workspace, int64_t workspace_bytes, hipStream_t stream) {
    if (!predict_fits(NS, NU, m->n_train, m->n_pad))
        return launch_predict_big<NS, NU>(m, z, P, mean, var, jac, workspace, workspace_bytes, stream);
    const int nw = kPredictThreads / 64;
    auto gc = make_gp_const<NS, NU>(m, nw);
    const size_t lds = gp_tile_lds_doubles(NS, NS + NU, m->n_train, m->n_pad, nw) * sizeof(double);
    if (int rc = allow_lds(gp_predict_kernel<NS, NU>, lds)) return rc;
    const int tiles = (P + SX_TILE - 1) / SX_TILE;
    const int grid = tiles < 4096 ? tiles : 4096;
    hipLaunchKernelGGL((gp_predict_kernel<NS, NU>), dim3(grid), dim3(kPredictThreads), lds, stream, gc, gc.stage_tab, z, P,
                       mean, var, jac);
    return check_launch();
}

template <int NS, int NU>
static int launch_reach(const sx_env* env, int P, const double* p, const double* Q, const double* u, const double* mean,
                        const double* var, const double* jac, double* p1, double* Q1, double* sigma, int* status,
                        hipStream_t stream) {
    ReachConst<NS, NU> rc;
    if (!make_reach_const<NS, NU>(env, rc)) return SX_ERR_ARG;
    const int threads = 64;
    hipLaunchKernelGGL((onestep_reach_kernel<NS, NU>), dim3((P + threads - 1) / threads), dim3(threads), 0, stream, rc,
                       P, p, Q, u, mean, var, jac, p1, Q1, sigma, status);
    return check_launch();
}

template <int NS>
static int launch_polytope(const sx_env* env, int P, const double* p, const double* Q, double c_safety, double* d,
                           uint8_t* inside, hipStream_t stream) {
    PolyArgs<NS> pa;
    std::memset(&pa, 0, sizeof(pa));
    for (int r = 0; r < env->m; ++r) {
        for (int i = 0; i < NS; ++i) pa.h_mat[r * NS + i] = env->h_mat[r * NS + i];
        pa.h_vec[r] = env->h_vec[r];
    }
    pa.m = env->m;
    const int threads = 64;
    hipLaunchKernelGGL((polytope_kernel<NS>), dim3((P + threads - 1) / threads), dim3(threads), 0, stream, pa, P,
                       c_safety, p, Q, d, inside);
    return check_launch();
}

// ---- sx_gp_predict for training sets beyond the LDS budget: the same Kstar / triangular-product kernels, then collect ----
template <int NS, int D>
__global__ void predict_init_big_kernel(const double* __restrict__ z, int64_t P, int64_t p128, BigWs ws) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= p128 * D) return;
    ws.zs[i] = (i < P * D) ? z[i] : 0.0;
}

template <int NS, int D>
__global__ void predict_collect_big_kernel(GpConst<NS, D> gc, BigWs ws, int64_t P, int64_t p128, int row_parts,
                                           double* __restrict__ mean, double* __restrict__ var, double* __restrict__ jac) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= P) return;
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        double q = 0.0;
        for (int r = 0; r < row_parts; ++r) q += ws.part[((int64_t)d * row_parts + r) * p128 + g];
        var[g * NS + d] = (gc.outputscale[d] - q) + gc.noise[d];
        const double m = ws.mj[((int64_t)d * (D + 1)) * p128 + g];
        mean[g * NS + d] = m;
        if (jac) {
#pragma unroll
            for (int j = 0; j < D; ++j)
                jac[(g * NS + d) * D + j] =
                    ws.mj[((int64_t)d * (D + 1) + 1 + j) * p128 + g] - ws.zs[g * D + j] * gc.inv_ls2[d * D + j] * m;
        }
    }
}

template <int NS, int NU>
static int launch_predict_big(const sx_gp_model* m, const double* z, int P, double* mean, double* var, double* jac,
                              double* workspace, int64_t workspace_bytes, hipStream_t stream) {
    constexpr int D = NS + NU;
    auto gc = make_gp_const<NS, NU>(m, kPredictThreads / 64);
    const int64_t p128 = ((int64_t)P + kBigTile - 1) / kBigTile * kBigTile;
    BigWs ws = big_ws_layout(workspace, NS, D, m->n_pad, P);
    if (!workspace || workspace_bytes < ws.total * (int64_t)sizeof(double)) return SX_ERR_ARG;
    const int row_tiles = (m->n_pad + kBigTile - 1) / kBigTile;
    hipLaunchKernelGGL((predict_init_big_kernel<NS, D>), dim3((unsigned)((p128 * D + 255) / 256)), dim3(256), 0, stream, z,
                       (int64_t)P, p128, ws);
    hipLaunchKernelGGL((kstar_big_kernel<NS, D>), dim3((unsigned)(p128 / 16), (unsigned)((m->n_pad + 255) / 256)), dim3(256),
                       0, stream, gc, ws);
    hipLaunchKernelGGL((trmm_reduce_kernel<NS, D>), dim3((unsigned)(p128 / kBigTile), (unsigned)row_tiles, NS),
                       dim3(kBigThreads), 0, stream, gc, ws, p128);
    hipLaunchKernelGGL((predict_collect_big_kernel<NS, D>), dim3((unsigned)((P + 63) / 64)), dim3(64), 0, stream, gc, ws,
                       (int64_t)P, p128, row_tiles * 2, mean, var, jac);
    return check_launch();
}

// does the fused kernel's LDS budget hold Kstar for this model?
static bool fused_fits(int ns, int nu, int n_train, int n_pad, int H) {
    const int nw = kRolloutThreads / 64;
    const size_t lds = (gp_tile_lds_doubles(ns, ns + nu, n_train, n_pad, nw) + (size_t)SX_TILE * H * nu) * sizeof(double);
    return lds <= kMaxLdsBytes && n_pad <= 1024;
}

template <int NS, int NU>
static int launch_rollout_big(const sx_gp_model* m, const sx_env* env, const RolloutPtrs& rp, double* workspace,
                              int64_t workspace_bytes, hipStream_t stream) {
    constexpr int D = NS + NU;
    auto gc = make_gp_const<NS, NU>(m, kRolloutThreads / 64);
    ReachConst<NS, NU> rc;
    if (!make_reach_const<NS, NU>(env, rc)) return SX_ERR_ARG;
    CostConst<SX_MAX_M, NS, NU> cc;
    make_cost_const<NS, NU>(env, cc);
    const int64_t total = (int64_t)rp.E * rp.P;
    const int64_t p128 = (total + kBigTile - 1) / kBigTile * kBigTile;
    BigWs ws = big_ws_layout(workspace, NS, D, m->n_pad, total);
    if (!workspace || workspace_bytes < ws.total * (int64_t)sizeof(double)) return SX_ERR_ARG;
    const int row_tiles = (m->n_pad + kBigTile - 1) / kBigTile;
    BigInit bi{rp.x0, rp.q0, rp.mean, rp.std, rp.noise, rp.actions, rp.obj_cost, rp.con_cost, rp.P, rp.H};
    hipLaunchKernelGGL((init_big_kernel<NS, NU>), dim3((unsigned)((p128 + 255) / 256)), dim3(256), 0, stream, bi, ws, total,
                       p128);
    for (int t = 0; t < rp.H; ++t) {
        hipLaunchKernelGGL((kstar_big_kernel<NS, D>), dim3((unsigned)(p128 / 16), (unsigned)((m->n_pad + 255) / 256)),
                           dim3(256), 0, stream, gc, ws);
        hipLaunchKernelGGL((trmm_reduce_kernel<NS, D>), dim3((unsigned)(p128 / kBigTile), (unsigned)row_tiles, NS),
                           dim3(kBigThreads), 0, stream, gc, ws, p128);
        BigStep bs{rp.actions, rp.traj, rp.sigma, rp.obj_cost, rp.con_cost, rp.status, rp.H, t, row_tiles * 2,
                   (t > 0 || rp.q0 != nullptr) ? 1 : 0};
        hipLaunchKernelGGL((step_big_kernel<NS, NU>), dim3((unsigned)((total + 63) / 64)), dim3(64), 0, stream, gc, rc, cc, bs,
                           ws, total, p128);
    }
    return check_launch();
}

template <int NS, int NU>
static int launch_rollout(const sx_gp_model* m, const sx_env* env, const RolloutPtrs& rp, double* workspace,
                          int64_t workspace_bytes, hipStream_t stream) {
    if (!fused_fits(NS, NU, m->n_train, m->n_pad, rp.H))
        return launch_rollout_big<NS, NU>(m, env, rp, workspace, workspace_bytes, stream);
    const int nw = kRolloutThreads / 64;
    auto gc = make_gp_const<NS, NU>(m, nw);
    ReachConst<NS, NU> rc;
    if (!make_reach_const<NS, NU>(env, rc)) return SX_ERR_ARG;
    CostConst<SX_MAX_M, NS, NU> cc;
    make_cost_const<NS, NU>(env, cc);
    const size_t lds =
        (gp_tile_lds_doubles(NS, NS + NU, m->n_train, m->n_pad, nw) + (size_t)SX_TILE * rp.H * NU) * sizeof(double);
    if (int r = allow_lds(cem_rollout_kernel<NS, NU>, lds)) return r;
    const int tiles = (rp.P + SX_TILE - 1) / SX_TILE;
    hipLaunchKernelGGL((cem_rollout_kernel<NS, NU>), dim3(rp.E * tiles), dim3(kRolloutThreads), lds, stream, gc,
                       gc.stage_tab, rc, cc, rp);
    return check_launch();
}

}  // namespace sx

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
#define SX_DISPATCH(ns, nu, CALL)                              \
    do {                                                       \
        if ((ns) == 2 && (nu) == 1) return CALL(2, 1);         \
        if ((ns) == 4 && (nu) == 1) return CALL(4, 1);         \
        if ((ns) == 2 && (nu) == 2) return CALL(2, 2);         \
        if ((ns) == 4 && (nu) == 2) return CALL(4, 2);         \
        if ((ns) == 3 && (nu) == 1) return CALL(3, 1);         \
        if ((ns) == 1 && (nu) == 1) return CALL(1, 1);         \
        return SX_ERR_UNSUPPORTED;                             \
    } while (0)

extern "C" {

#ifdef SX_STAMPS
int sx_debug_set_stamps(unsigned long long* dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(sx::g_stamp_buf), &dev_buf, sizeof(dev_buf)) == hipSuccess ? SX_OK : SX_ERR_LAUNCH;
}
#endif

const char* sx_version(void) { return "sxamd 0.1 gfx950"; }

int sx_gp_pack_sizes(int n_s, int n_u, int n_train, int64_t* a_doubles, int64_t* tab_ints) {
    if (n_s <= 0 || n_s > SX_MAX_NS || n_u <= 0 || n_u > SX_MAX_NU || n_train <= 0) return SX_ERR_ARG;
    const int n_pad = sx::gp_n_pad(n_train, n_s + n_u);
    if (a_doubles) *a_doubles = sx::a_pack_doubles(n_s, n_pad);
    if (tab_ints) *tab_ints = sx::gp_stage_tab_ints(n_s, n_pad, SX_WAVES);
    return SX_OK;
}

int sx_gp_fit(const sx_gp_model* model, const double* y_train, double* work, double* linv, double* alpha,
              double* logdet, int32_t* status, void* stream) {
    if (!model || !model->x_train || !y_train || !work || !linv || !alpha || !logdet || !status) return SX_ERR_ARG;
    if (model->n_s <= 0 || model->n_s > SX_MAX_NS || model->n_u <= 0 || model->n_u > SX_MAX_NU || model->n_train <= 0)
        return SX_ERR_ARG;
    if (model->n_train > sx::kFitMaxN) return SX_ERR_UNSUPPORTED;
    sx::FitArgs fa;
    std::memset(&fa, 0, sizeof(fa));
    const int D = model->n_s + model->n_u;
    for (int i = 0; i < model->n_s * D; ++i) fa.inv_ls2[i] = model->inv_ls2[i];
    for (int i = 0; i < model->n_s; ++i) {
        fa.outputscale[i] = model->outputscale[i];
        fa.noise[i] = model->noise[i];
    }
    fa.x = model->x_train;
    fa.y = y_train;
    fa.lmat = work;
    fa.linv = linv;
    fa.alpha = alpha;
    fa.logdet = logdet;
    fa.status = status;
    fa.n = model->n_train;
    fa.D = D;
    fa.n_s = model->n_s;
    // panel width: as wide as the LDS left beside the static arrays allows (vec 32 KB + red 8 KB), at most 32
    const size_t lds_budget = 112 * 1024;
    int nb = (int)(lds_budget / (sizeof(double) * (size_t)fa.n));
    nb = nb > 32 ? 32 : (nb < 1 ? 1 : nb);
    fa.panel_cols = nb;
    const size_t lds = sizeof(double) * (size_t)fa.n * nb;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(sx::gp_fit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return SX_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(sx::gp_fit_kernel, dim3(model->n_s), dim3(sx::kFitThreads), lds, (hipStream_t)stream, fa);
    return sx::check_launch();
}

int sx_gp_mll_grad(const sx_gp_model* model, const double* y_train, const double* linv, const double* alpha,
                   const double* logdet, double* mll, double* grad, void* stream) {
    if (!model || !model->x_train || !y_train || !linv || !alpha || !logdet || !mll || !grad) return SX_ERR_ARG;
    if (model->n_s <= 0 || model->n_s > SX_MAX_NS || model->n_u <= 0 || model->n_u > SX_MAX_NU || model->n_train <= 0)
        return SX_ERR_ARG;
    sx::MllArgs ma;
    std::memset(&ma, 0, sizeof(ma));
    const int D = model->n_s + model->n_u;
    for (int i = 0; i < model->n_s * D; ++i) ma.inv_ls2[i] = model->inv_ls2[i];
    for (int i = 0; i < model->n_s; ++i) {
        ma.outputscale[i] = model->outputscale[i];
        ma.noise[i] = model->noise[i];
    }
    ma.x = model->x_train;
    ma.y = y_train;
    ma.linv = linv;
    ma.alpha = alpha;
    ma.logdet = logdet;
    ma.mll = mll;
    ma.grad = grad;
    ma.n = model->n_train;
    ma.D = D;
    ma.n_s = model->n_s;
    hipLaunchKernelGGL(sx::gp_mll_grad_kernel, dim3(model->n_s), dim3(sx::kFitThreads), 0, (hipStream_t)stream, ma);
    return sx::check_launch();
}

int sx_gp_pack(sx_gp_model* model, const double* linv, const double* alpha, void* stream) {
    if (!model || !linv || !alpha || !model->x_train || !model->a_pack || !model->stage_tab) return SX_ERR_ARG;
    if (model->n_s <= 0 || model->n_s > SX_MAX_NS || model->n_u <= 0 || model->n_u > SX_MAX_NU || model->n_train <= 0)
        return SX_ERR_ARG;
    const int D = model->n_s + model->n_u;
    model->n_pad = sx::gp_n_pad(model->n_train, D);
    const bool has_tab = model->n_pad <= 1024;  // beyond that only the large-training-set path runs (no stage table)
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = sx::a_pack_doubles(model->n_s, model->n_pad);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    sx::PackArgs<SX_MAX_NS, SX_MAX_D> args;
    std::memset(&args, 0, sizeof(args));
    for (int i = 0; i < model->n_s * D; ++i) args.inv_ls2[i] = model->inv_ls2[i];
    hipLaunchKernelGGL(sx::pack_a_kernel, dim3(grid), dim3(256), 0, s, linv, alpha, model->x_train, args, model->n_s, D,
                       model->n_train, model->n_pad, const_cast<double*>(model->a_pack));
    if (has_tab)
        hipLaunchKernelGGL(sx::build_stage_tab_kernel, dim3(1), dim3(64), 0, s,
                           reinterpret_cast<int4*>(const_cast<int32_t*>(model->stage_tab)), model->n_s, model->n_train,
                           model->n_pad, SX_WAVES, sx::gp_stage_cap(model->n_s, model->n_pad, SX_WAVES));
    return sx::check_launch();
}

int64_t sx_gp_predict_workspace_bytes(const sx_gp_model* model, int P) {
    if (!model || P < 0) return -1;
    if (sx::predict_fits(model->n_s, model->n_u, model->n_train, model->n_pad)) return 0;
    return sx::big_ws_layout(nullptr, model->n_s, model->n_s + model->n_u, model->n_pad, P).total * (int64_t)sizeof(double);
}

int sx_gp_predict(const sx_gp_model* model, const double* z, int P, double* mean, double* var, double* jac,
                  void* workspace, int64_t workspace_bytes, void* stream) {
    if (!model || P < 0) return SX_ERR_ARG;
    if (P == 0) return SX_OK;   // an empty batch is not an error (its pointers may be NULL)
    if (!z || !mean || !var) return SX_ERR_ARG;
#define CALL(NS, NU) \
    sx::launch_predict<NS, NU>(model, z, P, mean, var, jac, (double*)workspace, workspace_bytes, (hipStream_t)stream)
    SX_DISPATCH(model->n_s, model->n_u, CALL);
#undef CALL
}

int sx_onestep_reach(const sx_env* env, int P, const double* p, const double* Q, const double* u, const double* mean,
                     const double* var, const double* jac, double* p1, double* Q1, double* sigma, int32_t* status,
                     void* stream) {
    if (!env || !p || !u || !mean || !var || !p1 || !Q1 || !sigma || !status || P < 0) return SX_ERR_ARG;
    if (Q && !jac) return SX_ERR_ARG;
    if (P == 0) return SX_OK;
#define CALL(NS, NU) \
    sx::launch_reach<NS, NU>(env, P, p, Q, u, mean, var, jac, p1, Q1, sigma, status, (hipStream_t)stream)
    SX_DISPATCH(env->n_s, env->n_u, CALL);
#undef CALL
}

int sx_polytope_distance(const sx_env* env, int P, const double* p, const double* Q, double c_safety, double* d,
                         uint8_t* inside, void* stream) {
    if (!env || !p || !Q || !d || P < 0) return SX_ERR_ARG;
    if (env->m <= 0 || env->m > SX_MAX_M) return SX_ERR_UNSUPPORTED;
    if (P == 0) return SX_OK;
    switch (env->n_s) {
        case 1: return sx::launch_polytope<1>(env, P, p, Q, c_safety, d, inside, (hipStream_t)stream);
        case 2: return sx::launch_polytope<2>(env, P, p, Q, c_safety, d, inside, (hipStream_t)stream);
        case 3: return sx::launch_polytope<3>(env, P, p, Q, c_safety, d, inside, (hipStream_t)stream);
        case 4: return sx::launch_polytope<4>(env, P, p, Q, c_safety, d, inside, (hipStream_t)stream);
        default: return SX_ERR_UNSUPPORTED;
    }
}

int64_t sx_cem_rollout_workspace_bytes(const sx_gp_model* model, int E, int P, int H) {
    if (!model || E <= 0 || P <= 0 || H <= 0) return -1;
    if (sx::fused_fits(model->n_s, model->n_u, model->n_train, model->n_pad, H)) return 0;
    return sx::big_ws_layout(nullptr, model->n_s, model->n_s + model->n_u, model->n_pad, (int64_t)E * P).total *
           (int64_t)sizeof(double);
}

int sx_cem_rollout(const sx_gp_model* model, const sx_env* env, int E, int P, int H, const double* x0, const double* q0,
                   const double* mean, const double* std, const double* noise, double* actions, double* traj,
                   double* sigma, double* obj_cost, double* con_cost, int32_t* status, void* workspace,
                   int64_t workspace_bytes, void* stream) {
    if (!model || !env || !x0 || !actions || !obj_cost || !con_cost || !status) return SX_ERR_ARG;
    if (E <= 0 || P <= 0 || H <= 0) return SX_ERR_ARG;
    if (noise && (!mean || !std)) return SX_ERR_ARG;
    if (model->n_s != env->n_s || model->n_u != env->n_u) return SX_ERR_ARG;
    if (env->m <= 0 || env->m > SX_MAX_M) return SX_ERR_UNSUPPORTED;
    sx::RolloutPtrs rp{x0, q0, mean, std, noise, actions, traj, sigma, obj_cost, con_cost, status, E, P, H};
#define CALL(NS, NU) \
    sx::launch_rollout<NS, NU>(model, env, rp, (double*)workspace, workspace_bytes, (hipStream_t)stream)
    SX_DISPATCH(model->n_s, model->n_u, CALL);
#undef CALL
}

int sx_cem_rank_refit(int E, int P, int k, int row_len, const double* con_cost, const double* obj_cost,
                      int64_t cost_stride, const double* actions, int64_t act_stride, int32_t* elite_idx,
                      double* elite_rows, double* mean, double* std, double* best, int32_t* best_ok, void* stream) {
    if (!con_cost || !obj_cost || !actions || E <= 0 || P <= 0 || k <= 0 || row_len <= 0) return SX_ERR_ARG;
    if (k > P) return SX_ERR_ARG;
    if (k > sx::kRankMaxK) return SX_ERR_UNSUPPORTED;
    sx::RankArgs ra{P,      k,          row_len,    con_cost, obj_cost, (long long)cost_stride,
                    actions, (long long)act_stride, elite_idx, elite_rows, mean,     std,
                    best,   best_ok};
    if (P > sx::kRankThreads * sx::kRankSlots) return SX_ERR_UNSUPPORTED;
    const int slots = (P + sx::kRankThreads - 1) / sx::kRankThreads;
    if (slots <= 4)
        hipLaunchKernelGGL(sx::cem_rank_kernel<4>, dim3(E), dim3(sx::kRankThreads), 0, (hipStream_t)stream, ra);
    else if (slots <= 8)
        hipLaunchKernelGGL(sx::cem_rank_kernel<8>, dim3(E), dim3(sx::kRankThreads), 0, (hipStream_t)stream, ra);
    else
        hipLaunchKernelGGL(sx::cem_rank_kernel<sx::kRankSlots>, dim3(E), dim3(sx::kRankThreads), 0, (hipStream_t)stream, ra);
    return sx::check_launch();
}

}  // extern "C"
