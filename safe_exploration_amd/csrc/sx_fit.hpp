// Warm path: sx_gp_fit / sx_gp_mll_grad / sx_gp_pack kernels (once per update_model or training iteration).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sx_amd.h"
#include "sx_gp.hpp"
#include "sx_reach.hpp"

namespace sx {

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
                // four independent partial sums: the loads of W (HBM/L2) overlap instead of queueing behind one
                // dependent accumulate chain
                double s0 = (c == i) ? 1.0 : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int kk = c;
                for (; kk + 4 <= i; kk += 4) {
                    s0 -= vec[kk] * W[(size_t)kk * n + c];
                    s1 -= vec[kk + 1] * W[(size_t)(kk + 1) * n + c];
                    s2 -= vec[kk + 2] * W[(size_t)(kk + 2) * n + c];
                    s3 -= vec[kk + 3] * W[(size_t)(kk + 3) * n + c];
                }
                for (; kk < i; ++kk) s0 -= vec[kk] * W[(size_t)kk * n + c];
                w = ((s0 + s1) + (s2 + s3)) * inv;
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
            double k0 = 0.0, k1 = 0.0, k2 = 0.0, k3 = 0.0;   // independent partial sums: loads overlap
            int r = i;
            for (; r + 4 <= n; r += 4) {
                k0 += W[(size_t)r * n + i] * W[(size_t)r * n + j];
                k1 += W[(size_t)(r + 1) * n + i] * W[(size_t)(r + 1) * n + j];
                k2 += W[(size_t)(r + 2) * n + i] * W[(size_t)(r + 2) * n + j];
                k3 += W[(size_t)(r + 3) * n + i] * W[(size_t)(r + 3) * n + j];
            }
            for (; r < n; ++r) k0 += W[(size_t)r * n + i] * W[(size_t)r * n + j];
            const double kinv = (k0 + k1) + (k2 + k3);
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

// the combined table (all outputs), then one table per output (sx_rollout.hpp, output-by-output mode)
__global__ void build_stage_tab_kernel(int4* tab, int ns, int n_train, int n_pad, int nw, int stage_cap, int stage_cap_one) {
    if ((int)threadIdx.x >= nw) return;
    gp_build_stage_tab(tab, ns, n_train, n_pad, nw, stage_cap, threadIdx.x);
    int4* one = tab + (size_t)nw * (1 + stage_cap);
    for (int d = 0; d < ns; ++d)
        gp_build_stage_tab(one + (size_t)d * nw * (1 + stage_cap_one), 1, n_train, n_pad, nw, stage_cap_one, threadIdx.x, d);
}

// ---------------------------------------------------------------------------------------------------------------
// sx_gp_predict_var_jac: d var_d / d z for the numpy StateSpaceModel adapter (the casadi-based solvers linearise the
// variance; the CEM path never needs it).  var_d = s_d + noise_d - k*^T (K_d + noise_d I)^-1 k*, so
//     d var_d / d z_j = 2 sum_i v_i k*_i (z_j - X_ij) / l_dj^2,      v = W_d^T (W_d k*).
// One workgroup per (query point, output); W_d streamed twice (rows for t = W k*, columns for v = W^T t).
// ---------------------------------------------------------------------------------------------------------------
struct VarJacArgs {
    double inv_ls2[SX_MAX_NS * SX_MAX_D];
    double outputscale[SX_MAX_NS];
    const double* x;     // [N x D]
    const double* linv;  // [n_s x N x N]
    const double* z;     // [P x D]
    double* jac_var;     // [P x n_s x D]
    int n, D, n_s;
};

__global__ __launch_bounds__(256) void gp_var_jac_kernel(VarJacArgs a) {
    extern __shared__ __attribute__((aligned(16))) double vj_smem[];
    double* ks = vj_smem;        // [N]  k*
    double* ts = vj_smem + a.n;  // [N]  t = W k*
    __shared__ double red[4][SX_MAX_D];
    const int p = blockIdx.x, d = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int n = a.n, D = a.D;
    const double* W = a.linv + (size_t)d * n * n;
    double z[SX_MAX_D], w2[SX_MAX_D];
    for (int j = 0; j < D; ++j) {
        z[j] = a.z[(size_t)p * D + j];
        w2[j] = a.inv_ls2[d * D + j];
    }
    for (int i = tid; i < n; i += blockDim.x) {
        double q = 0.0;
        for (int j = 0; j < D; ++j) {
            const double df = z[j] - a.x[(size_t)i * D + j];
            q += df * df * w2[j];
        }
        ks[i] = a.outputscale[d] * exp(-0.5 * q);
    }
    __syncthreads();
    for (int r = wave; r < n; r += nwaves) {  // a wave per row: coalesced along the row
        double s = 0.0;
        for (int i = lane; i <= r; i += 64) s += W[(size_t)r * n + i] * ks[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) ts[r] = s;
    }
    __syncthreads();
    double g[SX_MAX_D];
    for (int j = 0; j < D; ++j) g[j] = 0.0;
    for (int i = tid; i < n; i += blockDim.x) {  // a thread per column: coalesced across the threads
        double v = 0.0;
        for (int r = i; r < n; ++r) v += W[(size_t)r * n + i] * ts[r];
        const double w = v * ks[i];
        for (int j = 0; j < D; ++j) g[j] += w * (z[j] - a.x[(size_t)i * D + j]) * w2[j];
    }
    for (int j = 0; j < D; ++j) {
        double s = g[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) red[wave][j] = s;
    }
    __syncthreads();
    if (tid < D) {
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += red[w][tid];
        a.jac_var[((size_t)p * a.n_s + d) * D + tid] = 2.0 * s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sx_gp_predict_mean_hessian: d^2 mean_d / dz dz^T for the numpy StateSpaceModel adapter's linearize_predict (the
// reference differentiates gpytorch's mean twice with autograd: ssm_pytorch/gaussian_process.py:160-187).
//     mean_d(z) = sum_i alpha_di k_i(z),   k_i = s_d exp(-1/2 sum_j (z_j - X_ij)^2 / l_dj^2)
//     H_d[j][l] = sum_i alpha_di k_i [ (z_j - X_ij)(z_l - X_il) / (l_dj^2 l_dl^2) - delta_jl / l_dj^2 ]
// One workgroup per (query point, output); a thread per training point, D (D + 1) / 2 sums reduced through LDS in a fixed
// order (results do not depend on scheduling).
// ---------------------------------------------------------------------------------------------------------------
struct MeanHessArgs {
    double inv_ls2[SX_MAX_NS * SX_MAX_D];
    double outputscale[SX_MAX_NS];
    const double* x;      // [N x D]
    const double* alpha;  // [n_s x N]
    const double* z;      // [P x D]
    double* hess;         // [P x n_s x D x D]
    int n, D, n_s;
};

__global__ __launch_bounds__(256) void gp_mean_hessian_kernel(MeanHessArgs a) {
    constexpr int kPairs = SX_MAX_D * (SX_MAX_D + 1) / 2;
    __shared__ double red[4][kPairs];
    const int p = blockIdx.x, d = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n, D = a.D;
    double z[SX_MAX_D], w2[SX_MAX_D], acc[kPairs];
    for (int j = 0; j < D; ++j) {
        z[j] = a.z[(size_t)p * D + j];
        w2[j] = a.inv_ls2[d * D + j];
    }
    for (int c = 0; c < kPairs; ++c) acc[c] = 0.0;
    for (int i = tid; i < n; i += blockDim.x) {
        double q = 0.0, g[SX_MAX_D];
        for (int j = 0; j < D; ++j) {
            const double df = z[j] - a.x[(size_t)i * D + j];
            q += df * df * w2[j];
            g[j] = df * w2[j];
        }
        const double ak = a.alpha[(size_t)d * n + i] * a.outputscale[d] * exp(-0.5 * q);
        int c = 0;
        for (int j = 0; j < D; ++j)
            for (int l = j; l < D; ++l, ++c) acc[c] += ak * (g[j] * g[l] - (j == l ? w2[j] : 0.0));
    }
    const int npairs = D * (D + 1) / 2;
    for (int c = 0; c < npairs; ++c) {
        double s = acc[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) red[wave][c] = s;
    }
    __syncthreads();
    if (tid < npairs) {
        const double s = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        int c = 0;
        for (int j = 0; j < D; ++j)
            for (int l = j; l < D; ++l, ++c)
                if (c == tid) {
                    double* h = a.hess + ((size_t)p * a.n_s + d) * D * D;
                    h[j * D + l] = s;
                    h[l * D + j] = s;
                }
    }
}

}  // namespace sx
