// Blocked, multi-workgroup sx_gp_fit for training sets beyond a few hundred points (BASELINE config 4: N = 2000).
//
//   K_d + noise_d I = L_d L_d^T,   W_d = L_d^-1,   alpha_d = W_d^T W_d y_d,   sum log diag L_d
//
// in 64 x 64 blocks on the f64 matrix cores (v_mfma_f64_16x16x4_f64), all outputs d side by side in the grid:
//   kmat_kernel                      K (lower blocks) into `lmat`, zeros into the strictly upper blocks of `linv`
//   per block column p (right-looking Cholesky):
//     potrf_diag_kernel              L_pp = chol(A_pp) in LDS; L_pp^-1 straight into W_pp (it IS the diagonal block of W)
//     trsm_kernel                    L_tp = A_tp W_pp^T                        for every block row t > p
//     syrk_kernel                    A_ts -= L_tp L_sp^T                       for every block pair t >= s > p
//   trtri_kernel                     block column j of W by forward substitution, one workgroup per (j, d):
//                                    W_ij = -W_ii sum_{k=j}^{i-1} L_ik W_kj
//   alpha_logdet_kernel              alpha = W^T (W y), sum log diag L (one workgroup per output, fixed summation order)
// About 3 N / 64 + 3 launches, every one of them wide; the single-workgroup kernel in sx_fit.hpp stays for small N,
// where its one launch wins.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sx_amd.h"
#include "sx_gp.hpp"

namespace sx {

constexpr int kFB = 64;            // block edge
constexpr int kFLd = kFB + 1;      // LDS row stride of a staged block (odd: the 16 rows an MFMA operand touches spread over banks)
constexpr int kBlockedFitMinN = 96;  // at or below: the single-workgroup kernel (one launch) is as fast
constexpr int kFThreads = 256;     // 4 waves; wave w owns rows 16 w .. 16 w + 15 of the 64 x 64 result

struct BlockedFitArgs {
    double inv_ls2[SX_MAX_NS * SX_MAX_D];
    double outputscale[SX_MAX_NS];
    double noise[SX_MAX_NS];
    const double* x;   // [N x D]
    const double* y;   // [N x n_s]
    double* lmat;      // [n_s x N x N]  K then L (lower triangle)
    double* linv;      // [n_s x N x N]  W = L^-1
    double* alpha;     // [n_s x N]
    double* logdet;    // [n_s]
    int* status;
    int n, D, n_s, nblk;
};

// global (row-major, leading dimension n) block (bi, bj) -> LDS [64][kFLd]; outside the matrix: zero, or the identity
// when `unit_pad` (so that a padded diagonal block stays positive definite / invertible)
// (all 16 loads of a thread are issued before the first store: one memory round trip per block, not sixteen)
__device__ __forceinline__ void load_block(double* dst, const double* M, int n, int bi, int bj, bool unit_pad) {
    constexpr int kPer = kFB * kFB / kFThreads;
    double v[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int idx = threadIdx.x + u * kFThreads;
        const int r = idx >> 6, c = idx & 63;
        const int gr = bi * kFB + r, gc = bj * kFB + c;
        v[u] = (unit_pad && r == c) ? 1.0 : 0.0;
        if (gr < n && gc < n) v[u] = M[(size_t)gr * n + gc];
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int idx = threadIdx.x + u * kFThreads;
        dst[(idx >> 6) * kFLd + (idx & 63)] = v[u];
    }
}

// acc[c] (c = 0..3: column tiles) += A[16 w .. 16 w + 15][0..63] . B, 64-deep, operands staged in LDS.
//   NT: B[k][col] = Bs[col][k]  (C = A . Bs^T);   NN: B[k][col] = Bs[k][col]
//   AT: the left operand is staged transposed, A[row][k] = As[k][row]  (C = As^T . B)
template <bool NT, bool AT = false>
__device__ __forceinline__ void block_mma(v4d (&acc)[4], const double* As, const double* Bs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* arow = As + (16 * wave + (lane & 15)) * kFLd + (lane >> 4);
    const double* acol = As + (lane >> 4) * kFLd + 16 * wave + (lane & 15);
#pragma unroll 4
    for (int k0 = 0; k0 < kFB; k0 += 4) {
        const double a = AT ? acol[k0 * kFLd] : arow[k0];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double b = NT ? Bs[(16 * c + (lane & 15)) * kFLd + k0 + (lane >> 4)]
                                : Bs[(k0 + (lane >> 4)) * kFLd + 16 * c + (lane & 15)];
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
        }
    }
}

// the wave's 16 x 64 result strip: lane holds D[(lane >> 4) + 4 r][16 c + (lane & 15)]
template <class F>
__device__ __forceinline__ void for_each_result(const v4d (&acc)[4], F&& f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) f(16 * wave + (lane >> 4) + 4 * r, 16 * c + (lane & 15), acc[c][r]);
}

__global__ __launch_bounds__(kFThreads) void fit_kmat_kernel(BlockedFitArgs fa) {
    const int d = blockIdx.z, bi = blockIdx.y, bj = blockIdx.x, n = fa.n, D = fa.D;
    double* A = fa.lmat + (size_t)d * n * n;
    double* W = fa.linv + (size_t)d * n * n;
    for (int idx = threadIdx.x; idx < kFB * kFB; idx += kFThreads) {
        const int i = bi * kFB + (idx >> 6), j = bj * kFB + (idx & 63);
        if (i >= n || j >= n) continue;
        if (bj > bi) {
            W[(size_t)i * n + j] = 0.0;
        } else if (j <= i) {
            double q = 0.0;
            for (int c = 0; c < D; ++c) {
                const double df = fa.x[(size_t)i * D + c] - fa.x[(size_t)j * D + c];
                q += df * df * fa.inv_ls2[d * D + c];
            }
            A[(size_t)i * n + j] = fa.outputscale[d] * exp(-0.5 * q) + (i == j ? fa.noise[d] : 0.0);
        }
    }
}

// one workgroup per output: factor the 64 x 64 diagonal block p in LDS, invert the factor
__global__ __launch_bounds__(kFThreads) void fit_potrf_diag_kernel(BlockedFitArgs fa, int p) {
    __shared__ double L[kFB * kFLd];
    __shared__ double Wd[kFB * kFLd];
    const int d = blockIdx.x, tid = threadIdx.x, n = fa.n;
    double* A = fa.lmat + (size_t)d * n * n;
    double* W = fa.linv + (size_t)d * n * n;
    load_block(L, A, n, p, p, true);
    __syncthreads();
    bool bad = false;
    for (int j = 0; j < kFB; ++j) {
        const double ajj = L[j * kFLd + j];
        if (!(ajj > 0.0)) bad = true;
        const double piv = sqrt(ajj);
        __syncthreads();  // everyone has read the pivot
        if (tid >= j && tid < kFB) L[tid * kFLd + j] = (tid == j) ? piv : L[tid * kFLd + j] / piv;
        __syncthreads();
        // rank-1 update of the trailing lower triangle: rows r > j, columns j < c <= r
        for (int idx = ((j + 1) << 6) + tid; idx < kFB * kFB; idx += kFThreads) {
            const int r = idx >> 6, c = idx & 63;
            if (c > j && c <= r) L[r * kFLd + c] -= L[r * kFLd + j] * L[c * kFLd + j];
        }
        __syncthreads();
    }
    if (bad && tid == 0) atomicOr(fa.status, 8);
    // Wd = L^-1 by forward substitution, thread c solves column c and keeps it in registers (entries above the
    // diagonal are zero, so the sums may start at k = 0: static indices, fully unrolled, L read as LDS broadcasts)
    if (tid < kFB) {
        const int c = tid;
        double w[kFB];
#pragma unroll
        for (int r = 0; r < kFB; ++r) {
            double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; ++k) s -= L[r * kFLd + k] * w[k];
            w[r] = (r >= c) ? s / L[r * kFLd + r] : 0.0;
            Wd[r * kFLd + c] = w[r];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < kFB * kFB; idx += kFThreads) {
        const int r = idx >> 6, c = idx & 63;
        const int gr = p * kFB + r, gc = p * kFB + c;
        if (gr < n && gc < n) {
            if (c <= r) A[(size_t)gr * n + gc] = L[r * kFLd + c];
            W[(size_t)gr * n + gc] = Wd[r * kFLd + c];
        }
    }
}

// L_tp = A_tp W_pp^T for block rows t = p + 1 + blockIdx.x
__global__ __launch_bounds__(kFThreads) void fit_trsm_kernel(BlockedFitArgs fa, int p) {
    __shared__ double As[kFB * kFLd];
    __shared__ double Bs[kFB * kFLd];
    const int d = blockIdx.y, t = p + 1 + blockIdx.x, n = fa.n;
    double* A = fa.lmat + (size_t)d * n * n;
    const double* W = fa.linv + (size_t)d * n * n;
    load_block(As, A, n, t, p, false);
    load_block(Bs, W, n, p, p, true);
    __syncthreads();
    v4d acc[4] = {};
    block_mma<true>(acc, As, Bs);
    for_each_result(acc, [&](int r, int c, double v) {
        const int gr = t * kFB + r, gc = p * kFB + c;
        if (gr < n && gc < n) A[(size_t)gr * n + gc] = v;
    });
}

// A_ts -= L_tp L_sp^T for the trailing block pairs t >= s > p
__global__ __launch_bounds__(kFThreads) void fit_syrk_kernel(BlockedFitArgs fa, int p) {
    __shared__ double As[kFB * kFLd];
    __shared__ double Bs[kFB * kFLd];
    const int d = blockIdx.z, t = p + 1 + blockIdx.y, s = p + 1 + blockIdx.x, n = fa.n;
    if (s > t) return;
    double* A = fa.lmat + (size_t)d * n * n;
    load_block(As, A, n, t, p, false);
    load_block(Bs, A, n, s, p, false);
    __syncthreads();
    v4d acc[4] = {};
    block_mma<true>(acc, As, Bs);
    for_each_result(acc, [&](int r, int c, double v) {
        const int gr = t * kFB + r, gc = s * kFB + c;
        if (gr < n && gc < n && gc <= gr) A[(size_t)gr * n + gc] -= v;
    });
}

// block column j of W: W_jj is in place; W_ij = -W_ii sum_{k=j}^{i-1} L_ik W_kj for i > j.  The 64 columns of a block
// column are independent, so a workgroup takes a strip of 16 of them (blockIdx.z): four times the workgroups, a
// quarter of the serial chain each.
__device__ __forceinline__ void load_strip(double* dst, const double* M, int n, int bi, int bj, int strip) {
    // columns 16 strip .. 16 strip + 15 of block (bi, bj) -> dst[64][kFLd] (same column positions)
    constexpr int kPer = kFB * 16 / kFThreads;
    double v[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int idx = threadIdx.x + u * kFThreads;
        const int r = idx >> 4, c = 16 * strip + (idx & 15);
        const int gr = bi * kFB + r, gc = bj * kFB + c;
        v[u] = (gr < n && gc < n) ? M[(size_t)gr * n + gc] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int idx = threadIdx.x + u * kFThreads;
        dst[(idx >> 4) * kFLd + 16 * strip + (idx & 15)] = v[u];
    }
}

// acc += A[16 w .. 16 w + 15][0..63] . B[0..63][16 strip .. 16 strip + 15]   (NN)
__device__ __forceinline__ void strip_mma(v4d& acc, const double* As, const double* Bs, int strip) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* arow = As + (16 * wave + (lane & 15)) * kFLd + (lane >> 4);
    const double* bcol = Bs + (lane >> 4) * kFLd + 16 * strip + (lane & 15);
#pragma unroll 4
    for (int k0 = 0; k0 < kFB; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[k0], bcol[k0 * kFLd], acc, 0, 0, 0);
}

__global__ __launch_bounds__(kFThreads) void fit_trtri_kernel(BlockedFitArgs fa) {
    __shared__ double As[kFB * kFLd];
    __shared__ double Bs[kFB * kFLd];
    const int d = blockIdx.y, j = blockIdx.x, strip = blockIdx.z, n = fa.n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* A = fa.lmat + (size_t)d * n * n;
    double* W = fa.linv + (size_t)d * n * n;
    for (int i = j + 1; i < fa.nblk; ++i) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        for (int k = j; k < i; ++k) {
            __syncthreads();  // the previous product has been read out of LDS (and W_kj of the previous i is written)
            load_block(As, A, n, i, k, false);
            load_strip(Bs, W, n, k, j, strip);
            __syncthreads();
            strip_mma(acc, As, Bs, strip);
        }
        __syncthreads();
        load_block(As, W, n, i, i, false);
#pragma unroll
        for (int r = 0; r < 4; ++r) Bs[(16 * wave + (lane >> 4) + 4 * r) * kFLd + 16 * strip + (lane & 15)] = acc[r];
        __syncthreads();
        v4d out = {0.0, 0.0, 0.0, 0.0};
        strip_mma(out, As, Bs, strip);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = i * kFB + 16 * wave + (lane >> 4) + 4 * r, gc = j * kFB + 16 * strip + (lane & 15);
            if (gr < n && gc < n) W[(size_t)gr * n + gc] = -out[r];
        }
        __threadfence_block();
    }
}

// alpha = W^T (W y) and sum log diag L: one 1024-thread workgroup per output, t = W y kept in LDS (N <= 4096).
// Memory-bound on one CU (W is read twice: 64 MB per output at N = 2000, ~0.5 ms) -- a warm-path tail.
__global__ __launch_bounds__(1024) void fit_alpha_logdet_kernel(BlockedFitArgs fa) {
    extern __shared__ __attribute__((aligned(16))) double tvec[];  // [n]
    __shared__ double red[1024];
    const int d = blockIdx.x, tid = threadIdx.x, n = fa.n;
    const int lane = tid & 63, wave = tid >> 6;
    const double* A = fa.lmat + (size_t)d * n * n;
    const double* W = fa.linv + (size_t)d * n * n;
    for (int row = wave; row < n; row += 16) {  // a wave per row: coalesced along the row
        const double* wr = W + (size_t)row * n;
        double s = 0.0;
        for (int c = lane; c <= row; c += 64) s += wr[c] * fa.y[(size_t)c * fa.n_s + d];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) tvec[row] = s;
    }
    __syncthreads();
    for (int c = tid; c < n; c += 1024) {  // a thread per column: coalesced across the threads
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int i = c;
        for (; i + 4 <= n; i += 4) {
            s0 += W[(size_t)i * n + c] * tvec[i];
            s1 += W[(size_t)(i + 1) * n + c] * tvec[i + 1];
            s2 += W[(size_t)(i + 2) * n + c] * tvec[i + 2];
            s3 += W[(size_t)(i + 3) * n + c] * tvec[i + 3];
        }
        for (; i < n; ++i) s0 += W[(size_t)i * n + c] * tvec[i];
        fa.alpha[(size_t)d * n + c] = (s0 + s1) + (s2 + s3);
    }
    double ld = 0.0;
    for (int i = tid; i < n; i += 1024) ld += log(A[(size_t)i * n + i]);
    red[tid] = ld;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) fa.logdet[d] = red[0];
}

// ---------------------------------------------------------------------------------------------------------------
// Blocked sx_gp_mll_grad:  d mll / d theta = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta),  K^-1 = W^T W.
//   mll_pairs_kernel   one workgroup per block pair (bi >= bj) and output: the K^-1 block on the matrix cores
//                      (sum over kb >= bi of W[kb][bi]^T W[kb][bj]), contracted on the spot with dK/dtheta of its
//                      64 x 64 pairs; D + 2 partial sums per workgroup into `scratch`
//   mll_reduce_kernel  one workgroup per output: partial sums in fixed order (deterministic), y . alpha, the outputs
// ---------------------------------------------------------------------------------------------------------------
struct BlockedMllArgs {
    double inv_ls2[SX_MAX_NS * SX_MAX_D];
    double outputscale[SX_MAX_NS];
    const double* x;
    const double* y;
    const double* linv;
    const double* alpha;
    const double* logdet;
    double* scratch;   // [n_s x N x N] (only the first n_s x npairs x (D + 2) doubles of each output's slab are used)
    double* mll;       // [n_s]
    double* grad;      // [n_s x (D + 2)]
    int n, D, n_s, nblk;
};

__global__ __launch_bounds__(kFThreads) void mll_pairs_kernel(BlockedMllArgs ma) {
    __shared__ double As[kFB * kFLd];
    __shared__ double Bs[kFB * kFLd];
    __shared__ double red[kFThreads / 64][SX_MAX_D + 2];
    const int d = blockIdx.z, bi = blockIdx.y, bj = blockIdx.x, n = ma.n, D = ma.D;
    if (bj > bi) return;
    const double* W = ma.linv + (size_t)d * n * n;
    const double* al = ma.alpha + (size_t)d * n;
    v4d acc[4] = {};
    for (int kb = bi; kb < ma.nblk; ++kb) {   // W is lower triangular: W[kb][bi] vanishes for kb < bi
        __syncthreads();
        load_block(As, W, n, kb, bi, false);
        load_block(Bs, W, n, kb, bj, false);
        __syncthreads();
        block_mma<false, true>(acc, As, Bs);
    }
    double part[SX_MAX_D + 2];
#pragma unroll
    for (int c = 0; c < SX_MAX_D + 2; ++c) part[c] = 0.0;
    for_each_result(acc, [&](int r, int c, double kinv) {
        const int i = bi * kFB + r, j = bj * kFB + c;
        if (i >= n || j > i) return;
        const double g = al[i] * al[j] - kinv;
        double q = 0.0, dq[SX_MAX_D];
        for (int cc = 0; cc < D; ++cc) {
            const double df = ma.x[(size_t)i * D + cc] - ma.x[(size_t)j * D + cc];
            dq[cc] = df * df * ma.inv_ls2[d * D + cc];   // (x_ic - x_jc)^2 / l_c^2
            q += dq[cc];
        }
        const double kij = ma.outputscale[d] * exp(-0.5 * q);
        const double w = (i == j) ? 0.5 : 1.0;           // 1/2 tr(...) over the symmetric pair
        for (int cc = 0; cc < D; ++cc) part[cc] += w * g * kij * dq[cc];
        part[D] += w * g * kij;
        if (i == j) part[D + 1] += 0.5 * g;
    });
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = 0; c < D + 2; ++c) {
        double v = part[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[wave][c] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < D + 2) {
        const int pair = bi * (bi + 1) / 2 + bj;
        double v = 0.0;
        for (int w = 0; w < kFThreads / 64; ++w) v += red[w][threadIdx.x];
        ma.scratch[(size_t)d * n * n + (size_t)pair * (D + 2) + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(256) void mll_reduce_kernel(BlockedMllArgs ma) {
    __shared__ double red[256];
    const int d = blockIdx.x, tid = threadIdx.x, n = ma.n, D = ma.D;
    const int npairs = ma.nblk * (ma.nblk + 1) / 2;
    const double* part = ma.scratch + (size_t)d * n * n;
    for (int c = 0; c < D + 3; ++c) {
        double v = 0.0;
        if (c < D + 2) {
            for (int p = tid; p < npairs; p += 256) v += part[(size_t)p * (D + 2) + c];
        } else {
            for (int i = tid; i < n; i += 256) v += ma.y[(size_t)i * ma.n_s + d] * ma.alpha[(size_t)d * n + i];
        }
        red[tid] = v;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
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

}  // namespace sx
