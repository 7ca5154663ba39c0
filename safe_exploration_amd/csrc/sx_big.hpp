// Large training sets (BASELINE config 4: cart-pole, N = 2000): Kstar does not fit in LDS, so a rollout step becomes
// three launches over ALL particles instead of one fused kernel:
//   1. kstar_big_kernel       Kstar_d -> HBM, in MFMA B-fragment order                       (VALU, HBM-write bound)
//   2. trmm_reduce_kernel     T = W_d . Kstar_d^T on the f64 matrix cores, 128 x 128 tiles staged through LDS, with the
//                             sum of squares over rows (and the mean/Jacobian rows) taken in the epilogue  (MFMA bound)
//   3. step_big_kernel        one particle per lane: variance/mean/Jacobian assembly, reachability step, costs
// The partial sums of squares are written per (row tile, wave row) and added in a fixed order: results do not depend
// on scheduling.
#pragma once
#include <hip/hip_runtime.h>

#include "sx_gp.hpp"
#include "sx_reach.hpp"

namespace sx {

constexpr int kBigTile = 128;          // rows of W and particles per workgroup tile
constexpr int kBigRb = kBigTile / 16;  // row-blocks / particle tiles per workgroup tile
constexpr int kBigThreads = 256;

// Workspace carve-up (doubles).  P16 = particles rounded up to 16, P128 to 128.
struct BigWs {
    double* ks;       // [NS][P128/16][n_pad/8][128]   Kstar fragments
    double* part;     // [NS][row_tiles*2][P128]       partial sums of squares
    double* mj;       // [NS][D+1][P128]               mean / Jacobian rows
    double* zs;       // [P128][D]                     query points of the current step
    double* pst;      // [P128][NS]                    ellipsoid centres
    double* qst;      // [P128][NS][NS]                shape matrices
    int64_t total;
};

inline BigWs big_ws_layout(double* base, int ns, int d_in, int n_pad, int64_t particles) {
    const int64_t p128 = (particles + kBigTile - 1) / kBigTile * kBigTile;
    const int row_tiles = (n_pad + kBigTile - 1) / kBigTile;
    BigWs w;
    int64_t off = 0;
    auto take = [&](int64_t n) {
        double* p = base ? base + off : nullptr;
        off += (n + 1) & ~int64_t(1);
        return p;
    };
    w.ks = take((int64_t)ns * (p128 / 16) * (n_pad / 8) * 128);
    w.part = take((int64_t)ns * row_tiles * 2 * p128);
    w.mj = take((int64_t)ns * (d_in + 1) * p128);
    w.zs = take(p128 * d_in);
    w.pst = take(p128 * ns);
    w.qst = take(p128 * ns * ns);
    w.total = off;
    return w;
}

// ---- 0. init: sample the actions, set the start state and the first query point --------------------------------
struct BigInit {
    const double* x0;
    const double* q0;
    const double* mean;
    const double* std;
    const double* noise;
    double* actions;
    double* obj;
    double* con;
    int P, H;   // P = particles per problem; total = E * P
};

template <int NS, int NU>
__global__ void init_big_kernel(BigInit bi, BigWs ws, int64_t total, int64_t p128) {
    constexpr int D = NS + NU;
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= p128) return;
    const bool valid = g < total;
    const int64_t e = valid ? g / bi.P : 0;
    if (valid) {
        for (int r = 0; r < bi.H * NU; ++r) {
            const int64_t gi = g * (bi.H * NU) + r;
            if (bi.noise) bi.actions[gi] = bi.mean[e * bi.H * NU + r] + bi.std[e * bi.H * NU + r] * bi.noise[gi];
        }
        bi.obj[g] = 0.0;
        bi.con[g] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double pi = valid ? bi.x0[e * NS + i] : 0.0;
        ws.pst[g * NS + i] = pi;
        ws.zs[g * D + i] = pi;
#pragma unroll
        for (int j = 0; j < NS; ++j) ws.qst[(g * NS + i) * NS + j] = (valid && bi.q0) ? bi.q0[(e * NS + i) * NS + j] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < NU; ++c) ws.zs[g * D + NS + c] = valid ? bi.actions[g * (bi.H * NU) + c] : 0.0;
}

// ---- 1. Kstar -> HBM in fragment order ---------------------------------------------------------------------------
// grid (P128 / 16, k chunks of 256); thread = (query point c = tid & 15, k = chunk * 256 + (tid >> 4) + 16 i)
template <int NS, int D>
__global__ __launch_bounds__(256) void kstar_big_kernel(GpConst<NS, D> gc, BigWs ws) {
    const int tile = blockIdx.x, c = threadIdx.x & 15;
    // the table-driven exp of the fused path (sx_gp.hpp): 2^(j/256) in LDS, behind it the NaN table of a NaN query
    __shared__ double etab_s[2 * kExpTab];
    etab_s[threadIdx.x] = kExp2Tab[threadIdx.x];
    etab_s[kExpTab + threadIdx.x] = __builtin_nan("");
    static_assert(kExpTab == 256, "one table entry per thread of this kernel");
    double z[D];
    bool znan = false;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        z[j] = ws.zs[((int64_t)tile * 16 + c) * D + j];
        znan = znan || (z[j] != z[j]);
    }
    __syncthreads();
    const lds_f64* etab = (const lds_f64*)etab_s + (znan ? kExpTab : 0);
    const int64_t tstride = (int64_t)(gc.n_pad >> 3) * 128;
    const int64_t dstride = (int64_t)gridDim.x * tstride;
    const int kbase = blockIdx.y * 256;
    for (int i = 0; i < 16; i += 2) {
        double arg[2 * NS], val[2 * NS];
        int ks[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = kbase + (threadIdx.x >> 4) + 16 * (i + h);
            ks[h] = k;
            const int kk = k < gc.n_train ? k : 0;
            double sq[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double df = z[j] - gc.x_train[(int64_t)kk * D + j];
                sq[j] = df * df;
            }
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                double a = gc.k_log_os[d];   // exponent in units of ln 2 / 256
#pragma unroll
                for (int j = 0; j < D; ++j) a = fma(sq[j], gc.k_nh_ils2[d * D + j], a);
                arg[h * NS + d] = a;
            }
        }
        exp_tab_f64_n<2 * NS>(arg, val, etab);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (ks[h] < gc.n_pad) {
                const int fi = frag_index(c, ks[h]);
#pragma unroll
                for (int d = 0; d < NS; ++d)
                    ws.ks[d * dstride + tile * tstride + fi] = (ks[h] < gc.n_train) ? val[h * NS + d] : 0.0;
            }
        }
    }
}

// ---- 2. triangular product + row reduction ---------------------------------------------------------------------------
// grid (P128 / 128, row tiles, NS); 4 waves: wave w owns row-blocks 4 (w >> 1) .. +3 and particle tiles 4 (w & 1) .. +3
template <int NS, int D>
__global__ __launch_bounds__(kBigThreads) void trmm_reduce_kernel(GpConst<NS, D> gc, BigWs ws, int64_t p128) {
    __shared__ __attribute__((aligned(16))) v2d sA[kBigRb][2][64];
    __shared__ __attribute__((aligned(16))) v2d sB[kBigRb][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int pg = blockIdx.x, rt = blockIdx.y, d = blockIdx.z;
    const int nrb = gc.n_pad >> 4;
    const int rb0 = rt * kBigRb;
    const int rb_end = (rb0 + kBigRb < nrb) ? rb0 + kBigRb : nrb;   // exclusive
    const int npairs = 2 * rb_end;                                   // K extent of the tile's longest row-block
    const int64_t wpo = w_pairs_per_output(nrb);
    const v2d* apack = reinterpret_cast<const v2d*>(gc.a_pack) + (int64_t)d * wpo * 64;
    const int64_t tstride = (int64_t)(gc.n_pad >> 3) * 64;           // v2d per particle tile
    const v2d* ks = reinterpret_cast<const v2d*>(ws.ks) + ((int64_t)d * (p128 / 16) + (int64_t)pg * kBigRb) * tstride;

    v4d acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};

    // Global -> register -> LDS staging, one chunk (2 fragment pairs = 16 k) ahead: the loads of chunk c + 1 are in
    // flight while the 32 MFMAs per wave of chunk c run.  A chunk is 8 row-blocks x 2 pairs of W and 8 particle tiles
    // x 2 pairs of Kstar: 2 x 1024 v2d, 4 + 4 per thread.
    v2d ra[4], rb_[4];
    auto fetch = [&](int q0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * kBigThreads + tid;   // 0 .. 1023
            const int blk = idx >> 7, pr = (idx >> 6) & 1, ln = idx & 63;
            const int rb = rb0 + blk, q = q0 + pr;
            ra[it] = v2d{0.0, 0.0};
            if (rb < rb_end && q < 2 * (rb + 1)) ra[it] = apack[((int64_t)rb * (rb + 1) + q) * 64 + ln];
            rb_[it] = ks[(int64_t)blk * tstride + (int64_t)q * 64 + ln];
        }
    };
    fetch(0);
    for (int q0 = 0; q0 < npairs; q0 += 2) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * kBigThreads + tid;
            const int blk = idx >> 7, pr = (idx >> 6) & 1, ln = idx & 63;
            sA[blk][pr][ln] = ra[it];
            sB[blk][pr][ln] = rb_[it];
        }
        __syncthreads();
        if (q0 + 2 < npairs) fetch(q0 + 2);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            v2d a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = sA[4 * wr + m][pr][lane];
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = sB[4 * wc + n][pr][lane];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                // beyond the diagonal of row-block rb0 + 4 wr + m its W fragments are zero padding: nothing to add
                // (wave-uniform; only the last chunks of a tile's K extent are affected)
                if (q0 + pr >= 2 * (rb0 + 4 * wr + m + 1)) continue;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[n].x, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b[n].y, acc[m][n], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // epilogue: rows < N are squared and summed, rows N .. N + D are the mean / Jacobian rows
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int64_t p = ((int64_t)pg * kBigRb + 4 * wc + n) * 16 + (lane & 15);
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row0 = (rb0 + 4 * wr + m) * 16 + (lane >> 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * r;
                const double v = acc[m][n][r];
                if (row < gc.n_train)
                    s = fma(v, v, s);
                else if (row - gc.n_train <= D)
                    ws.mj[((int64_t)d * (D + 1) + (row - gc.n_train)) * p128 + p] = v;
            }
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lane < 16) ws.part[((int64_t)d * gridDim.y * 2 + rt * 2 + wr) * p128 + p] = s;
    }
}

// ---- 3. per-particle step ------------------------------------------------------------------------------------------
struct BigStep {
    const double* actions;
    double* traj;
    double* sigma;
    double* obj;
    double* con;
    int* status;
    int H, t, row_parts;   // row_parts = row_tiles * 2
    int have_q;
};

template <int NS, int NU>
__global__ void step_big_kernel(GpConst<NS, NS + NU> gc, ReachConst<NS, NU> rc, CostConst<SX_MAX_M, NS, NU> cc,
                                BigStep bs, BigWs ws, int64_t total, int64_t p128) {
    constexpr int D = NS + NU;
    constexpr int S = NS + NS * NS;
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= total) return;
    double p[NS], Q[NS][NS], z[D], u[NU], mean[NS], var[NS], jac[NS][D], p1[NS], Q1[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        p[i] = ws.pst[g * NS + i];
#pragma unroll
        for (int j = 0; j < NS; ++j) Q[i][j] = ws.qst[(g * NS + i) * NS + j];
    }
#pragma unroll
    for (int j = 0; j < D; ++j) z[j] = ws.zs[g * D + j];
#pragma unroll
    for (int c = 0; c < NU; ++c) u[c] = z[NS + c];
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        double q = 0.0;
        for (int r = 0; r < bs.row_parts; ++r) q += ws.part[((int64_t)d * bs.row_parts + r) * p128 + g];
        var[d] = (gc.outputscale[d] - q) + gc.noise[d];
        const double m = ws.mj[((int64_t)d * (D + 1)) * p128 + g];
        mean[d] = m;
#pragma unroll
        for (int j = 0; j < D; ++j)
            jac[d][j] = ws.mj[((int64_t)d * (D + 1) + 1 + j) * p128 + g] - z[j] * gc.inv_ls2[d * D + j] * m;
    }
    int st = 0;
    if (bs.have_q)
        reach_ellipsoid<NS, NU>(rc, p, Q, u, mean, var, jac, p1, Q1, st);
    else
        reach_point<NS, NU>(rc, p, u, mean, var, p1, Q1, st);
    double obj = bs.obj[g] + objective_cost<SX_MAX_M, NS, NU>(cc, p1, var);
    double con = bs.con[g];
    bool uviol = false;
#pragma unroll
    for (int c = 0; c < NU; ++c) uviol = uviol || (u[c] < cc.u_min[c]) || (u[c] > cc.u_max[c]);
    if (uviol) con += SX_ACTION_VIOLATION_COST;
    if (cc.con_mode == SX_CON_ALL_STATES || bs.t == bs.H - 1) {
        if (polytope_violated<SX_MAX_M, NS>(cc.h_mat, cc.h_vec, cc.m, 1.0, p1, Q1, nullptr)) con += SX_STATE_VIOLATION_COST;
    }
    bs.obj[g] = obj;
    bs.con[g] = con;
    if (bs.traj) {
        double* tr = bs.traj + (g * bs.H + bs.t) * S;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            tr[i] = p1[i];
#pragma unroll
            for (int j = 0; j < NS; ++j) tr[NS + i * NS + j] = Q1[i][j];
        }
    }
    if (bs.sigma) {
#pragma unroll
        for (int i = 0; i < NS; ++i) bs.sigma[(g * bs.H + bs.t) * NS + i] = var[i];
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        ws.pst[g * NS + i] = p1[i];
        ws.zs[g * D + i] = p1[i];
#pragma unroll
        for (int j = 0; j < NS; ++j) ws.qst[(g * NS + i) * NS + j] = Q1[i][j];
    }
    if (bs.t + 1 < bs.H) {
#pragma unroll
        for (int c = 0; c < NU; ++c) ws.zs[g * D + NS + c] = bs.actions[g * (bs.H * NU) + (bs.t + 1) * NU + c];
    }
    if (st) atomicOr(bs.status, st);
}

}  // namespace sx
