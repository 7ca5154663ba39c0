// GP posterior for one tile of SX_TILE (=16) query points per workgroup, on the f64 matrix cores.
//
// For output d the predictive variance is  s_d + noise_d - || W_d k*_d ||^2  with  W_d = chol(K_d + noise_d I)^-1
// (lower triangular), so the dense work is the triangular product  T = W_d . Kstar_d^T  ([N x N] x [N x 16]) followed
// by a column sum of squares.  Mean and mean-Jacobian are 1 + D more rows of the same product:
//     R_d = [ alpha_d ; alpha_d * X_0 / l_d0^2 ; ... ]      mean = R_d[0] . k*,   J_j = R_d[1+j] . k* - z_j / l_dj^2 * mean
//
// v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md, "f64 MFMA does NOT use these maps"):
//     A: lane l holds A[row = l & 15][k = l >> 4]        -> W rows
//     B: lane l holds B[k = l >> 4][col = l & 15]        -> Kstar, col = query point
//     D: lane l holds D[row = (l >> 4) + 4 r][col = l & 15], r = 0..3
// so a lane keeps ONE query point and the sum of squares over rows is an in-lane sum plus one xor-16 / xor-32 fold.
//
// Fragment order in memory (both W in HBM/L2 and Kstar in LDS): k-blocks are stored in PAIRS so that one 16-byte
// access per lane (global_load_dwordx4 / ds_read_b128) feeds two MFMAs:
//     pair q, lane l, slot s  ->  element (row|col = l & 15, k = 8 q + 4 s + (l >> 4))        index (q * 64 + l) * 2 + s
// W_d row-block rb (16 rows) has 2 (rb + 1) pairs (k <= 16 rb + 15), stored at pair offset rb (rb + 1).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/sx_amd.h"

namespace sx {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__host__ __device__ inline int64_t w_pairs_per_output(int nrb) { return (int64_t)nrb * (nrb + 1); }
__host__ __device__ inline int64_t w_pack_doubles(int n_s, int n_pad) {
    return (int64_t)n_s * w_pairs_per_output(n_pad / 16) * 128;
}
__host__ __device__ inline int64_t r_pack_doubles(int n_s, int n_pad) { return (int64_t)n_s * (n_pad / 8) * 128; }

// index of element (c in 0..15, k) inside a fragment-ordered strip
__device__ __forceinline__ int frag_index(int c, int k) {
    return (((k >> 3) * 64 + ((k & 3) << 4) + c) << 1) + ((k >> 2) & 1);
}

template <int NS, int D>
struct GpConst {
    double inv_ls2[NS * D];
    double outputscale[NS];
    double noise[NS];
    const double* x_train;
    const double* w_pack;
    const double* r_pack;
    int n_train;
    int n_pad;
    int stage_cap;  // gp_stage_cap(NS, n_pad, waves per workgroup)
};

// LDS carve-up of one GP tile (all in doubles, 16-byte aligned pieces)
template <int NS, int D>
struct GpTileLds {
    double* xs;     // [n_train x D]      training inputs, loaded once per kernel
    double* kfrag;  // [NS][n_pad x 16]   Kstar in fragment order
    double* mj;     // [NS][16 rows][16]  mean / Jacobian rows
    double* part;   // [NW][NS][16]       per-wave partial sums of squares
    double* zs;     // [16][D]            query points
    int4* stages;   // [NW][stage_cap]    static MFMA operand stream of each wave (built once per kernel)
    int* nstages;   // [NW]
    __device__ double* carve(double* base, int n_train, int n_pad, int nw, int stage_cap) {
        xs = base;
        kfrag = xs + ((n_train * D + 1) & ~1);
        mj = kfrag + (size_t)NS * n_pad * 16;
        part = mj + NS * 256;
        zs = part + (size_t)nw * NS * 16;
        stages = reinterpret_cast<int4*>(zs + 16 * D);
        nstages = reinterpret_cast<int*>(stages + (size_t)nw * stage_cap);
        return zs + 16 * D + 2 * (size_t)nw * stage_cap + ((nw + 1) >> 1);
    }
};

// MFMA work decomposition shared by host (LDS sizing) and device (stream construction).
//   task j <  NS : the mean/Jacobian rows of output j            (n_pad / 8 fragment pairs)
//   task j >= NS : row-block rb of W_d, rb descending             (2 (rb + 1) pairs)
// snake-assigned to the waves by cost; every task is cut into stages of 4 pairs (a 2-pair tail when needed).
__host__ __device__ inline int gp_task_of(int round, int wave, int nw) {
    return round * nw + ((round & 1) ? (nw - 1 - wave) : wave);
}
__host__ __device__ inline int gp_task_pairs(int j, int ns, int nrb) {
    return j < ns ? 2 * nrb : 2 * (nrb - (j - ns) / ns);
}
inline int gp_stage_cap(int ns, int n_pad, int nw) {
    const int nrb = n_pad >> 4, ntask = ns * (nrb + 1), rounds = (ntask + nw - 1) / nw;
    int cap = 0;
    for (int w = 0; w < nw; ++w) {
        int n = 0;
        for (int r = 0; r < rounds; ++r) {
            const int j = gp_task_of(r, w, nw);
            if (j < ntask) n += (gp_task_pairs(j, ns, nrb) + 3) / 4;
        }
        cap = n > cap ? n : cap;
    }
    return (cap + 1) & ~1;  // even: the stream is consumed two stages at a time
}

inline size_t gp_tile_lds_doubles(int ns, int d, int n_train, int n_pad, int nw) {
    return (size_t)((n_train * d + 1) & ~1) + (size_t)ns * n_pad * 16 + ns * 256 + (size_t)nw * ns * 16 + 16 * d +
           2 * (size_t)nw * gp_stage_cap(ns, n_pad, nw) + ((nw + 1) >> 1);
}

template <int NS, int D>
__device__ __forceinline__ void gp_load_xs(const GpConst<NS, D>& gc, GpTileLds<NS, D>& lds) {
    for (int i = threadIdx.x; i < gc.n_train * D; i += blockDim.x) lds.xs[i] = gc.x_train[i];
}

// Phase 1: Kstar_d[c][k] = s_d exp(-1/2 sum_j (z_cj - X_kj)^2 / l_dj^2) for the tile's 16 points, all k, all d.
// blockDim.x must be a multiple of 16: a thread keeps the same query point c = tid & 15 for every k it visits.
template <int NS, int D>
__device__ __forceinline__ void gp_kstar_phase(const GpConst<NS, D>& gc, GpTileLds<NS, D>& lds) {
    const int c = threadIdx.x & 15;
    double z[D];
#pragma unroll
    for (int j = 0; j < D; ++j) z[j] = lds.zs[c * D + j];
    const int kstep = blockDim.x >> 4;
    for (int k = threadIdx.x >> 4; k < gc.n_pad; k += kstep) {
        const int fi = frag_index(c, k);
        if (k < gc.n_train) {
            double sq[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double df = z[j] - lds.xs[k * D + j];
                sq[j] = df * df;
            }
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                double arg = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) arg += sq[j] * gc.inv_ls2[d * D + j];
                lds.kfrag[(size_t)d * gc.n_pad * 16 + fi] = gc.outputscale[d] * exp(-0.5 * arg);
            }
        } else {
#pragma unroll
            for (int d = 0; d < NS; ++d) lds.kfrag[(size_t)d * gc.n_pad * 16 + fi] = 0.0;
        }
    }
}

// Phase 2: the triangular products on the matrix cores.
//
// A wave's share of the work never changes, so it is written to LDS once per kernel (gp_build_stages) as a flat
// stream of stage descriptors
//     int4 { x: pair offset into w_pack (r_pack when extra), y: v2d offset into kfrag, z: output d, w: flags }
// flags: 1 = four pairs (else two), 2 = last stage of its task, 4 = mean/Jacobian rows ("extra"), 8 = valid.
// One stage = up to 4 fragment pairs = 8 MFMAs = 512 matrix-core cycles.  The stream is consumed two stages at a
// time with two register sets, so the A fragments (global: L2-resident W) and B fragments (LDS: Kstar) of stage
// i + 1 are in flight while stage i computes -- across task boundaries too.
constexpr int kStageFour = 1, kStageLast = 2, kStageExtra = 4, kStageValid = 8;

template <int NS, int D>
__device__ __forceinline__ void gp_build_stages(const GpConst<NS, D>& gc, GpTileLds<NS, D>& lds, int wave, int nw,
                                                int lane) {
    const int nrb = gc.n_pad >> 4;
    const int ntask = NS * (nrb + 1);
    const int wpo = (int)w_pairs_per_output(nrb);
    // lane = round; (host guarantees rounds <= 64)
    const int rounds = (ntask + nw - 1) / nw;
    const int j = gp_task_of(lane, wave, nw);
    const bool has = lane < rounds && j < ntask;
    const int npairs = has ? gp_task_pairs(j, NS, nrb) : 0;
    const int mine = (npairs + 3) >> 2;
    int incl = mine;  // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    const int total = __shfl(incl, 63);
    int4* out = lds.stages + (size_t)wave * gc.stage_cap;
    if (has) {
        int d, a0, extra;
        if (j < NS) {
            d = j;
            a0 = j * (gc.n_pad >> 3);
            extra = kStageExtra;
        } else {
            const int jj = j - NS;
            const int rb = nrb - 1 - jj / NS;
            d = jj % NS;
            a0 = d * wpo + rb * (rb + 1);
            extra = 0;
        }
        int pos = incl - mine;
        for (int q = 0; q < npairs; q += 4, ++pos) {
            const int fl = kStageValid | extra | ((npairs - q >= 4) ? kStageFour : 0) | ((q + 4 >= npairs) ? kStageLast : 0);
            out[pos] = int4{a0 + q, d * gc.n_pad * 8 + q * 64, d, fl};
        }
    }
    if (lane == 0) {
        if (total & 1) out[total] = int4{0, 0, 0, 0};  // pad to an even count with an invalid stage
        lds.nstages[wave] = (total + 1) & ~1;
    }
}

struct MfmaStage {
    v2d a[4];
    v2d b[4];
    int z, w;
};

template <int NS, int D>
__device__ __forceinline__ void gp_mfma_phase(const GpConst<NS, D>& gc, GpTileLds<NS, D>& lds, int wave, int nw,
                                              int lane) {
    double ssq[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) ssq[d] = 0.0;

    const int4* stages = lds.stages + (size_t)wave * gc.stage_cap;
    const int nst = __builtin_amdgcn_readfirstlane(lds.nstages[wave]);
    const v2d* wbase = reinterpret_cast<const v2d*>(gc.w_pack) + lane;
    const v2d* rbase = reinterpret_cast<const v2d*>(gc.r_pack) + lane;
    const v2d* kbase = reinterpret_cast<const v2d*>(lds.kfrag) + lane;

    // descriptors are wave-uniform: SGPRs, scalar branches.  Always four loads per operand: a 2-pair stage re-reads
    // its second pair (never consumed) instead of branching around loads.
    auto issue = [&](MfmaStage& st, int i) {
        const int4 t = stages[i];
        const int x = __builtin_amdgcn_readfirstlane(t.x), y = __builtin_amdgcn_readfirstlane(t.y);
        st.z = __builtin_amdgcn_readfirstlane(t.z);
        st.w = __builtin_amdgcn_readfirstlane(t.w);
        const v2d* ap = ((st.w & kStageExtra) ? rbase : wbase) + (int64_t)x * 64;
        const v2d* bp = kbase + y;
        const int i2 = (st.w & kStageFour) ? 128 : 64, i3 = (st.w & kStageFour) ? 192 : 64;
        st.a[0] = ap[0];
        st.a[1] = ap[64];
        st.a[2] = ap[i2];
        st.a[3] = ap[i3];
        st.b[0] = bp[0];
        st.b[1] = bp[64];
        st.b[2] = bp[i2];
        st.b[3] = bp[i3];
    };

    v4d acc0 = {0.0, 0.0, 0.0, 0.0};
    v4d acc1 = {0.0, 0.0, 0.0, 0.0};
    auto compute = [&](const MfmaStage& st) {
        if (!(st.w & kStageValid)) return;
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(st.a[0].x, st.b[0].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(st.a[0].y, st.b[0].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(st.a[1].x, st.b[1].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(st.a[1].y, st.b[1].y, acc1, 0, 0, 0);
        if (st.w & kStageFour) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(st.a[2].x, st.b[2].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(st.a[2].y, st.b[2].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(st.a[3].x, st.b[3].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(st.a[3].y, st.b[3].y, acc1, 0, 0, 0);
        }
        if (st.w & kStageLast) {
            const v4d t = acc0 + acc1;
            if (st.w & kStageExtra) {
                // row (lane >> 4) + 4 r of the 16 extra rows, column = query point lane & 15
#pragma unroll
                for (int r = 0; r < 4; ++r) lds.mj[st.z * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = t[r];
            } else {
                const double s = t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
#pragma unroll
                for (int dd = 0; dd < NS; ++dd) ssq[dd] += (st.z == dd) ? s : 0.0;
            }
            acc0 = v4d{0.0, 0.0, 0.0, 0.0};
            acc1 = v4d{0.0, 0.0, 0.0, 0.0};
        }
    };

    if (nst > 0) {
        MfmaStage sx, sy;
        issue(sx, 0);
        for (int i = 0; i < nst; i += 2) {
            issue(sy, i + 1);
            compute(sx);
            if (i + 2 < nst) issue(sx, i + 2);
            compute(sy);
        }
    }
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        double v = ssq[d];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) lds.part[(wave * NS + d) * 16 + lane] = v;
    }
}

// Phase 3 (one thread per query point c): assemble mean, variance (noise included) and the mean-Jacobian.
template <int NS, int D, bool WITH_JAC>
__device__ __forceinline__ void gp_collect(const GpConst<NS, D>& gc, const GpTileLds<NS, D>& lds, int nw, int c,
                                           const double (&z)[D], double (&mean)[NS], double (&var)[NS],
                                           double (&jac)[NS][D]) {
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        double q = 0.0;
        for (int w = 0; w < nw; ++w) q += lds.part[(w * NS + d) * 16 + c];
        var[d] = (gc.outputscale[d] - q) + gc.noise[d];
        const double m = lds.mj[d * 256 + c];
        mean[d] = m;
        if constexpr (WITH_JAC) {
#pragma unroll
            for (int j = 0; j < D; ++j) jac[d][j] = lds.mj[d * 256 + (1 + j) * 16 + c] - z[j] * gc.inv_ls2[d * D + j] * m;
        }
    }
}

}  // namespace sx
