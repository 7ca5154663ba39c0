// MC-dropout ensembles on the f64 matrix cores (reference ssm_cem/dropout_ssm_cem.py, gal_concrete_dropout.py;
// the model itself is specified in sx_mlp.hpp).  Networks with ONE or TWO hidden layers of up to 64 units -- the reference's
// default (64 x 64, experiments/sacred_helper.py:100) and everything gal_concrete_dropout.py allows (:77) -- take this path;
// anything else stays on the VALU kernel of sx_mlp.hpp.
//
// A workgroup (8 waves, one CU) owns a tile of 16 particles -- the N dimension of v_mfma_f64_16x16x4_f64 -- for the whole
// rollout.  The ensemble members are dealt out to the waves; a wave takes ONE member through its forward pass and its
// reverse sweeps entirely in registers, with no barrier and no LDS traffic for the activations.  What makes that possible
// is that the D layout of this MFMA is its own B layout one layer later:
//     D: lane l holds D[row = (l >> 4) + 4 r][col = l & 15], r = 0..3        (sx_gp.hpp)
//     B: lane l holds B[k   = (l >> 4)      ][col = l & 15]  of K-chunk kc
// so accumulator element r of row-block rb IS the B operand of chunk kc = 4 rb + r of the next layer (unit = 4 kc + (l >> 4)
// either way): relu and the dropout mask are applied in place, and 16 doubles per lane carry a 64-unit layer for 16 particles.
// The weights are the A operands, shared by all members, tiles and steps: they sit in LDS in fragment-pair order (one
// ds_read_b128 per lane feeds two MFMAs), packed there once per kernel from the plain row-major `net` of the C ABI:
//     W1a  [w1 x (D + 1)]   layer 1 with the bias as column D (the B operand carries a constant 1 in that row)
//     W2   [w2 x w1]        (two hidden layers)                     W2T  [w1 x w2]   its transpose, for the reverse sweep
//     Wout [n_out x wL]     mean rows (and log-std rows)            W1T  [D x w1]    the Jacobian rows
// Per member and tile at 64 x 64: 4 + 64 + 16 MFMAs forward, (64 + 16) per output backward = 244 for the pendulum; the masks
// (per member, unit) come from global memory through the vector cache, 16 per lane and layer.
// After its members a wave leaves (count, mean, M2) of Welford's recurrence, the aleatoric sums and the Jacobian sums in LDS;
// lanes 0-15 of wave 0 merge the waves' partials (Chan's pairwise update), run the reachability step and the costs for their
// particle exactly as the VALU kernel does, and publish the next query point.  Two barriers per step.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sx_amd.h"
#include "sx_gp.hpp"
#include "sx_mlp.hpp"

namespace sx {

constexpr int kMmWaves = 8;
constexpr int kMmThreads = kMmWaves * 64;
constexpr int kMmTile = 16;
constexpr int kMmW = 64;           // widest hidden layer; a layer is padded with zero units to a multiple of 16
constexpr int kMmPair = 128;       // doubles per fragment pair (64 lanes x 2 slots)
constexpr int kMmMaskRow = 8 + 2 * kMmW;   // a member's masks: input (padded to 8), hidden layer 1, hidden layer 2

// LDS map (doubles).  Fragment matrices first, then the plain vectors, the query points and the waves' partial sums.
template <int NS, int D>
struct MmLds {
    static constexpr int kSlots = NS * (3 + D);                    // mean, M2, aleatoric sum, D Jacobian entries per output
    static constexpr int w1a = 0;                                  // 4 row-blocks x 1 pair
    static constexpr int w2 = w1a + 4 * kMmPair;                   // 4 row-blocks x 8 pairs
    static constexpr int w2t = w2 + 32 * kMmPair;
    static constexpr int wout = w2t + 32 * kMmPair;                // 1 row-block x 8 pairs
    static constexpr int w1t = wout + 8 * kMmPair;
    static constexpr int b2 = w1t + 8 * kMmPair;                   // [64] bias of hidden layer 2
    static constexpr int bout = b2 + kMmW;                         // [16]
    static constexpr int woutp = bout + 16;                        // [NS][64] plain mean rows of the output layer
    static constexpr int zbuf = woutp + NS * kMmW;                 // [16][8] query points
    static constexpr int part = zbuf + kMmTile * 8;                // [waves][kSlots][16]
    static constexpr int wmask = part + kMmWaves * kSlots * kMmTile;   // [waves][kMmMaskRow] the current member's masks
    static constexpr int state = wmask + kMmWaves * kMmMaskRow;        // [16][NS * NS + 2] rollout: Q, objective, constraint cost
    static constexpr int total = state + kMmTile * (NS * NS + 2);
};

struct MmDims {
    int w1, w2, n_out, n_samples, predict_std;
    int nrb1, nrb2;          // row-blocks of the hidden layers (w2 = nrb2 = 0 with one hidden layer)
    int moff1, moff2, msum;  // offsets of the layer masks inside a member's mask row
};

__host__ __device__ inline bool mlp_mfma_ok(const MlpConst& mc) {
    if (mc.n_hidden < 1 || mc.n_hidden > 2 || mc.n_out > 16 || mc.d_in + 1 > 8) return false;
    for (int l = 1; l <= mc.n_hidden; ++l)
        if (mc.width[l] > kMmW) return false;
    return true;
}

__host__ __device__ inline MmDims mm_dims(const MlpConst& mc) {
    MmDims d;
    d.w1 = mc.width[1];
    d.w2 = mc.n_hidden == 2 ? mc.width[2] : 0;
    d.n_out = mc.n_out;
    d.n_samples = mc.n_samples;
    d.predict_std = mc.predict_std;
    d.nrb1 = (d.w1 + 15) / 16;
    d.nrb2 = (d.w2 + 15) / 16;
    d.moff1 = mc.d_in;
    d.moff2 = mc.d_in + d.w1;
    d.msum = mc.d_in + d.w1 + d.w2;
    return d;
}
// every hidden layer 64 wide: the kernels' straight-line instantiation
__host__ __device__ inline bool mlp_mfma_full(const MlpConst& mc) {
    for (int l = 1; l <= mc.n_hidden; ++l)
        if (mc.width[l] != kMmW) return false;
    return true;
}

// element (row, k) of a fragment-pair matrix with `npairs` pairs per row-block -> index in doubles
__device__ __forceinline__ void mm_frag_decode(int idx, int npairs, int& row, int& k) {
    const int slot = idx & 1, lane = (idx >> 1) & 63, pair = idx >> 7;
    const int rb = pair / npairs, q = pair - rb * npairs;
    row = 16 * rb + (lane & 15);
    k = 8 * q + 4 * slot + (lane >> 4);
}

// every thread of the workgroup: net (row-major, C ABI) -> LDS fragments.  Padding rows / columns are zero.
template <int NS, int D>
__device__ __forceinline__ void mm_pack(const MlpConst& mc, const MmDims& dm, double* lds, int tid) {
    using M = MmLds<NS, D>;
    const int L = mc.n_hidden;
    const double* W1 = mc.net;
    const double* b1 = W1 + (size_t)dm.w1 * D;
    const double* W2 = b1 + dm.w1;                       // (two hidden layers)
    const double* b2 = W2 + (size_t)dm.w2 * dm.w1;
    const int wl = L == 2 ? dm.w2 : dm.w1;
    const double* Wo = L == 2 ? b2 + dm.w2 : b1 + dm.w1;
    const double* bo = Wo + (size_t)dm.n_out * wl;
    int row, k;
    for (int i = tid; i < 4 * kMmPair; i += kMmThreads) {
        mm_frag_decode(i, 1, row, k);
        double v = 0.0;
        if (row < dm.w1) v = k < D ? W1[row * D + k] : (k == D ? b1[row] : 0.0);
        lds[M::w1a + i] = v;
    }
    for (int i = tid; i < 32 * kMmPair; i += kMmThreads) {
        mm_frag_decode(i, 8, row, k);
        lds[M::w2 + i] = (L == 2 && row < dm.w2 && k < dm.w1) ? W2[row * dm.w1 + k] : 0.0;
        lds[M::w2t + i] = (L == 2 && row < dm.w1 && k < dm.w2) ? W2[k * dm.w1 + row] : 0.0;
    }
    for (int i = tid; i < 8 * kMmPair; i += kMmThreads) {
        mm_frag_decode(i, 8, row, k);
        lds[M::wout + i] = (row < dm.n_out && k < wl) ? Wo[row * wl + k] : 0.0;
        lds[M::w1t + i] = (row < D && k < dm.w1) ? W1[k * D + row] : 0.0;
    }
    for (int i = tid; i < kMmW; i += kMmThreads) lds[M::b2 + i] = (L == 2 && i < dm.w2) ? b2[i] : 0.0;
    for (int i = tid; i < 16; i += kMmThreads) lds[M::bout + i] = i < dm.n_out ? bo[i] : 0.0;
    for (int i = tid; i < NS * kMmW; i += kMmThreads) {
        const int d = i / kMmW, u = i - d * kMmW;
        lds[M::woutp + i] = u < wl ? Wo[d * wl + u] : 0.0;
    }
    for (int i = tid; i < kMmWaves * kMmMaskRow; i += kMmThreads) lds[M::wmask + i] = 0.0;
}

#define SX_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// 1.0 where bit k of `bits` is set, else 0.0 -- as arithmetic on the high word, so that the compiler multiplies by it
// instead of branching around the LDS reads of the other factor
__device__ __forceinline__ double mm_gate(unsigned bits, int k) {
    const int all = (int)(bits << (31 - k)) >> 31;   // 0 or -1
    return __hiloint2double(all & 0x3ff00000, 0);
}

// ---- every hidden layer 64 wide (the reference's default): straight-line code --------------------------------------
// acc[rb] += A[rb][:] . B for the four row-blocks of a 64 x 64 layer: K-pairs outermost, so that the four accumulators
// are independent chains and the fragments of pair q + 1 are in flight while pair q computes.
__device__ __forceinline__ void mm_layer64(const v2d* __restrict__ frag, const double (&bop)[16], v4d (&acc)[4]) {
    constexpr bool FULL = true;
    constexpr int nrb = 4, npairs = 8;
    v2d fr[4], nx[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) fr[rb] = (FULL || rb < nrb) ? frag[(rb * 8) * 64] : v2d{0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (FULL || q < npairs) {
            if (q + 1 < 8) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
                    nx[rb] = (FULL || (rb < nrb && q + 1 < npairs)) ? frag[(rb * 8 + q + 1) * 64] : v2d{0.0, 0.0};
            }
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                if (FULL || rb < nrb) acc[rb] = SX_MFMA(fr[rb].x, bop[2 * q], acc[rb]);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                if (FULL || rb < nrb) acc[rb] = SX_MFMA(fr[rb].y, bop[2 * q + 1], acc[rb]);
            if (q + 1 < 8) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) fr[rb] = nx[rb];
            }
            SX_PIN();
        }
    }
}

// acc += A[0][:] . B for ONE row-block (output layer, Jacobian rows): a dependent chain, fragments two pairs ahead
__device__ __forceinline__ v4d mm_rows16(const v2d* __restrict__ frag, const double (&bop)[16], v4d acc) {
    constexpr bool FULL = true;
    constexpr int npairs = 8;
    v2d fr[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) fr[q] = (FULL || q < npairs) ? frag[q * 64] : v2d{0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (FULL || q < npairs) {
            acc = SX_MFMA(fr[q].x, bop[2 * q], acc);
            acc = SX_MFMA(fr[q].y, bop[2 * q + 1], acc);
        }
    }
    SX_PIN();
    return acc;
}

// One member for the wave's 16 particles.  `zc` is the query point of particle lane & 15, `wm` the member's masks in LDS
// ([0..7] input, [8 + unit] hidden layer 1, [8 + 64 + unit] hidden layer 2; zero beyond the layer's width).  Returns the
// output rows (row = (lane >> 4) + 4 r) and, WITH_JAC, d out_d / d z_j at (lane >> 4) + 4 r = j for r = 0, 1.
template <int NS, int D, int L, bool WITH_JAC>
__device__ __forceinline__ void mm_member_full(const double* lds, const double* wm, int lane, const double (&zc)[D],
                                               v4d& out, double (&jrow)[NS][2]) {
    using M = MmLds<NS, D>;
    constexpr bool FULL = true;
    const int g = lane >> 4;
    const v2d* w1a = reinterpret_cast<const v2d*>(lds + M::w1a) + lane;
    const v2d* w2 = reinterpret_cast<const v2d*>(lds + M::w2) + lane;
    const v2d* w2t = reinterpret_cast<const v2d*>(lds + M::w2t) + lane;
    const v2d* wo = reinterpret_cast<const v2d*>(lds + M::wout) + lane;
    const v2d* w1t = reinterpret_cast<const v2d*>(lds + M::w1t) + lane;
    const double* wm1 = wm + 8 + g;            // mask of unit 4 kc + g of hidden layer 1 at wm1[4 kc]
    const double* wm2 = wm + 8 + kMmW + g;
    constexpr int nrb1 = 4;

    // input rows of the B operand: masked z, then the constant 1 that picks up the bias column
    double zin[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) zin[j] = j < D ? zc[j] * wm[j] : (j == D ? 1.0 : 0.0);
    const double b0x = g == 0 ? zin[0] : (g == 1 ? zin[1] : (g == 2 ? zin[2] : zin[3]));
    const double b0y = g == 0 ? zin[4] : (g == 1 ? zin[5] : (g == 2 ? zin[6] : zin[7]));

    // a1 / a2: the activations (B operands of the next layer); on1 / on2: bit kc = unit 4 kc + g is active, which is all
    // the reverse sweep keeps of them (the masks are read again from LDS: registers)
    double a1[16], a2[16];
    unsigned on1 = 0, on2 = 0;
    {
        v2d fr[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) fr[rb] = w1a[rb * 64];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            if (FULL || rb < nrb1) {
                acc = SX_MFMA(fr[rb].x, b0x, acc);
                if constexpr (D + 1 > 4) acc = SX_MFMA(fr[rb].y, b0y, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double m = wm1[4 * (4 * rb + r)];
                const double v = fmax(acc[r], 0.0) * m;   // (no select: the compiler would branch around the mask read)
                a1[4 * rb + r] = v;
                on1 |= (v > 0.0 ? 1u : 0u) << (4 * rb + r);
            }
        }
        SX_PIN();
    }
    if constexpr (L == 2) {
        v4d acc[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rb][r] = lds[M::b2 + 16 * rb + 4 * r + g];
        mm_layer64(w2, a1, acc);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double m = wm2[4 * (4 * rb + r)];
                const double v = fmax(acc[rb][r], 0.0) * m;
                a2[4 * rb + r] = v;
                on2 |= (v > 0.0 ? 1u : 0u) << (4 * rb + r);
            }
        SX_PIN();
    }
    const double(&al)[16] = L == 2 ? a2 : a1;       // last hidden layer
    const unsigned onl = L == 2 ? on2 : on1;
    const double* wml = L == 2 ? wm2 : wm1;
    {
        v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = lds[M::bout + 4 * r + g];
        out = mm_rows16(wo, al, acc);
    }
    if constexpr (WITH_JAC) {
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            double gl[16];   // d out_d / d (pre-activation) of the last hidden layer, B layout
#pragma unroll
            for (int kc = 0; kc < 16; ++kc) {
                const double v = lds[M::woutp + d * kMmW + 4 * kc + g] * wml[4 * kc];
                gl[kc] = v * mm_gate(onl, kc);
            }
            SX_PIN();
            double g1[16];
            if constexpr (L == 2) {
                v4d acc[4];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) acc[rb] = v4d{0.0, 0.0, 0.0, 0.0};
                mm_layer64(w2t, gl, acc);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double m = wm1[4 * (4 * rb + r)];
                        g1[4 * rb + r] = acc[rb][r] * m * mm_gate(on1, 4 * rb + r);
                    }
                SX_PIN();
            } else {
#pragma unroll
                for (int kc = 0; kc < 16; ++kc) g1[kc] = gl[kc];
            }
            const v4d acc = mm_rows16(w1t, g1, v4d{0.0, 0.0, 0.0, 0.0});
            // row j = g + 4 r of W1^T g1 is d out_d / d (masked z_j): the input mask once more (wm[j] = 0 for j >= D)
            jrow[d][0] = acc[0] * wm[g];
            jrow[d][1] = acc[1] * wm[g + 4];
        }
    }
}

// (generic widths) One member for the wave's 16 particles: row-block by row-block, run-time (uniform) guards.  `zc` is the query point of particle lane & 15, `wm` the member's masks in LDS
// ([0..7] input, [8 + unit] hidden layer 1, [8 + 64 + unit] hidden layer 2; zero beyond the layer's width).  Returns the
// output rows (row = (lane >> 4) + 4 r) and, WITH_JAC, d out_d / d z_j at (lane >> 4) + 4 r = j for r = 0, 1.
template <int NS, int D, int L, bool WITH_JAC>
__device__ __forceinline__ void mm_member_generic(const MmDims& dm, const double* lds, const double* wm, int lane,
                                          const double (&zc)[D], v4d& out, double (&jrow)[NS][2]) {
    using M = MmLds<NS, D>;
    const int g = lane >> 4;
    const v2d* w1a = reinterpret_cast<const v2d*>(lds + M::w1a) + lane;
    const v2d* w2 = reinterpret_cast<const v2d*>(lds + M::w2) + lane;
    const v2d* w2t = reinterpret_cast<const v2d*>(lds + M::w2t) + lane;
    const v2d* wo = reinterpret_cast<const v2d*>(lds + M::wout) + lane;
    const v2d* w1t = reinterpret_cast<const v2d*>(lds + M::w1t) + lane;
    const double* wm1 = wm + 8 + g;            // mask of unit 4 kc + g of hidden layer 1 at wm1[4 kc]
    const double* wm2 = wm + 8 + kMmW + g;

    // input rows of the B operand: masked z, then the constant 1 that picks up the bias column
    double zin[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) zin[j] = j < D ? zc[j] * wm[j] : (j == D ? 1.0 : 0.0);
    const double b0x = g == 0 ? zin[0] : (g == 1 ? zin[1] : (g == 2 ? zin[2] : zin[3]));
    const double b0y = g == 0 ? zin[4] : (g == 1 ? zin[5] : (g == 2 ? zin[6] : zin[7]));

    // a1 / a2: the activations (B operands of the next layer); on1 / on2: bit kc = unit 4 kc + g is active, which is all
    // the reverse sweep keeps of them (the masks are read again from LDS: registers)
    double a1[16], a2[16];
    unsigned on1 = 0, on2 = 0;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        if (rb < dm.nrb1) {
            const v2d a = w1a[rb * 64];
            v4d acc = SX_MFMA(a.x, b0x, (v4d{0.0, 0.0, 0.0, 0.0}));
            if constexpr (D + 1 > 4) acc = SX_MFMA(a.y, b0y, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double m = wm1[4 * (4 * rb + r)];
                const double v = fmax(acc[r], 0.0) * m;   // (no select: the compiler would branch around the mask read)
                a1[4 * rb + r] = v;
                on1 |= (v > 0.0 ? 1u : 0u) << (4 * rb + r);
            }
            SX_PIN();
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) a1[4 * rb + r] = 0.0;
        }
    }
    if constexpr (L == 2) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            if (rb < dm.nrb2) {
                v4d acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = lds[M::b2 + 16 * rb + 4 * r + g];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (q < 2 * dm.nrb1) {
                        const v2d a = w2[(rb * 8 + q) * 64];
                        acc = SX_MFMA(a.x, a1[2 * q], acc);
                        acc = SX_MFMA(a.y, a1[2 * q + 1], acc);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double m = wm2[4 * (4 * rb + r)];
                    const double v = fmax(acc[r], 0.0) * m;   // (no select: the compiler would branch around the mask read)
                    a2[4 * rb + r] = v;
                    on2 |= (v > 0.0 ? 1u : 0u) << (4 * rb + r);
                }
                SX_PIN();
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) a2[4 * rb + r] = 0.0;
            }
        }
    }
    const double(&al)[16] = L == 2 ? a2 : a1;       // last hidden layer
    const unsigned onl = L == 2 ? on2 : on1;
    const double* wml = L == 2 ? wm2 : wm1;
    const int nrbl = L == 2 ? dm.nrb2 : dm.nrb1;
    {
        v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = lds[M::bout + 4 * r + g];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q < 2 * nrbl) {
                const v2d a = wo[q * 64];
                acc = SX_MFMA(a.x, al[2 * q], acc);
                acc = SX_MFMA(a.y, al[2 * q + 1], acc);
            }
        }
        out = acc;
        SX_PIN();
    }
    if constexpr (WITH_JAC) {
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            double gl[16];   // d out_d / d (pre-activation) of the last hidden layer, B layout
#pragma unroll
            for (int kc = 0; kc < 16; ++kc)
            {
                const double v = lds[M::woutp + d * kMmW + 4 * kc + g] * wml[4 * kc];
                gl[kc] = v * mm_gate(onl, kc);
            }
            SX_PIN();
            double g1[16];
            if constexpr (L == 2) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    if (rb < dm.nrb1) {
                        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            if (q < 2 * dm.nrb2) {
                                const v2d a = w2t[(rb * 8 + q) * 64];
                                acc = SX_MFMA(a.x, gl[2 * q], acc);
                                acc = SX_MFMA(a.y, gl[2 * q + 1], acc);
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double m = wm1[4 * (4 * rb + r)];
                            g1[4 * rb + r] = acc[r] * m * mm_gate(on1, 4 * rb + r);
                        }
                        SX_PIN();
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) g1[4 * rb + r] = 0.0;
                    }
                }
            } else {
#pragma unroll
                for (int kc = 0; kc < 16; ++kc) g1[kc] = gl[kc];
            }
            v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q < 2 * dm.nrb1) {
                    const v2d a = w1t[q * 64];
                    acc = SX_MFMA(a.x, g1[2 * q], acc);
                    acc = SX_MFMA(a.y, g1[2 * q + 1], acc);
                }
            }
            SX_PIN();
            // row j = g + 4 r of W1^T g1 is d out_d / d (masked z_j): the input mask once more (wm[j] = 0 for j >= D)
            jrow[d][0] = acc[0] * wm[g];
            jrow[d][1] = acc[1] * wm[g + 4];
        }
    }
}

// The ensemble for one tile: every wave runs its members and leaves its partial sums in LDS (the caller synchronises).
// slots per wave: [d] mean, [NS + d] M2, [2 NS + d] aleatoric sum, [3 NS + d D + j] Jacobian sum
template <int NS, int D, int L, bool WITH_JAC, bool FULL>
__device__ __forceinline__ void mm_ensemble_wave(const MlpConst& mc, const MmDims& dm, double* lds, int wave, int lane,
                                                 const double (&zc)[D]) {
    using M = MmLds<NS, D>;
    const int g = lane >> 4, c = lane & 15;
    double mean = 0.0, m2 = 0.0, alea[2] = {0.0, 0.0}, jsum[NS][2];
#pragma unroll
    for (int d = 0; d < NS; ++d) jsum[d][0] = jsum[d][1] = 0.0;
    int n = 0;
    double* wm = lds + M::wmask + wave * kMmMaskRow;
    // A member's masks go global -> registers -> this wave's LDS row (padding stays zero, see mm_pack); the loads for
    // the NEXT member are issued before the current one computes, so their latency is never waited for.  A wave's LDS
    // operations execute in order: no barrier between these writes and the reads in mm_member, only compiler fences.
    double mreg[3] = {0.0, 0.0, 0.0};
    auto fetch = [&](int s) {
        const double* mk = mc.masks + (size_t)s * dm.msum;
        if (lane < D) mreg[0] = mk[lane];
        if (lane < dm.w1) mreg[1] = mk[dm.moff1 + lane];
        if (L == 2 && lane < dm.w2) mreg[2] = mk[dm.moff2 + lane];
    };
    if (wave < dm.n_samples) fetch(wave);
    for (int s = wave; s < dm.n_samples; s += kMmWaves) {
        if (lane < D) wm[lane] = mreg[0];
        if (lane < dm.w1) wm[8 + lane] = mreg[1];
        if (L == 2 && lane < dm.w2) wm[8 + kMmW + lane] = mreg[2];
        if (s + kMmWaves < dm.n_samples) fetch(s + kMmWaves);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        v4d out;
        double jrow[NS][2];
        if constexpr (FULL)
            mm_member_full<NS, D, L, WITH_JAC>(lds, wm, lane, zc, out, jrow);
        else
            mm_member_generic<NS, D, L, WITH_JAC>(dm, lds, wm, lane, zc, out, jrow);
        __builtin_amdgcn_wave_barrier();   // (the next member's masks overwrite the row)
        ++n;
        // Welford on row g (r = 0): the mean outputs are rows 0 .. NS - 1 <= 3
        const double delta = out[0] - mean;
        mean += delta / (double)n;
        m2 = fma(delta, out[0] - mean, m2);
        if (dm.predict_std) {   // rows NS + d hold log sigma_d
            alea[0] += exp(2.0 * out[0]);
            alea[1] += exp(2.0 * out[1]);
        }
        if constexpr (WITH_JAC) {
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                jsum[d][0] += jrow[d][0];
                jsum[d][1] += jrow[d][1];
            }
        }
    }
    double* part = lds + M::part + (size_t)wave * M::kSlots * kMmTile;
    if (g < NS) {
        part[g * kMmTile + c] = mean;
        part[(NS + g) * kMmTile + c] = m2;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = g + 4 * r;
        if (row >= NS && row < 2 * NS) part[(2 * NS + row - NS) * kMmTile + c] = alea[r];
        if constexpr (WITH_JAC) {
            if (row < D) {
#pragma unroll
                for (int d = 0; d < NS; ++d) part[(3 * NS + d * D + row) * kMmTile + c] = jsum[d][r];
            }
        }
    }
}

// lanes 0-15 of one wave, after the barrier: merge the waves' partial sums for particle c
template <int NS, int D, bool WITH_JAC>
__device__ __forceinline__ void mm_merge(const MmDims& dm, const double* lds, int c, double (&mean)[NS], double (&var)[NS],
                                         double (&jac)[NS][D]) {
    using M = MmLds<NS, D>;
    const int S = dm.n_samples;
    double m2[NS], alea[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        mean[d] = m2[d] = alea[d] = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) jac[d][j] = 0.0;
    }
    int na = 0;
    for (int w = 0; w < kMmWaves; ++w) {
        const int nb = w < S ? (S - w + kMmWaves - 1) / kMmWaves : 0;
        if (nb == 0) continue;
        const double* part = lds + M::part + (size_t)w * M::kSlots * kMmTile;
        const int nn = na + nb;
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            const double mb = part[d * kMmTile + c], m2b = part[(NS + d) * kMmTile + c];
            const double delta = mb - mean[d];
            mean[d] += delta * ((double)nb / (double)nn);
            m2[d] += m2b + delta * delta * ((double)na * (double)nb / (double)nn);
            if (dm.predict_std) alea[d] += part[(2 * NS + d) * kMmTile + c];
            if constexpr (WITH_JAC) {
#pragma unroll
                for (int j = 0; j < D; ++j) jac[d][j] += part[(3 * NS + d * D + j) * kMmTile + c];
            }
        }
        na = nn;
    }
    const double inv_s = 1.0 / (double)S;
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        var[d] = (S > 1 ? m2[d] / (double)(S - 1) : 0.0) + (dm.predict_std ? alea[d] * inv_s : 0.0);
        if constexpr (WITH_JAC) {
#pragma unroll
            for (int j = 0; j < D; ++j) jac[d][j] *= inv_s;
        }
    }
}

template <int NS, int NU, int L, bool FULL>
__global__ __launch_bounds__(kMmThreads) void mlp_predict_mfma_kernel(MlpConst mc, const double* __restrict__ zin, int P,
                                                                      double* __restrict__ mean, double* __restrict__ var,
                                                                      double* __restrict__ jac) {
    constexpr int D = NS + NU;
    extern __shared__ __attribute__((aligned(16))) double mm_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const MmDims dm = mm_dims(mc);
    mm_pack<NS, D>(mc, dm, mm_smem, tid);
    const int64_t base = (int64_t)blockIdx.x * kMmTile;
    const int64_t gc = base + (lane & 15);
    double zc[D];
#pragma unroll
    for (int j = 0; j < D; ++j) zc[j] = gc < P ? zin[gc * D + j] : 0.0;
    __syncthreads();
    if (jac)
        mm_ensemble_wave<NS, D, L, true, FULL>(mc, dm, mm_smem, wave, lane, zc);
    else
        mm_ensemble_wave<NS, D, L, false, FULL>(mc, dm, mm_smem, wave, lane, zc);
    __syncthreads();
    if (tid < kMmTile && base + tid < P) {
        double m[NS], v[NS], jc[NS][D];
        if (jac)
            mm_merge<NS, D, true>(dm, mm_smem, tid, m, v, jc);
        else
            mm_merge<NS, D, false>(dm, mm_smem, tid, m, v, jc);
        const int64_t gp = base + tid;
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            mean[gp * NS + d] = m[d];
            var[gp * NS + d] = v[d];
            if (jac) {
#pragma unroll
                for (int j = 0; j < D; ++j) jac[(gp * NS + d) * D + j] = jc[d][j];
            }
        }
    }
}

// the CEM particle rollout over the ensemble (arguments as FeatRolloutPtrs; results as cem_rollout_mlp_kernel)
template <int NS, int NU, int L, bool FULL>
__global__ __launch_bounds__(kMmThreads) void cem_rollout_mlp_mfma_kernel(MlpConst mc, ReachConst<NS, NU> rc,
                                                                          CostConst<SX_MAX_M, NS, NU> cc,
                                                                          FeatRolloutPtrs rp) {
    constexpr int D = NS + NU;
    constexpr int S = NS + NS * NS;
    using M = MmLds<NS, D>;
    extern __shared__ __attribute__((aligned(16))) double mm_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const MmDims dm = mm_dims(mc);
    mm_pack<NS, D>(mc, dm, mm_smem, tid);

    // Particle state: owned by lanes 0-15 of wave 0, but kept in LDS between the steps (the centre and the action in the
    // query-point buffer, Q and the two costs beside it), so that no register is held across the members' matrix work.
    const bool owner = tid < kMmTile;
    const int64_t total = (int64_t)rp.E * rp.P;
    const int64_t g = (int64_t)blockIdx.x * kMmTile + (tid & 15);
    const bool valid = owner && g < total;
    const int64_t gg = g < total ? g : 0;
    const int e = (int)(gg / rp.P);
    const int H = rp.H;
    bool have_q = rp.q0 != nullptr;    // (uniform)
    int st = 0;
    double* zbuf = mm_smem + M::zbuf;
    double* sbuf = mm_smem + M::state + (tid & 15) * (NS * NS + 2);
    auto publish = [&](int t, const double (&p)[NS]) {        // the action of step t and the query point (p, u) -> LDS
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
            zbuf[tid * 8 + NS + c] = a;
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) zbuf[tid * 8 + j] = p[j];
    };
    if (owner) {
        double p[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            p[i] = rp.x0[(int64_t)e * NS + i];
#pragma unroll
            for (int j = 0; j < NS; ++j) sbuf[i * NS + j] = have_q ? rp.q0[((int64_t)e * NS + i) * NS + j] : 0.0;
        }
        sbuf[NS * NS] = sbuf[NS * NS + 1] = 0.0;
        publish(0, p);
    }
    for (int t = 0; t < H; ++t) {
        __syncthreads();   // weights packed (t = 0), query points published
        {
            double zc[D];
#pragma unroll
            for (int j = 0; j < D; ++j) zc[j] = zbuf[(lane & 15) * 8 + j];
            if (have_q)
                mm_ensemble_wave<NS, D, L, true, FULL>(mc, dm, mm_smem, wave, lane, zc);
            else
                mm_ensemble_wave<NS, D, L, false, FULL>(mc, dm, mm_smem, wave, lane, zc);
        }
        __syncthreads();   // partial sums complete; every wave has read its query points
        if (owner) {
            double p[NS], Q[NS][NS], u[NU], mean[NS], var[NS], jac[NS][D], p1[NS], Q1[NS][NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                p[i] = zbuf[tid * 8 + i];
#pragma unroll
                for (int j = 0; j < NS; ++j) Q[i][j] = sbuf[i * NS + j];
            }
#pragma unroll
            for (int c = 0; c < NU; ++c) u[c] = zbuf[tid * 8 + NS + c];
            double obj = sbuf[NS * NS], con = sbuf[NS * NS + 1];
            if (have_q) {
                mm_merge<NS, D, true>(dm, mm_smem, tid, mean, var, jac);
                reach_ellipsoid<NS, NU>(rc, p, Q, u, mean, var, jac, p1, Q1, st);
            } else {
                mm_merge<NS, D, false>(dm, mm_smem, tid, mean, var, jac);
                reach_point<NS, NU>(rc, p, u, mean, var, p1, Q1, st);
            }
            obj += objective_cost<SX_MAX_M, NS, NU>(cc, p1, var);
            bool uviol = false;
#pragma unroll
            for (int c = 0; c < NU; ++c) uviol = uviol || (u[c] < cc.u_min[c]) || (u[c] > cc.u_max[c]);
            if (uviol) con += SX_ACTION_VIOLATION_COST;
            if (cc.con_mode == SX_CON_ALL_STATES || t == H - 1) {
                if (polytope_violated<SX_MAX_M, NS>(cc.h_mat, cc.h_vec, cc.m, 1.0, p1, Q1, nullptr))
                    con += SX_STATE_VIOLATION_COST;
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
            for (int i = 0; i < NS; ++i)
#pragma unroll
                for (int j = 0; j < NS; ++j) sbuf[i * NS + j] = Q1[i][j];
            sbuf[NS * NS] = obj;
            sbuf[NS * NS + 1] = con;
            if (t + 1 < H) publish(t + 1, p1);
        }
        have_q = true;
    }
    if (valid) {
        rp.obj_cost[g] = sbuf[NS * NS];
        rp.con_cost[g] = sbuf[NS * NS + 1];
        if (st) atomicOr(rp.status, st);
    }
}

}  // namespace sx
