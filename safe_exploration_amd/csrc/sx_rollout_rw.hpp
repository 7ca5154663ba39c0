// The fused CEM particle rollout with the GP's triangular factors RESIDENT IN THE REGISTER FILE (round 3).
//
// A compute unit of gfx950 has 4 SIMDs x 512 registers x 64 lanes x 4 B = 512 KB of vector registers (VGPRs and AGPRs are
// one file, and an MFMA takes its A operand from either half) -- more than the 372 KB that W_d = chol(K_d + noise_d I)^-1
// of a pendulum-sized model (n_s = 2, N <= 204) occupies in MFMA fragment order.  cem_rollout_kernel (sx_rollout.hpp)
// streams those fragments from L2 on every step of every particle tile: 364 KB per workgroup and step, one
// buffer_load + descriptor decode per 4 MFMAs, four rotating register sets, a dynamic stage loop whose per-row-block
// epilogue (1.35k of a step's 19k cycles) cannot overlap the next block because the accumulator is one register tuple.
// Here a workgroup is 4 waves, one per SIMD, each with all 512 registers: a wave loads ITS static share of W once per
// launch (<= 92 fragment pairs = 368 registers) and keeps it; the matrix phase of a step is then straight-line code --
// ds_read_b128 of a Kstar fragment pair at an immediate offset, two MFMAs per row-block that needs the pair, one
// accumulator tuple PER row-block (so no epilogue sits between dependent MFMAs and the first MFMA of a block takes the
// inline constant 0 as C) -- with no global load, no scalar load, no address arithmetic and no branch.  A Kstar pair is
// read from LDS once per wave instead of once per row-block (100 instead of 364 KB per step).
//
// The work split is static (RwPlan): row-blocks in descending size go to the least loaded wave of the output's wave
// group; for n_s = 2, N = 200: waves {0,1} take output 0 as row-blocks {12,9,8,5,4,1,0} / {11,10,7,6,3,2} (92 / 90 pairs),
// waves {2,3} output 1 alike.  Everything else -- Kstar phase, finish(), refit prologue, data layout, results -- is
// cem_rollout_kernel's: the two kernels are interchangeable bit for bit in the Kstar values and the per-row dot
// products; only the ORDER in which a particle's row sums of squares are added differs (per row-block set here, per
// stage stream there), i.e. the variance agrees to rounding (~1e-16 relative), which the parity suite's 1e-9 covers.
//
// Instantiated per (n_s, n_u, n_pad / 16); the launcher takes it when W fits the register budget and falls back to
// cem_rollout_kernel otherwise (SX_ROLLOUT=stream forces the fallback, for A/B runs).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sx_amd.h"
#include "sx_exp2_tab2048.hpp"
#include "sx_gp.hpp"
#include "sx_reach.hpp"
#include "sx_refit.hpp"
#include "sx_rollout.hpp"
#include "sx_rw_launch.hpp"

// build-time switches for A/B runs (defaults = the product)
#ifndef SX_RW_PIPE
#define SX_RW_PIPE 1       // software-pipelined Kstar phase (gp_kstar_phase_pipe)
#endif
#ifndef SX_RW_LDSCONST
#define SX_RW_LDSCONST 1   // finish() reads its constants from LDS instead of (spilled) SGPRs
#endif
#ifndef SX_RW_POLYLANES
#define SX_RW_POLYLANES 1  // the state constraint's polytope rows on the four lane groups of wave 0 instead of a serial loop
#endif
#ifndef SX_RW_PF
#define SX_RW_PF 2         // Kstar fragment pairs in flight ahead of the MFMAs that consume them (3 spills at n_pad = 208)
#endif
#ifndef SX_RW_RCLOCAL
#define SX_RW_RCLOCAL 0    // n_s >= 3: finish() copies the reachability constants from LDS to registers in one batch (measured: no gain, 5 spilled dwords at <4,1,6>)
#endif
#ifndef SX_RW_POLYFOLD
#define SX_RW_POLYFOLD 0   // the cubic in w = r / u with the unit folded into its coefficients (one VALU instruction less per value)
#endif
#ifndef SX_RW_DIET
#define SX_RW_DIET 1       // Kstar phase with the expanded exponent, the 2048-entry table and no all-padding pairs (rw_kstar_phase)
#endif
#if SX_RW_PIPE
#define SX_RW_KSTAR gp_kstar_phase_pipe
#else
#define SX_RW_KSTAR gp_kstar_phase
#endif

#if SX_RW_DIET
#define RW_KSTAR_CALL(qb, qe) rw_kstar_phase(gc, kl, lds.kfrag, qb, qe, zq)
#else
#define RW_KSTAR_CALL(qb, qe) SX_RW_KSTAR(gc, lds, qb, qe, zq)
#endif

namespace sx {

constexpr int kRwThreads = 64 * kRwWaves;
constexpr int kRwMaxPairs = 93;   // 372 registers of W per wave; the rest of the 512 is the step's working set

// compile-time loop: f(std::integral_constant<int, I>{}) for I = B .. E - 1
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// Static assignment of the (output d, row-block rb) tasks to the 4 waves.
template <int NS, int NRB>
struct RwPlan {
    int wave_of[NS][NRB];
    int pairs[kRwWaves];
    constexpr RwPlan() : wave_of{}, pairs{} {
        for (int w = 0; w < kRwWaves; ++w) pairs[w] = 0;
        for (int rb = NRB - 1; rb >= 0; --rb)
            for (int d = 0; d < NS; ++d) {
                int lo = 0, hi = kRwWaves;
                if (kRwWaves % NS == 0) {   // whole wave groups per output: a Kstar pair is read by as few waves as possible
                    lo = d * (kRwWaves / NS);
                    hi = lo + kRwWaves / NS;
                }
                int best = lo;
                for (int w = lo + 1; w < hi; ++w)
                    if (pairs[w] < pairs[best]) best = w;
                wave_of[d][rb] = best;
                pairs[best] += 2 * (rb + 1);
            }
    }
    constexpr int max_pairs() const {
        int m = 0;
        for (int w = 0; w < kRwWaves; ++w) m = pairs[w] > m ? pairs[w] : m;
        return m;
    }
    constexpr bool has(int w, int d) const {
        for (int rb = 0; rb < NRB; ++rb)
            if (wave_of[d][rb] == w) return true;
        return false;
    }
    // pairs of Kstar_d the wave walks: up to its largest row-block of that output
    constexpr int qmax(int w, int d) const {
        for (int rb = NRB - 1; rb >= 0; --rb)
            if (wave_of[d][rb] == w) return 2 * (rb + 1);
        return 0;
    }
};

template <int NS, int NRB>
constexpr bool rw_fits() {
    return RwPlan<NS, NRB>{}.max_pairs() <= kRwMaxPairs;
}

// A wave's share of W in the order rw_mfma_phase consumes it -- canonical order (d, rb descending, q) -- as a table of pair
// offsets into a_pack: ONE load sequence for all waves, driven by a constant table (wave-uniform index, scalar loads).
// (The first version selected one of four compile-time load sequences with a switch over the wave; the compiler merged
// their tails and, at n_s = 1, left three of one wave's addresses undefined on the merged path: a memory fault.  A table
// leaves nothing to merge.)  Entries past a wave's share repeat pair 0: a valid address, a value never used.
template <int NS, int NRB, int MAXP>
struct RwLoadTab {
    int pair[kRwWaves][MAXP];
    constexpr RwLoadTab() : pair{} {
        const RwPlan<NS, NRB> plan{};
        constexpr int wpo = NRB * (NRB + 1);
        for (int w = 0; w < kRwWaves; ++w) {
            int idx = 0;
            for (int d = 0; d < NS; ++d)
                for (int rb = NRB - 1; rb >= 0; --rb) {
                    if (plan.wave_of[d][rb] != w) continue;
                    for (int q = 0; q < 2 * (rb + 1); ++q) pair[w][idx++] = d * wpo + rb * (rb + 1) + q;
                }
            for (; idx < MAXP; ++idx) pair[w][idx] = 0;
        }
    }
};

template <int NS, int D, int NRB, int MAXP>
__device__ __forceinline__ void rw_load_w(const GpConst<NS, D>& gc, int wave, int lane, v2d (&wreg)[MAXP]) {
    static constexpr RwLoadTab<NS, NRB, MAXP> tab{};
    const v2d* __restrict__ ap = reinterpret_cast<const v2d*>(gc.a_pack) + lane;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) wreg[i] = ap[(size_t)tab.pair[wave][i] * 64];
}

// The matrix phase of one step for wave WAVE: T = W_d Kstar_d^T for the wave's row-blocks, column sums of squares, the
// mean / Jacobian rows to LDS.  Pair-major: Kstar pair q of output d is read once and multiplied into every row-block of
// the wave that reaches it (k <= 16 rb + 15  <=>  q < 2 (rb + 1)).
template <int NS, int D, int NRB, int WAVE, int MAXP, int PF = SX_RW_PF>
__device__ __forceinline__ void rw_mfma_phase(const GpConst<NS, D>& gc, GpTileLds<NS, D>& lds, int lane,
                                              const v2d (&wreg)[MAXP]) {
    constexpr RwPlan<NS, NRB> plan{};
    const v2d* kbase = reinterpret_cast<const v2d*>(lds.kfrag) + lane;
    // register index of the first pair of (d, rb) in rw_load_w's order
    int first[NS][NRB];
    {
        int idx = 0;
#pragma unroll
        for (int d = 0; d < NS; ++d)
#pragma unroll
            for (int rb = NRB - 1; rb >= 0; --rb) {
                first[d][rb] = idx;
                if (plan.wave_of[d][rb] == WAVE) idx += 2 * (rb + 1);
            }
    }
    static_for<0, NS>([&](auto dtag) {
        constexpr int d = decltype(dtag)::value;
        constexpr int qmax = plan.qmax(WAVE, d);
        if constexpr (qmax > 0) {
        v4d acc[NRB];
        v2d b[PF + 1];
#pragma unroll
        for (int q = 0; q < PF && q < qmax; ++q) b[q] = kbase[(q * NS + d) * 64];
#pragma unroll
        for (int q = 0; q < qmax; ++q) {
            if (q + PF < qmax) b[(q + PF) % (PF + 1)] = kbase[((q + PF) * NS + d) * 64];
            SX_PIN();
            const v2d bq = b[q % (PF + 1)];
#pragma unroll
            for (int rb = NRB - 1; rb >= 0; --rb) {
                if (plan.wave_of[d][rb] != WAVE || q >= 2 * (rb + 1)) continue;
                const v2d a = wreg[first[d][rb] + q];
                if (q == 0)
                    acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bq.x, v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
                else
                    acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bq.x, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bq.y, acc[rb], 0, 0, 0);
            }
            SX_PIN();
        }
        // epilogues: independent chains, one per row-block
        double ssq = 0.0;
#pragma unroll
        for (int rb = NRB - 1; rb >= 0; --rb) {
            if (plan.wave_of[d][rb] != WAVE) continue;
            double s;
            if (rb >= NRB - 2) {
                // rows >= N live in the last row-block(s): N .. N + D are the mean / Jacobian rows, above is padding
                const int row0 = rb * 16 + (lane >> 4);
                s = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 4 * r;
                    if (row < gc.n_train)
                        s = fma(acc[rb][r], acc[rb][r], s);
                    else if (row - gc.n_train <= D)
                        lds.mj[d * 256 + (row - gc.n_train) * 16 + (lane & 15)] = acc[rb][r];
                }
            } else {
                s = fma(acc[rb][3], acc[rb][3], fma(acc[rb][2], acc[rb][2], fma(acc[rb][1], acc[rb][1], acc[rb][0] * acc[rb][0])));
            }
            ssq += s;
        }
        // the four lanes l, l ^ 16, l ^ 32, l ^ 48 hold the partial sums of one query point: ones(16 x 4) . B totals them
        const v4d tot = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, ssq, v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        if (lane < 16) lds.part[(WAVE * NS + d) * 16 + lane] = tot[0];
        }
    });
}

// ---------------------------------------------------------------------------------------------------------------
// The Kstar phase of the register-resident kernel.  A lone wave per SIMD issues an f64 VALU instruction every ~5.8
// cycles however many independent chains it has (tools/valu_probe3.hip: 2.7 ns; two waves per SIMD: 2.4 ns) and the
// pipelined loop of gp_kstar_phase_pipe already runs at that rate, so the phase only gets shorter with FEWER
// instructions per kernel value (85 per 2 rows x 2 outputs there):
//   * expanded exponent.  With the rows centred, x' = x - xbar, z' = z - xbar, and in table units (ln 2 / 2048):
//         y_d(k) = L_d + sum_j kk_dj (z'_j - x'_kj)^2 = [L_d + sum_j kk_dj z'_j^2] + [sum_j kk_dj x'_kj^2] + sum_j (-2 kk_dj z'_j) x'_kj
//                =            zz_d (per thread and step)  +   c_kd (per row, once)   +  sum_j g_dj x'_kj
//     one add and D fma per value instead of D (sub + mul) per row and D fma per value.  The centring keeps the
//     cancellation harmless: the terms are O(radius^2 / l^2), not O(|x|^2 / l^2).
//   * 2^(j/2048) table (16 KB of LDS, correctly rounded: sx_exp2_tab2048.hpp): |r| <= ln 2 / 4096, so a cubic suffices
//     (truncation r^4/24 = 3.4e-17): two instructions less per value than the 256-entry table's quartic.
//   * pairs of 8 rows that hold padding only (rows >= N) are not computed at all: their Kstar entries are zeroed once
//     (the columns of W they meet are zero, they only have to be finite).
// (The launcher hands this kernel k_nh_ils2 / k_log_os in units of ln 2 / 2048: 8 x GpConst's usual ones, exactly.)
// 70 VALU instructions per 2 rows x 2 outputs.  The values differ from gp_kstar_phase's in the last bits (<= ~4e-16
// relative for query points inside the data; the oracle comparisons allow 1e-9).
// ---------------------------------------------------------------------------------------------------------------
constexpr double kExpScale2 = 2048.0 / 0.693147180559945309417232;   // 2048 / ln 2 = 8 kExpScale
constexpr double kExpUnit2 = 0.693147180559945309417232 / 2048.0;

template <int NS, int D>
struct RwKstarLds {
    double* rows;   // [n_pad][D + NS]: centred training inputs x' and c_kd (rows >= N: zeros)
    double* tab;    // [2048] 2^(j/2048), then one NaN (the "table" of a NaN query point), one pad
    double* xbar;   // [D] the centre (+ pad to an even count)
    static __host__ __device__ size_t doubles(int n_pad) { return (size_t)n_pad * (D + NS) + (n_pad & 1) + kExpTab2 + 2 + ((D + 1) & ~1); }
    __device__ void carve(double* base, int n_pad) {
        rows = base;
        tab = rows + (size_t)n_pad * (D + NS) + (n_pad & 1);
        xbar = tab + kExpTab2 + 2;
    }
};

// once per workgroup (the caller barriers afterwards): centre, rows, table, zeroed padding pairs of the Kstar buffer
template <int NS, int D>
__device__ __forceinline__ void rw_kstar_setup(const GpConst<NS, D>& gc, const RwKstarLds<NS, D>& kl, double* kfrag) {
    const int tid = threadIdx.x, lane = tid & 63;
    // centre: mean of the first min(N, 64) rows -- any point inside the data cloud serves; every wave computes the same
    double xb[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double v = lane < gc.n_train ? gc.x_train[lane * D + j] : 0.0;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        xb[j] = v / (double)(gc.n_train < 64 ? gc.n_train : 64);
        if (tid == 0) kl.xbar[j] = xb[j];
    }
    for (int k = tid; k < gc.n_pad; k += blockDim.x) {
        double xr[D], cc[NS];
#pragma unroll
        for (int d = 0; d < NS; ++d) cc[d] = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            xr[j] = k < gc.n_train ? gc.x_train[k * D + j] - xb[j] : 0.0;
#pragma unroll
            for (int d = 0; d < NS; ++d) cc[d] = fma(gc.k_nh_ils2[d * D + j], xr[j] * xr[j], cc[d]);
        }
#pragma unroll
        for (int j = 0; j < D; ++j) kl.rows[k * (D + NS) + j] = xr[j];
#pragma unroll
        for (int d = 0; d < NS; ++d) kl.rows[k * (D + NS) + D + d] = cc[d];
    }
    for (int i = tid; i < kExpTab2; i += blockDim.x) kl.tab[i] = kExp2Tab2048[i];
    if (tid < 2) kl.tab[kExpTab2 + tid] = __builtin_nan("");
    // pairs past the last training row are never written by the Kstar phase
    const int first_pad = ((gc.n_train + 7) >> 3) * NS * 128;
    for (int i = first_pad + tid; i < (gc.n_pad >> 3) * NS * 128; i += blockDim.x) kfrag[i] = 0.0;
}

template <int NS, int D>
__device__ __forceinline__ void rw_kstar_phase(const GpConst<NS, D>& gc, const RwKstarLds<NS, D>& kl, double* kfrag,
                                               int q_begin, int q_end, const double (&z)[D]) {
    constexpr int M = 2 * NS, RS = D + NS;
    const int n = q_end - q_begin;
    if (n <= 0) return;
    const int lane = (int)threadIdx.x & 63;
    const int c = lane & 15;
    // per thread and step: z' = z - xbar, zz_d, g_dj
    double zz[NS], g[NS][D];
    bool znan = false;
    {
        double zc[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            zc[j] = z[j] - kl.xbar[j];
            znan = znan || (z[j] != z[j]);
        }
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            double a = gc.k_log_os[d];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double kz = gc.k_nh_ils2[d * D + j] * zc[j];
                a = fma(kz, zc[j], a);
                g[d][j] = -2.0 * kz;
            }
            zz[d] = a;
        }
    }
    // a NaN query point must give NaN rows: its thread reads the one-entry NaN table
    const lds_f64* etab = (const lds_f64*)kl.tab + (znan ? kExpTab2 : 0);
    const int emask = znan ? 0 : kExpTab2 - 1;
    const int k0 = 8 * q_begin + (lane >> 4);
    const lds_f64* x = (const lds_f64*)kl.rows + k0 * RS;
    lds_f64* f = (lds_f64*)kfrag + kfrag_index(NS, c, k0, 0);

    struct Set {
        double t[M], p[M];   // table entry (in flight after A1); r after A1, the polynomial after A2
        int mi[M];
    };
    auto load_x = [&](double (&xx)[2 * RS]) {
        asm volatile("" : "+v"(x));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < RS; ++j) xx[h * RS + j] = x[h * 4 * RS + j];
        x += 8 * RS;
    };
    auto stage_a1 = [&](const double (&xx)[2 * RS], Set& s) {
        double yc[M], m[M];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                double a = xx[h * RS + D + d] + zz[d];
#pragma unroll
                for (int j = 0; j < D; ++j) a = fma(g[d][j], xx[h * RS + j], a);
                yc[h * NS + d] = a;
            }
#pragma unroll
        for (int i = 0; i < M; ++i) yc[i] = __builtin_fmax(yc[i], -800.0 * kExpScale2);
#pragma unroll
        for (int i = 0; i < M; ++i) m[i] = __builtin_rint(yc[i]);
#pragma unroll
        for (int i = 0; i < M; ++i) asm("v_cvt_i32_f64 %0, %1" : "=v"(s.mi[i]) : "v"(m[i]));
#pragma unroll
        for (int i = 0; i < M; ++i) s.t[i] = etab[s.mi[i] & emask];
#pragma unroll
#if SX_RW_POLYFOLD
        for (int i = 0; i < M; ++i) s.p[i] = yc[i] - m[i];   // w = r / u, exact
#else
        for (int i = 0; i < M; ++i) s.p[i] = (yc[i] - m[i]) * kExpUnit2;   // r (the difference is exact)
#endif
    };
    auto stage_a2 = [&](Set& s) {
        double r[M];
#pragma unroll
        for (int i = 0; i < M; ++i) r[i] = s.p[i];
#pragma unroll
#if SX_RW_POLYFOLD
        // exp(u w) - 1 = w (u + w (u^2/2 + w u^3/6)): the unit folded into the coefficients, one multiplication less
        for (int i = 0; i < M; ++i) s.p[i] = fma(r[i], kExpUnit2 * kExpUnit2 * kExpUnit2 / 6.0, kExpUnit2 * kExpUnit2 * 0.5);
#pragma unroll
        for (int i = 0; i < M; ++i) s.p[i] = fma(s.p[i], r[i], kExpUnit2);
#else
        for (int i = 0; i < M; ++i) s.p[i] = fma(r[i], 1.66666666666666666667e-01, 0.5);
#pragma unroll
        for (int i = 0; i < M; ++i) s.p[i] = fma(s.p[i], r[i], 1.0);
#endif
#pragma unroll
        for (int i = 0; i < M; ++i) s.p[i] = s.p[i] * r[i];
    };
    auto stage_b = [&](const Set& s) {
        asm volatile("" : "+v"(f));
        double val[M];
#pragma unroll
        for (int i = 0; i < M; ++i) val[i] = ldexp(fma(s.t[i], s.p[i], s.t[i]), s.mi[i] >> 11);
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            f[d * 128] = val[d];
            f[d * 128 + 1] = val[NS + d];
        }
        f += NS * 128;
    };
    double xr[2 * RS];
    Set s0, s1;
    load_x(xr);           // trip 0
    stage_a1(xr, s0);
    load_x(xr);           // trip 1
    stage_a2(s0);
    auto body = [&](const Set& sold, Set& snew) {
        stage_a1(xr, snew);
        SX_PIN();
        load_x(xr);
        stage_b(sold);
        SX_PIN();
        stage_a2(snew);
        SX_PIN();
    };
    const int n1 = n - 1;
    for (int k = 0; k < (n1 >> 1); ++k) {
        body(s0, s1);
        body(s1, s0);
    }
    if (n1 & 1) {
        body(s0, s1);
        stage_b(s1);
    } else {
        stage_b(s0);
    }
}

// What finish() needs of the kernel arguments, copied to LDS once per workgroup.  As kernel arguments these ~50 doubles
// live in SGPRs, twice as many as a wave has: cem_rollout_kernel's finish() reads ~130 of them back from spill lanes
// (v_readlane) per call, on the pipe the phase is bound by.  There two waves share a SIMD and the Kstar waves need the
// time anyway; here wave 0 is alone on its SIMD and finish() IS the phase's critical path, so every instruction counts:
// from LDS a constant is one ds_read (no VALU slot), requested ahead of its use.
template <int NS, int NU>
struct alignas(16) RwConst {
    ReachConst<NS, NU> rc;
    CostConst<SX_MAX_M, NS, NU> cc;
    double inv_ls2[NS * (NS + NU)];
    double outputscale[NS];
    double noise[NS];
};

// gp_collect for the static split: only the waves that own row-blocks of output d hand in a partial sum
template <int NS, int D, int NRB, bool WITH_JAC, typename G>
__device__ __forceinline__ void rw_collect(const G& gc, const GpTileLds<NS, D>& lds, int c,
                                           const double (&z)[D], double (&mean)[NS], double (&var)[NS],
                                           double (&jac)[NS][D]) {
    constexpr RwPlan<NS, NRB> plan{};
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        double q = 0.0;
#pragma unroll
        for (int w = 0; w < kRwWaves; ++w)
            if (plan.has(w, d)) q += lds.part[(w * NS + d) * 16 + c];
        var[d] = (gc.outputscale[d] - q) + gc.noise[d];
        const double m = lds.mj[d * 256 + c];
        mean[d] = m;
        if constexpr (WITH_JAC) {
#pragma unroll
            for (int j = 0; j < D; ++j) jac[d][j] = lds.mj[d * 256 + (1 + j) * 16 + c] - z[j] * gc.inv_ls2[d * D + j] * m;
        }
    }
}

template <int NS, int NU, int NRB>
__global__ __launch_bounds__(kRwThreads) __attribute__((amdgpu_waves_per_eu(1, 1)))
void cem_rollout_rw_kernel(GpConst<NS, NS + NU> gc, ReachConst<NS, NU> rc, CostConst<SX_MAX_M, NS, NU> cc, RolloutPtrs rp) {
    constexpr int D = NS + NU;
    constexpr int S = NS + NS * NS;
    constexpr int PQS = S + 3;   // a particle's row in LDS: p, Q (row-major), objective cost, constraint cost, status bits
    constexpr RwPlan<NS, NRB> plan{};
    constexpr int MAXP = plan.max_pairs();
    static_assert(MAXP <= kRwMaxPairs, "W does not fit the register budget");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    GpTileLds<NS, D> lds;
    constexpr int nw = kRwWaves;
    double* acts = lds.carve(smem, gc.n_train, gc.n_pad, nw, NS);  // [16][H][NU]
    const int tid = threadIdx.x;
#if SX_RW_LDSCONST
    RwConst<NS, NU>* const cst = reinterpret_cast<RwConst<NS, NU>*>(acts + (((size_t)SX_TILE * rp.H * NU + 1) & ~(size_t)1));
    {
        static_assert(sizeof(RwConst<NS, NU>) % 8 == 0 && sizeof(ReachConst<NS, NU>) % 8 == 0 &&
                      sizeof(CostConst<SX_MAX_M, NS, NU>) % 8 == 0, "copied in 8-byte words");
        constexpr int nrc = sizeof(ReachConst<NS, NU>) / 8, ncc = sizeof(CostConst<SX_MAX_M, NS, NU>) / 8;
        double* dst = reinterpret_cast<double*>(cst);
        const double* src_rc = reinterpret_cast<const double*>(&rc);
        const double* src_cc = reinterpret_cast<const double*>(&cc);
        for (int i = tid; i < nrc; i += kRwThreads) dst[i] = src_rc[i];
        for (int i = tid; i < ncc; i += kRwThreads) dst[nrc + i] = src_cc[i];
        if (tid < NS * D) cst->inv_ls2[tid] = gc.inv_ls2[tid];
        if (tid < NS) {
            cst->outputscale[tid] = gc.outputscale[tid];
            cst->noise[tid] = gc.noise[tid];
        }
    }
    // The tile's per-particle state between steps, [16][PQS]: (p, Q), the cost sums and the status bits.  It lives in LDS,
    // not in registers: next to 368 registers of W the step loop has no room for values that idle through the Kstar and
    // matrix phases (17 registers per lane in every wave, though only 16 lanes of wave 0 use them); finish() reads them
    // at its start and writes them back, and the lanes that check the polytope rows read (p, Q) from here too.
    double* const pq = reinterpret_cast<double*>(cst + 1);
    const RwConst<NS, NU>& fc = *cst;            // finish()'s view of the constants
    const ReachConst<NS, NU>& frc = cst->rc;
    const CostConst<SX_MAX_M, NS, NU>& fcc = cst->cc;
#else
    double* const pq = acts + (((size_t)SX_TILE * rp.H * NU + 1) & ~(size_t)1);
    const GpConst<NS, D>& fc = gc;
    const ReachConst<NS, NU>& frc = rc;
    const CostConst<SX_MAX_M, NS, NU>& fcc = cc;
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int H = rp.H;
    const int tiles_per_problem = (rp.P + SX_TILE - 1) / SX_TILE;
    const int total_tiles = rp.E * tiles_per_problem;

    // this wave's share of W: requested first, it travels while X, the exp table and the first tile's prologue are set up
    v2d wreg[MAXP];
    rw_load_w<NS, D, NRB>(gc, wave, lane, wreg);
#if SX_RW_DIET
    RwKstarLds<NS, D> kl;
    kl.carve(pq + (((size_t)SX_TILE * PQS + 1) & ~(size_t)1), gc.n_pad);
    rw_kstar_setup(gc, kl, lds.kfrag);
    const int kstar_pairs = (gc.n_train + 7) >> 3;   // pairs of 8 rows with at least one training row
#else
    gp_load_xs(gc, lds);
    const int kstar_pairs = gc.n_pad >> 3;
#endif

    const bool owner = tid < SX_TILE;
    // Kstar shares (pairs of fragments).  Step 0: all waves alike.  From step 1 on wave 0 runs finish() meanwhile.
    int q0_begin, q0_end, q_begin = 0, q_end = 0;
    kstar_pair_range(kstar_pairs, wave, 1, nw, q0_begin, q0_end);
    if (wave > 0) kstar_pair_range(kstar_pairs, wave - 1, 1, nw - 1, q_begin, q_end);
    double* const zs_base = lds.zs;

    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int e = tile / tiles_per_problem;
        const int c0 = (tile - e * tiles_per_problem) * SX_TILE;  // first particle of the tile within problem e
        const bool valid = owner && (c0 + tid < rp.P);

        // ---- the sampling distribution and this tile's action sequences (as cem_rollout_kernel's prologue) ----
        const double* dist_mean = rp.mean + (int64_t)e * H * NU;
        const double* dist_std = rp.std + (int64_t)e * H * NU;
        if (rp.elite_rows) {
            const int L = H * NU, W = 2 + L;
            double* const ms = lds.kfrag;   // [2][L]: the Kstar buffer is free until the first step
            const double* rows = rp.elite_rows + (int64_t)e * rp.elite_k * W + 2;
            const bool publish = c0 == 0 && rp.mean_out;
            int cshift = 0;
            while ((nw << cshift) < L && cshift < 6) ++cshift;
            const int ccol = lane & ((1 << cshift) - 1);
            for (int cb = wave << cshift; cb < L; cb += nw << cshift) {
                const int col = cb + ccol;
                double m, sd;
                wave_refit_columns(rows + (col < L ? col : L - 1), rp.elite_k, W, lane, cshift, m, sd);
                if ((lane >> cshift) == 0 && col < L) {
                    ms[col] = m;
                    ms[L + col] = sd;
                    if (publish) {
                        rp.mean_out[(int64_t)e * L + col] = m;
                        rp.std_out[(int64_t)e * L + col] = sd;
                    }
                }
            }
            __syncthreads();
            dist_mean = ms;
            dist_std = ms + L;
        }
        for (int i = tid; i < SX_TILE * H * NU; i += kRwThreads) {
            const int c = i / (H * NU);
            const int r = i - c * (H * NU);
            double a = 0.0;
            if (c0 + c < rp.P) {
                const int64_t gi = ((int64_t)e * rp.P + c0 + c) * (H * NU) + r;
                if (rp.noise) {
                    a = dist_mean[r] + dist_std[r] * rp.noise[gi];
                    rp.actions[gi] = a;
                } else {
                    a = rp.actions[gi];
                }
            }
            acts[i] = a;
        }
        // per-particle state: LDS row of thread c (tid < 16)
        bool have_q = rp.q0 != nullptr;
        double* const my = pq + (owner ? tid : 0) * PQS;
#ifdef SX_STAMPS
        unsigned long long c_f_collect = 0, c_f_reach = 0;
#endif
        if (owner) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                my[i] = rp.x0[(int64_t)e * NS + i];
#pragma unroll
                for (int j = 0; j < NS; ++j) my[NS + i * NS + j] = have_q ? rp.q0[((int64_t)e * NS + i) * NS + j] : 0.0;
            }
            my[S] = 0.0;       // objective cost
            my[S + 1] = 0.0;   // constraint cost
            reinterpret_cast<int*>(my + S + 2)[0] = 0;   // status bits
        }
        __syncthreads();
        if (owner) {
#pragma unroll
            for (int i = 0; i < NS; ++i) lds.zs[tid * D + i] = my[i];
#pragma unroll
            for (int cidx = 0; cidx < NU; ++cidx) lds.zs[tid * D + NS + cidx] = acts[(tid * H + 0) * NU + cidx];
        }
        __syncthreads();

        // centre of particle c at step t >= 1 from z_{t-1} and the means of step t - 1 (cem_rollout_kernel's chain, bit for bit)
        auto next_centre = [&](const ReachConst<NS, NU>& R, int c, const double* z_prev, double (&out)[NS]) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                double s = lds.mj[i * 256 + c];
#pragma unroll
                for (int j = 0; j < NS; ++j) s = fma(R.a[i * NS + j], z_prev[j], s);
#pragma unroll
                for (int cidx = 0; cidx < NU; ++cidx) s = fma(R.b[i * NU + cidx], z_prev[NS + cidx], s);
                out[i] = s;
            }
        };
        auto finish = [&](int t) {
            double z[D], u[NU], mean[NS], var[NS], jac[NS][D], p1[NS], Q1[NS][NS];
            double p[NS], Q[NS][NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                p[i] = my[i];
#pragma unroll
                for (int j = 0; j < NS; ++j) Q[i][j] = my[NS + i * NS + j];
            }
            double obj = my[S], con = my[S + 1];
#pragma unroll
            for (int j = 0; j < NS; ++j) z[j] = p[j];
#pragma unroll
            for (int cidx = 0; cidx < NU; ++cidx) {
                u[cidx] = acts[(tid * H + t) * NU + cidx];
                z[NS + cidx] = u[cidx];
            }
            int st_step = 0;
            if (have_q) {
#ifdef SX_STAMPS
                const unsigned long long f0 = stamp();
#endif
                rw_collect<NS, D, NRB, true>(fc, lds, tid, z, mean, var, jac);
#ifdef SX_STAMPS
                const unsigned long long f1 = stamp();
#endif
#if SX_RW_RCLOCAL
                if constexpr (NS >= 3) {
                    // the reachability constants (~60 doubles at n_s = 4) in ONE batch of LDS reads at the head of the chain:
                    // left in LDS, each is requested where it is used, an exposed round trip of ~130 cycles every time
                    const ReachConst<NS, NU> rloc = frc;
                    SX_PIN();
                    reach_ellipsoid<NS, NU>(rloc, p, Q, u, mean, var, jac, p1, Q1, st_step);
                } else
#endif
                reach_ellipsoid<NS, NU>(frc, p, Q, u, mean, var, jac, p1, Q1, st_step);
#ifdef SX_STAMPS
                c_f_collect += f1 - f0;
                c_f_reach += stamp() - f1;
#endif
            } else {
                rw_collect<NS, D, NRB, false>(fc, lds, tid, z, mean, var, jac);
                reach_point<NS, NU>(frc, p, u, mean, var, p1, Q1, st_step);
            }
            have_q = true;
            {
                double zt[D];
#pragma unroll
                for (int j = 0; j < NS; ++j) zt[j] = p[j];
#pragma unroll
                for (int cidx = 0; cidx < NU; ++cidx) zt[NS + cidx] = u[cidx];
                next_centre(frc, tid, zt, p1);
                if (t + 1 < H) {
                    double* zn = zs_base + ((t + 1) & 1) * 16 * D + tid * D;
#pragma unroll
                    for (int i = 0; i < NS; ++i) zn[i] = p1[i];
#pragma unroll
                    for (int cidx = 0; cidx < NU; ++cidx) zn[NS + cidx] = acts[(tid * H + t + 1) * NU + cidx];
                }
            }
            if (valid && st_step) reinterpret_cast<int*>(my + S + 2)[0] |= st_step;
            obj += objective_cost<SX_MAX_M, NS, NU>(fcc, p1, var);
            bool uviol = false;
#pragma unroll
            for (int cidx = 0; cidx < NU; ++cidx) uviol = uviol || (u[cidx] < fcc.u_min[cidx]) || (u[cidx] > fcc.u_max[cidx]);
            if (uviol) con += SX_ACTION_VIOLATION_COST;
#if !SX_RW_POLYLANES
            if (fcc.con_mode == SX_CON_ALL_STATES || t == H - 1) {
                if (polytope_violated<SX_MAX_M, NS>(fcc.h_mat, fcc.h_vec, fcc.m, 1.0, p1, Q1, nullptr))
                    con += SX_STATE_VIOLATION_COST;
            }
#endif
            // the new state (with SX_RW_POLYLANES the state constraint is checked from here by all four lane groups of the
            // wave, one polytope row each: finish_polytope)
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                my[i] = p1[i];
#pragma unroll
                for (int j = 0; j < NS; ++j) my[NS + i * NS + j] = Q1[i][j];
            }
            my[S] = obj;
            my[S + 1] = con;
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
        };

#if SX_RW_POLYLANES
        // State constraint of step t (safempc_cem.py:102-132, gp_reachability_pytorch.py:184-231) on ALL 64 lanes of wave 0:
        // lane l checks polytope rows (l >> 4) + 4 i of particle l & 15 -- d = h.p + sqrt(h.Q h) - b >= 0 violates (a NaN
        // distance does not, as in the reference).  In finish() the rows were a serial loop on 16 lanes: m dependent sqrt
        // chains, ~48 instructions each, on the wave the phase waits for; here the pendulum's 4 rows are ONE pass.  The
        // verdicts of a particle's four lanes meet in a ballot (scalar unit).  Runs right after finish(t) on the same wave:
        // the LDS hand-over of (p1, Q1) needs no barrier (a wave's LDS operations complete in order).
        auto finish_polytope = [&](int t) {
            if (!(fcc.con_mode == SX_CON_ALL_STATES || t == H - 1)) return;
            const int c = lane & 15;
            const double* o = pq + c * PQS;
            double pp[NS], QQ[NS][NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                pp[i] = o[i];
#pragma unroll
                for (int j = 0; j < NS; ++j) QQ[i][j] = o[NS + i * NS + j];
            }
            bool viol = false;
            const int m = fcc.m;
            for (int r = lane >> 4; r < m; r += 4) {
                double hc = 0.0, hq = 0.0;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const double hi = fcc.h_mat[r * NS + i];
                    hc += hi * pp[i];
                    double sacc = 0.0;
#pragma unroll
                    for (int j = 0; j < NS; ++j) sacc += QQ[i][j] * fcc.h_mat[r * NS + j];
                    hq += hi * sacc;
                }
                const double dist = hc + sqrt(hq) - fcc.h_vec[r];
                viol = viol || (dist >= 0.0);
            }
            unsigned long long bits = __ballot(viol);
            bits |= bits >> 32;
            bits |= bits >> 16;
            if (owner && ((bits >> tid) & 1ull)) my[S + 1] += SX_STATE_VIOLATION_COST;
        };
#else
        auto finish_polytope = [&](int) {};
#endif

#ifdef SX_STAMPS
        unsigned long long c_k = 0, c_kb = 0, c_m = 0, c_mb = 0;
        c_f_collect = c_f_reach = 0;
        unsigned long long rt0;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
        const unsigned long long ct0 = stamp();
#endif
        for (int t = 0; t < H; ++t) {
#ifdef SX_STAMPS
            const unsigned long long t0 = stamp();
#endif
            const bool kstar_wave = !(t > 0 && wave == 0);
            if (kstar_wave) {
                double zq[D];
                const int c = lane & 15;
                if (t == 0) {
#pragma unroll
                    for (int j = 0; j < D; ++j) zq[j] = zs_base[c * D + j];
                    RW_KSTAR_CALL(q0_begin, q0_end);
                } else {
                    double pc[NS];
                    next_centre(rc, c, zs_base + ((t - 1) & 1) * 16 * D + c * D, pc);
#pragma unroll
                    for (int i = 0; i < NS; ++i) zq[i] = pc[i];
#pragma unroll
                    for (int cidx = 0; cidx < NU; ++cidx) zq[NS + cidx] = acts[(c * H + t) * NU + cidx];
                    RW_KSTAR_CALL(q_begin, q_end);
                }
            } else {
                if (owner) finish(t - 1);
                finish_polytope(t - 1);
            }
#ifdef SX_STAMPS
            const unsigned long long t1 = stamp();
#endif
            __syncthreads();
#ifdef SX_STAMPS
            const unsigned long long t2 = stamp();
#endif
            switch (wave) {
                case 0: rw_mfma_phase<NS, D, NRB, 0>(gc, lds, lane, wreg); break;
                case 1: rw_mfma_phase<NS, D, NRB, 1>(gc, lds, lane, wreg); break;
                case 2: rw_mfma_phase<NS, D, NRB, 2>(gc, lds, lane, wreg); break;
                default: rw_mfma_phase<NS, D, NRB, 3>(gc, lds, lane, wreg); break;
            }
#ifdef SX_STAMPS
            const unsigned long long t3 = stamp();
#endif
            __syncthreads();
#ifdef SX_STAMPS
            const unsigned long long t4 = stamp();
            c_k += t1 - t0; c_kb += t2 - t1; c_m += t3 - t2; c_mb += t4 - t3;
#endif
        }
#ifdef SX_STAMPS
        if (rp.stamps && lane == 0 && tile == (int)blockIdx.x) {
            unsigned long long* o = rp.stamps + ((size_t)blockIdx.x * nw + wave) * 8;
            o[0] = c_k; o[1] = c_kb; o[2] = c_m; o[3] = c_mb; o[4] = c_f_collect; o[5] = c_f_reach;   // (finish(): collect | reachability)
            unsigned long long rt1;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
            o[6] = stamp() - ct0;   // shader cycles of the step loop
            o[7] = rt1 - rt0;       // the same span in 10 ns ticks
        }
#endif
        if (owner) finish(H - 1);
        if (wave == 0) finish_polytope(H - 1);
        if (valid) {
            const int64_t g = (int64_t)e * rp.P + c0 + tid;
            rp.obj_cost[g] = my[S];
            rp.con_cost[g] = my[S + 1];
            const int st = reinterpret_cast<const int*>(my + S + 2)[0];
            if (st) atomicOr(rp.status, st);
        }
        __syncthreads();   // the next tile's prologue reuses the Kstar buffer and the action table
    }
}

}  // namespace sx
