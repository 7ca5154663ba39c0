// GP posterior for one tile of SX_TILE (=16) query points per workgroup, on the f64 matrix cores.
//
// For output d the predictive variance is  s_d + noise_d - || W_d k*_d ||^2  with  W_d = chol(K_d + noise_d I)^-1
// (lower triangular), so the dense work is the triangular product  T = W_d . Kstar_d^T  ([N x N] x [N x 16]) followed
// by a column sum of squares.  Mean and mean-Jacobian are 1 + D more rows of the same product, stored in the rows
// N .. N + D that the padding of W_d to a multiple of 16 leaves free:
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
#include <type_traits>
#include "../../include/sx_amd.h"

namespace sx {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__host__ __device__ inline int64_t w_pairs_per_output(int nrb) { return (int64_t)nrb * (nrb + 1); }
__host__ __device__ inline int64_t w_pack_doubles(int n_s, int n_pad) {
    return (int64_t)n_s * w_pairs_per_output(n_pad / 16) * 128;
}
__host__ __device__ inline int64_t a_pack_doubles(int n_s, int n_pad) { return w_pack_doubles(n_s, n_pad); }
// Rows N .. N + D of the padded W_d hold the mean / Jacobian rows R_d (they ride along in the last row-block(s) for
// free), so the padding must leave room for 1 + D of them.
__host__ __device__ inline int gp_n_pad(int n_train, int d_in) { return (n_train + 1 + d_in + 15) / 16 * 16; }

// index of element (c in 0..15, k) inside a fragment-ordered strip
__device__ __forceinline__ int frag_index(int c, int k) {
    return (((k >> 3) * 64 + ((k & 3) << 4) + c) << 1) + ((k >> 2) & 1);
}

// 2^(j/256), j = 0..255, correctly rounded (generated with mpmath at 200 bits).
constexpr int kExpTab = 256;
__device__ const double kExp2Tab[kExpTab] = {
    1.0, 1.0027112750502025, 1.0054299011128027, 1.0081558981184175,
    1.0108892860517005, 1.0136300849514894, 1.016378314910953, 1.019133996077738,
    1.0218971486541166, 1.0246677928971357, 1.0274459491187637, 1.030231637686041,
    1.0330248790212284, 1.0358256936019572, 1.0386341019613787, 1.041450124688316,
    1.0442737824274138, 1.0471050958792898, 1.0499440858006872, 1.0527907730046264,
    1.0556451783605572, 1.0585073227945128, 1.061377227289262, 1.0642549128844645,
    1.0671404006768237, 1.0700337118202419, 1.0729348675259756, 1.075843889062791,
    1.0787607977571199, 1.0816856149932152, 1.0846183622133092, 1.0875590609177697,
    1.0905077326652577, 1.0934643990728858, 1.0964290818163769, 1.099401802630222,
    1.102382583307841, 1.1053714457017412, 1.1083684117236787, 1.1113735033448175,
    1.1143867425958924, 1.1174081515673693, 1.1204377524096067, 1.12347556733302,
    1.1265216186082418, 1.129575928566288, 1.1326385195987192, 1.1357094141578055,
    1.1387886347566916, 1.1418762039695616, 1.1449721444318042, 1.148076478840179,
    1.1511892299529827, 1.154310420590216, 1.1574400736337511, 1.1605782120274988,
    1.1637248587775775, 1.1668800369524817, 1.1700437696832502, 1.1732160801636373,
    1.1763969916502812, 1.1795865274628758, 1.182784710984341, 1.1859915656609938,
    1.189207115002721, 1.1924313825831512, 1.1956643920398273, 1.1989061670743806,
    1.202156731452703, 1.2054161090051239, 1.2086843236265816, 1.2119613992768012,
    1.215247359980469, 1.2185422298274085, 1.2218460329727576, 1.2251587936371455,
    1.22848053610687, 1.2318112847340759, 1.2351510639369334, 1.2384998981998165,
    1.241857812073484, 1.245224830175258, 1.2486009771892048, 1.2519862778663162,
    1.255380757024691, 1.2587844395497165, 1.2621973503942507, 1.2656195145788063,
    1.2690509571917332, 1.2724917033894028, 1.275941778396392, 1.2794012075056693,
    1.2828700160787783, 1.2863482295460256, 1.2898358734066657, 1.2933329732290895,
    1.2968395546510096, 1.3003556433796506, 1.3038812651919358, 1.3074164459346773,
    1.3109612115247644, 1.3145155879493546, 1.318079601266064, 1.3216532776031575,
    1.3252366431597413, 1.3288297242059544, 1.3324325470831615, 1.3360451382041458,
    1.339667524053303, 1.3432997311868353, 1.3469417862329458, 1.3505937158920345,
    1.3542555469368927, 1.3579273062129011, 1.3616090206382248, 1.365300717204012,
    1.3690024229745905, 1.3727141650876684, 1.3764359707545302, 1.380167867260238,
    1.383909881963832, 1.387662042298529, 1.3914243757719262, 1.3951969099662003,
    1.3989796725383112, 1.4027726912202048, 1.4065759938190154, 1.4103896082172707,
    1.4142135623730951, 1.4180478843204152, 1.4218926021691656, 1.4257477441054942,
    1.42961333839197, 1.433489413367789, 1.4373759974489824, 1.4412731191286257,
    1.4451808069770467, 1.449099089642035, 1.4530279958490526, 1.4569675544014438,
    1.460917794180647, 1.4648787441464057, 1.4688504333369818, 1.4728328908693675,
    1.4768261459394993, 1.4808302278224719, 1.4848451658727524, 1.488870989524397,
    1.4929077282912648, 1.4969554117672355, 1.5010140696264256, 1.5050837316234065,
    1.5091644275934228, 1.5132561874526098, 1.5173590411982147, 1.5214730189088146,
    1.5255981507445384, 1.529734466947287, 1.533881997840956, 1.5380407738316568,
    1.5422108254079407, 1.5463921831410214, 1.550584877685, 1.5547889397770887,
    1.559004400237837, 1.5632312899713576, 1.567469639965553, 1.5717194812923414,
    1.5759808451078865, 1.5802537626528246, 1.5845382652524937, 1.588834384317164,
    1.593142151342267, 1.597461597908627, 1.6017927556826934, 1.606135656416771,
    1.6104903319492543, 1.6148568142048607, 1.6192351351948637, 1.6236253270173289,
    1.6280274218573478, 1.632441451987275, 1.6368674497669644, 1.6413054476440063,
    1.645755478153965, 1.6502175739206177, 1.6546917676561943, 1.6591780921616162,
    1.6636765803267364, 1.6681872651305825, 1.6727101796415966, 1.6772453570178785,
    1.681792830507429, 1.6863526334483934, 1.6909247992693053, 1.6955093614893326,
    1.7001063537185235, 1.7047158096580513, 1.709337763100463, 1.713972247929926,
    1.718619298122478, 1.723278947746274, 1.7279512309618377, 1.732636182022311,
    1.7373338352737062, 1.7420442251551564, 1.746767386199169, 1.7515033530318782,
    1.7562521603732995, 1.761013843037584, 1.7657884359332727, 1.7705759740635547,
    1.7753764925265212, 1.7801900265154245, 1.785016611318935, 1.789856282321401,
    1.7947090750031072, 1.7995750249405351, 1.804454167806624, 1.809346539371032,
    1.8142521755003989, 1.8191711121586085, 1.8241033854070534, 1.8290490314048973,
    1.8340080864093424, 1.8389805867758937, 1.843966568958626, 1.8489660695104508,
    1.8539791250833855, 1.8590057724288205, 1.864046048397789, 1.8690999899412386,
    1.8741676341103, 1.8792490180565602, 1.8843441790323345, 1.8894531543909392,
    1.8945759815869656, 1.8997126981765553, 1.9048633418176741, 1.9100279502703899,
    1.9152065613971474, 1.9203992131630474, 1.925605943636125, 1.930826790987627,
    1.9360617934922943, 1.9413109895286405, 1.9465744175792332, 1.9518521162309783,
    1.9571441241754002, 1.9624504802089273, 1.9677712232331759, 1.9731063922552343,
    1.978456026387951, 1.9838201648502194, 1.9891988469672663, 1.9945921121709402,
};

constexpr double kExpScale = 256.0 / 0.693147180559945309417232;   // 256 / ln 2
constexpr double kExpUnit = 0.693147180559945309417232 / 256.0;    // ln 2 / 256

// e^x for M arguments at once, table-driven; the caller passes y = x * 256 / ln 2 (it folds the factor into the
// constants that produce x).  y = 256 n + j + f, |f| <= 1/2, so e^x = 2^n T[j] e^r with r = f ln2/256, |r| <= ln2/512:
// T from LDS (`tab`, 256 doubles), e^r - 1 by a degree-4 polynomial (truncation 4e-17).  14 VALU instructions and one
// LDS read per value (a polynomial-only exp with the same accuracy needs 19); error <= ~1 ulp on top of the rounding of y itself
// (|x| 2e-16, the same as rounding x would cost).  Below e^-800 the result is 0 whatever the argument (also -inf, also
// 1e300 lengthscales away): the clamp keeps the reduction exact for every input.  A NaN argument does NOT give NaN
// (max drops it): callers that must propagate NaN pass a table of NaNs (gp_kstar_phase does).  The M chains advance
// in lock-step: a dependent f64 op issues every ~9 cycles, an independent one every ~4 (tools/valu_probe.hip), and the
// f64 VALU is the unit this kernel's MFMAs compete with.
template <int M>
__device__ __forceinline__ void exp_tab_f64_n(const double (&y)[M], double (&out)[M],
                                              const __attribute__((address_space(3))) double* tab) {
    double m[M], r[M], t[M], p[M], yc[M];
    int mi[M];
#pragma unroll
    for (int i = 0; i < M; ++i) yc[i] = __builtin_fmax(y[i], -800.0 * kExpScale);
#pragma unroll
    for (int i = 0; i < M; ++i) m[i] = __builtin_rint(yc[i]);
#pragma unroll
    for (int i = 0; i < M; ++i) asm("v_cvt_i32_f64 %0, %1" : "=v"(mi[i]) : "v"(m[i]));
#pragma unroll
    for (int i = 0; i < M; ++i) t[i] = tab[mi[i] & (kExpTab - 1)];
#pragma unroll
    for (int i = 0; i < M; ++i) r[i] = (yc[i] - m[i]) * kExpUnit;   // the difference is exact
#pragma unroll
    for (int i = 0; i < M; ++i) p[i] = fma(r[i], 4.16666666666666666667e-02, 1.66666666666666666667e-01);
#pragma unroll
    for (int i = 0; i < M; ++i) p[i] = fma(p[i], r[i], 0.5);
#pragma unroll
    for (int i = 0; i < M; ++i) p[i] = fma(p[i], r[i], 1.0);
#pragma unroll
    for (int i = 0; i < M; ++i) p[i] = p[i] * r[i];
#pragma unroll
    for (int i = 0; i < M; ++i) out[i] = ldexp(fma(t[i], p[i], t[i]), mi[i] >> 8);
}

template <int NS, int D>
struct GpConst {
    double inv_ls2[NS * D];
    double nh_ils2[NS * D];   // -1 / (2 l^2)
    double log_os[NS];        // ln(outputscale)
    double k_nh_ils2[NS * D];  // the same two, times 256 / ln 2: the fused Kstar phase computes the exponent in units of
    double k_log_os[NS];       // ln 2 / 256 (exp_tab_f64_n)
    double outputscale[NS];
    double noise[NS];
    const double* x_train;
    const double* a_pack;    // W_d fragments, then (at pair offset NS * w_pairs_per_output) the mean/Jacobian rows
    const int4* stage_tab;   // [nw] headers {stage count}, then [nw][stage_cap] stage descriptors
    int n_train;
    int n_pad;
    int stage_cap;      // gp_stage_cap(NS, n_pad, waves per workgroup)
    int stage_cap_one;  // gp_stage_cap(1, ...): the per-output streams of the output-by-output rollout, which follow the
                        // combined table in the same buffer: table d at stage_tab + nw (1 + stage_cap) + d nw (1 + stage_cap_one)
};

// LDS carve-up of one GP tile (all in doubles, 16-byte aligned pieces)
template <int NS, int D>
struct GpTileLds {
    double* xs;     // [n_pad x D]        training inputs (rows >= n_train zero), loaded once per kernel
    double* kfrag;  // [NS][n_pad x 16]   Kstar in fragment order
    double* mj;     // [NS][16 rows][16]  mean / Jacobian rows
    double* part;   // [NW][NS][16]       per-wave partial sums of squares
    double* zs;     // [2][16][D]         query points (the rollout alternates between the halves; predict uses the first)
    double* etab;   // [512]              2^(j/256) for exp_tab_f64_n, then 256 NaNs (the table of a NaN query point)
    // ns_lds: outputs whose Kstar is in LDS at the same time (NS, or 1 in the output-by-output rollout)
    __device__ double* carve(double* base, int n_train, int n_pad, int nw, int ns_lds = NS) {
        xs = base;
        kfrag = xs + ((n_pad * D + 1) & ~1);
        mj = kfrag + (size_t)ns_lds * n_pad * 16;
        part = mj + NS * 256;
        zs = part + (size_t)nw * NS * 16;
        etab = zs + 32 * D;
        return etab + 2 * kExpTab;
    }
};

// MFMA work decomposition shared by host (table sizing) and device (stream construction).
//   task j : row-block rb = nrb - 1 - j / NS of output d = j % NS (descending cost)       2 (rb + 1) fragment pairs
// Every task is cut into stages of 2 pairs (4 MFMAs; pair counts are even).  Tasks go to the waves by greedy
// longest-processing-time assignment.  (Handing the second-dispatched half of the workgroup less work does not help:
// it loses the arbitration for its SIMD's matrix pipe, but the phase ends when the SIMD's TOTAL is done -- measured.)
__host__ __device__ inline int gp_task_pairs(int j, int ns, int nrb) { return 2 * (nrb - j / ns); }
constexpr int kStagePad = 8;  // dummy descriptors behind a wave's stream: prefetches past the end stay in bounds
constexpr int kMaxWaves = 16;
__host__ __device__ inline int gp_wave_speed(int, int) { return 100; }
// Replays the assignment; returns the wave of task `upto` and leaves the per-wave stage counts before it in `load`.
__host__ __device__ inline int gp_assign(int ns, int nrb, int nw, int upto, int* load) {
    for (int w = 0; w < nw; ++w) load[w] = 0;
    int wave = 0;
    for (int j = 0; j <= upto; ++j) {
        const int cost = gp_task_pairs(j, ns, nrb) / 2;
        long best = -1;
        for (int w = 0; w < nw; ++w) {
            const long finish = (long)(load[w] + cost) * 1000 / gp_wave_speed(w, nw);
            if (best < 0 || finish < best) {
                best = finish;
                wave = w;
            }
        }
        if (j < upto) load[wave] += cost;
    }
    return wave;
}
inline int gp_stage_cap(int ns, int n_pad, int nw) {
    const int nrb = n_pad >> 4, ntask = ns * nrb;
    int load[kMaxWaves];
    const int w_last = gp_assign(ns, nrb, nw, ntask - 1, load);
    load[w_last] += gp_task_pairs(ntask - 1, ns, nrb) / 2;
    int cap = 0;
    for (int w = 0; w < nw; ++w) cap = load[w] > cap ? load[w] : cap;
    return cap + kStagePad;
}

// the combined table, then one table per output
inline int64_t gp_stage_tab_ints(int ns, int n_pad, int nw) {
    return 4 * (int64_t)nw * ((1 + gp_stage_cap(ns, n_pad, nw)) + (int64_t)ns * (1 + gp_stage_cap(1, n_pad, nw)));
}

inline size_t gp_tile_lds_doubles(int ns, int d, int n_train, int n_pad, int nw, int ns_lds = -1) {
    if (ns_lds < 0) ns_lds = ns;
    return (size_t)((n_pad * d + 1) & ~1) + (size_t)ns_lds * n_pad * 16 + ns * 256 + (size_t)nw * ns * 16 + 32 * d + 2 * kExpTab;
}

template <int NS, int D>
__device__ __forceinline__ void gp_load_xs(const GpConst<NS, D>& gc, GpTileLds<NS, D>& lds) {
    for (int i = threadIdx.x; i < gc.n_pad * D; i += blockDim.x) lds.xs[i] = i < gc.n_train * D ? gc.x_train[i] : 0.0;
    if (threadIdx.x < kExpTab) {
        lds.etab[threadIdx.x] = kExp2Tab[threadIdx.x];
        lds.etab[kExpTab + threadIdx.x] = __builtin_nan("");
    }
}

// Phase 1: Kstar_d[c][k] = s_d exp(-1/2 sum_j (z_cj - X_kj)^2 / l_dj^2) for the tile's 16 points, all k < n_pad, all d.
// The rows are dealt out in PAIRS of fragments (8 rows: k = 8 q .. 8 q + 7, n_pad / 8 of them); the calling wave
// works off the pairs [q_begin, q_end), one pair per trip: lane (c = lane & 15, kk = lane >> 4) computes rows 8 q + kk
// and 8 q + kk + 4, the two slots of ONE 16-byte fragment element.  Waves may own different numbers of pairs
// (kstar_pair_range): in the rollout wave 0 finishes the previous step meanwhile and the wave that shares its SIMD
// gets half a share.
//
// The phase is bound by VALU issue, on the pipe the f64 MFMAs use too (tools/overlap_probe.hip), so what counts is the
// instruction count per value.  Columns k >= n_train of W and of the mean/Jacobian rows are zero (pack_a_kernel), so the
// padding entries of Kstar only have to be finite: they are computed like any other from the zero rows gp_load_xs
// appends to X -- no index clamp, no select; and from trip to trip both the X rows and the fragment element move by a
// constant.
// Kstar lives in LDS as [pair q = k >> 3][output d][lane = ((k & 3) << 4) + c][slot = (k >> 2) & 1]: one ds_read_b128
// per lane feeds two MFMAs of one output, and the NS values a Kstar thread produces for one (c, k) are a constant
// 1 KB apart (an immediate offset of the store).
__device__ __forceinline__ int kfrag_index(int ns, int c, int k, int d) {
    return ((((k >> 3) * ns + d) * 64 + ((k & 3) << 4) + c) << 1) + ((k >> 2) & 1);
}

// pairs [begin, end) of worker `w` out of `total` pairs, for workers of the given weights (cumulative weight before w,
// own weight, weight sum): proportional shares, every pair exactly once
__device__ __forceinline__ void kstar_pair_range(int total, int before, int weight, int wsum, int& begin, int& end) {
    begin = total * before / wsum;
    end = total * (before + weight) / wsum;
}

typedef __attribute__((address_space(3))) double lds_f64;

#define SX_PIN() __builtin_amdgcn_sched_barrier(0)

template <int NS, int D>
__device__ __forceinline__ void gp_kstar_phase(const GpConst<NS, D>& gc, GpTileLds<NS, D>& lds, int q_begin, int q_end,
                                               const double (&z)[D]) {
    // z: the query point of this thread's c = lane & 15 (the caller loads or derives it)
    const int lane = (int)threadIdx.x & 63;
    const int c = lane & 15;
    // loop invariants the compiler would otherwise re-materialise from SGPRs on every trip (VOP3 takes one SGPR)
    double log_os[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        log_os[d] = gc.k_log_os[d];
        asm volatile("" : "+v"(log_os[d]));
    }
    // a NaN query point must give NaN rows (the status word reports it downstream): its thread reads the NaN table
    bool znan = false;
#pragma unroll
    for (int j = 0; j < D; ++j) znan = znan || (z[j] != z[j]);
    const lds_f64* etab = (const lds_f64*)lds.etab + (znan ? kExpTab : 0);

    // 32-bit LDS pointers, advanced by a constant per trip and hidden from the optimiser, which would otherwise turn
    // them back into base + offset and spend an add per access
    const int k0 = 8 * q_begin + (lane >> 4);
    const lds_f64* x = (const lds_f64*)lds.xs + k0 * D;
    lds_f64* f = (lds_f64*)lds.kfrag + kfrag_index(NS, c, k0, 0);
    for (int q = q_begin; q < q_end; ++q) {
        asm volatile("" : "+v"(x), "+v"(f));
        // the two rows of this thread at once: 2 NS independent exp chains in flight
        double arg[2 * NS], val[2 * NS];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double sq[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double df = z[j] - x[h * 4 * D + j];
                sq[j] = df * df;
            }
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                double a = log_os[d];  // s_d exp(-q/2) = exp(ln s_d - q/2), in units of ln 2 / 256
#pragma unroll
                for (int j = 0; j < D; ++j) a = fma(sq[j], gc.k_nh_ils2[d * D + j], a);
                arg[h * NS + d] = a;
            }
        }
        exp_tab_f64_n<2 * NS>(arg, val, etab);
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            f[d * 128] = val[d];           // slot 0: row 8 q + kk
            f[d * 128 + 1] = val[NS + d];  // slot 1: row 8 q + kk + 4
        }
        x += 8 * D;
        f += NS * 128;
    }
}

// gp_kstar_phase, software-pipelined for ONE wave per SIMD (the register-resident rollout, sx_rollout_rw.hpp).  With two
// waves per SIMD the two LDS round trips of a trip (the X rows, then the 2^(j/256) table) hide behind the other wave's
// arithmetic; a lone wave waits them out: 600 cycles per trip measured against 340 of issue.  Here a trip is split in
// three stages that overlap across trips -- the X rows of trip i + 2 are requested while trip i + 1 computes its
// exponents and requests its table entries (A1), trip i is finished and stored (B), and trip i + 1 evaluates its
// polynomial (A2) -- so nothing waits for a load that was not issued a stage earlier.  Two register sets alternate (no
// moves); the X rows need one set only (the next trip's are requested as soon as A1 has consumed this trip's).  The arithmetic, operation for operation, is exp_tab_f64_n's: the values are bit-identical to gp_kstar_phase's.
// Reads up to one trip (8 rows of X) past q_end: rows past n_pad are the start of the Kstar buffer (finite or not,
// their results are never stored).
template <int NS, int D>
__device__ __forceinline__ void gp_kstar_phase_pipe(const GpConst<NS, D>& gc, GpTileLds<NS, D>& lds, int q_begin, int q_end,
                                                    const double (&z)[D]) {
    constexpr int M = 2 * NS;
    const int n = q_end - q_begin;
    if (n <= 0) return;
    const int lane = (int)threadIdx.x & 63;
    const int c = lane & 15;
    double log_os[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        log_os[d] = gc.k_log_os[d];
        asm volatile("" : "+v"(log_os[d]));
    }
    bool znan = false;
#pragma unroll
    for (int j = 0; j < D; ++j) znan = znan || (z[j] != z[j]);
    const lds_f64* etab = (const lds_f64*)lds.etab + (znan ? kExpTab : 0);
    const int k0 = 8 * q_begin + (lane >> 4);
    const lds_f64* x = (const lds_f64*)lds.xs + k0 * D;
    lds_f64* f = (lds_f64*)lds.kfrag + kfrag_index(NS, c, k0, 0);

    struct Set {
        double t[M], p[M];   // table entry (in flight after A1); r = y - rint(y) after A1, the polynomial after A2
        int mi[M];
    };
    auto load_x = [&](double (&xx)[2 * D]) {
        asm volatile("" : "+v"(x));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < D; ++j) xx[h * D + j] = x[h * 4 * D + j];
        x += 8 * D;
    };
    auto stage_a1 = [&](const double (&xx)[2 * D], Set& s) {
        double yc[M], m[M];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double sq[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double df = z[j] - xx[h * D + j];
                sq[j] = df * df;
            }
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                double a = log_os[d];
#pragma unroll
                for (int j = 0; j < D; ++j) a = fma(sq[j], gc.k_nh_ils2[d * D + j], a);
                yc[h * NS + d] = a;
            }
        }
#pragma unroll
        for (int i = 0; i < M; ++i) yc[i] = __builtin_fmax(yc[i], -800.0 * kExpScale);
#pragma unroll
        for (int i = 0; i < M; ++i) m[i] = __builtin_rint(yc[i]);
#pragma unroll
        for (int i = 0; i < M; ++i) asm("v_cvt_i32_f64 %0, %1" : "=v"(s.mi[i]) : "v"(m[i]));
#pragma unroll
        for (int i = 0; i < M; ++i) s.t[i] = etab[s.mi[i] & (kExpTab - 1)];
#pragma unroll
        for (int i = 0; i < M; ++i) s.p[i] = (yc[i] - m[i]) * kExpUnit;   // r (the difference is exact)
    };
    auto stage_a2 = [&](Set& s) {
        double r[M];
#pragma unroll
        for (int i = 0; i < M; ++i) r[i] = s.p[i];
#pragma unroll
        for (int i = 0; i < M; ++i) s.p[i] = fma(r[i], 4.16666666666666666667e-02, 1.66666666666666666667e-01);
#pragma unroll
        for (int i = 0; i < M; ++i) s.p[i] = fma(s.p[i], r[i], 0.5);
#pragma unroll
        for (int i = 0; i < M; ++i) s.p[i] = fma(s.p[i], r[i], 1.0);
#pragma unroll
        for (int i = 0; i < M; ++i) s.p[i] = s.p[i] * r[i];
    };
    auto stage_b = [&](const Set& s) {
        asm volatile("" : "+v"(f));
        double val[M];
#pragma unroll
        for (int i = 0; i < M; ++i) val[i] = ldexp(fma(s.t[i], s.p[i], s.t[i]), s.mi[i] >> 8);
#pragma unroll
        for (int d = 0; d < NS; ++d) {
            f[d * 128] = val[d];
            f[d * 128 + 1] = val[NS + d];
        }
        f += NS * 128;
    };
    double xr[2 * D];     // the X rows of the next trip to enter A1
    Set s0, s1;
    load_x(xr);           // trip 0
    stage_a1(xr, s0);
    load_x(xr);           // trip 1
    stage_a2(s0);
    // body: sold = trip i (A1 + A2 done), snew <- trip i + 1 (its rows are in xr), xr <- rows of trip i + 2
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

// The same for ONE output DD, into a Kstar buffer that holds a single output ([pair][lane][slot]): the
// output-by-output rollout for training sets whose n_s Kstar buffers do not fit in LDS together.
template <int NS, int D, int DD>
__device__ __forceinline__ void gp_kstar_phase_one(const GpConst<NS, D>& gc, GpTileLds<NS, D>& lds, int q_begin, int q_end,
                                                   const double (&z)[D]) {
    const int lane = (int)threadIdx.x & 63;
    const int c = lane & 15;
    double log_os = gc.k_log_os[DD];
    asm volatile("" : "+v"(log_os));
    bool znan = false;
#pragma unroll
    for (int j = 0; j < D; ++j) znan = znan || (z[j] != z[j]);
    const lds_f64* etab = (const lds_f64*)lds.etab + (znan ? kExpTab : 0);
    const int k0 = 8 * q_begin + (lane >> 4);
    const lds_f64* x = (const lds_f64*)lds.xs + k0 * D;
    lds_f64* f = (lds_f64*)lds.kfrag + kfrag_index(1, c, k0, 0);
    for (int q = q_begin; q < q_end; ++q) {
        asm volatile("" : "+v"(x), "+v"(f));
        double arg[2], val[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double a = log_os;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double df = z[j] - x[h * 4 * D + j];
                a = fma(df * df, gc.k_nh_ils2[DD * D + j], a);
            }
            arg[h] = a;
        }
        exp_tab_f64_n<2>(arg, val, etab);
        f[0] = val[0];
        f[1] = val[1];
        x += 8 * D;
        f += 128;
    }
}

// Phase 2: the triangular products on the matrix cores.
//
// A wave's share of the work is static, so sx_gp_pack writes it once (build_stage_tab_kernel) to global memory as a
// flat stream of stage descriptors
//     int4 { x: pair offset into a_pack, y: v2d offset into kfrag, z: output d | row-block << 8, w: flags }
// flags: 2 = last stage of its task, 4 = the row-block reaches past row N (mean/Jacobian rows, zero padding).
// One stage = 2 fragment pairs = 4 MFMAs = 256 matrix-core cycles.  Four register sets rotate, so the A fragments
// (global: L2-resident W) and B fragments (LDS: Kstar) of stages i+1 .. i+3 are in flight while stage i computes --
// across task boundaries too.
//
// On gfx950 the f64 MFMA does NOT hide VALU work of its own wave (measured, tools/mfma_probe2.hip: every VALU
// instruction placed between two v_mfma_f64_16x16x4 costs ~8 cycles, v_readfirstlane ~28), so the stream is driven
// by the SCALAR unit: descriptors arrive by s_load (uniform global loads), the A address is an SGPR base + a constant
// per-lane offset, and the only vector instruction per stage besides loads and MFMAs is the B address add.
constexpr int kStageLast = 2, kStageExtra = 4;

// one thread per wave (warm path): replay the assignment and write that wave's stream.
// tab = [nw] headers {stage count}, then [nw][stage_cap] descriptors
// The stream covers the outputs d_first .. d_first + ns - 1, whose Kstar sits in LDS side by side ([pair][ns][lane]):
// all of them in the fused kernel (d_first = 0), one at a time in the output-by-output kernel (ns = 1).  `ns_total`
// is the model's output count (it sets where an output's W fragments start).
__device__ __forceinline__ void gp_build_stage_tab(int4* tab, int ns, int n_train, int n_pad, int nw, int stage_cap,
                                                   int wave, int d_first = 0) {
    const int nrb = n_pad >> 4;
    const int ntask = ns * nrb;
    const int wpo = (int)w_pairs_per_output(nrb);
    int4* out = tab + nw + (size_t)wave * stage_cap;
    int load[kMaxWaves];
    for (int w = 0; w < nw; ++w) load[w] = 0;
    int pos = 0;
    for (int j = 0; j < ntask; ++j) {
        const int npairs = gp_task_pairs(j, ns, nrb);
        const int cost = npairs / 2;
        long best = -1;
        int target = 0;
        for (int w = 0; w < nw; ++w) {
            const long finish = (long)(load[w] + cost) * 1000 / gp_wave_speed(w, nw);
            if (best < 0 || finish < best) {
                best = finish;
                target = w;
            }
        }
        load[target] += cost;
        if (target != wave) continue;
        const int rb = nrb - 1 - j / ns;
        const int dl = j % ns;          // position of the output's Kstar in LDS
        const int d = d_first + dl;     // the output itself
        const int a0 = d * wpo + rb * (rb + 1);
        const int extra = (16 * rb + 15 >= n_train) ? kStageExtra : 0;  // block holds mean/Jacobian (or padding) rows
        for (int q = 0; q < npairs; q += 2, ++pos)
            out[pos] = int4{a0 + q, (q * ns + dl) * 64, d | (rb << 8), extra | ((q + 2 >= npairs) ? kStageLast : 0)};
    }
    // dummy descriptors (valid addresses, never computed on) behind the stream
    for (int i = 0; i < kStagePad; ++i) out[pos + i] = int4{0, 0, 0, 0};
    tab[wave] = int4{pos, 0, 0, 0};
}

struct MfmaStage {
    v2d a0, a1, b0, b1;
    int z, w;
};


// What a wave needs to open its MFMA phase: the stage count, the descriptors of its first four stages and the W
// fragments of the first three.  The stream is static, so a kernel fetches all of it ONCE and keeps it in registers:
// neither the scalar-load chain (header -> descriptor) nor the L2 round trip for W is paid per phase any more.
struct MfmaHead {
    v2d a[3][2];
    int nst;
    int y[3], z[3], w[3];
    int4 desc3;
};

template <int NS, int D>
__device__ __forceinline__ MfmaHead gp_mfma_head(const GpConst<NS, D>& gc, const int4* __restrict__ stage_tab, int wave,
                                                 int nw, int lane, int stage_cap) {
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    // (a stream is followed by kStagePad valid dummy descriptors, so four stages can always be requested)
    const int4* __restrict__ stages = stage_tab + nw + (size_t)swave * stage_cap;
    const __amdgpu_buffer_rsrc_t arsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(gc.a_pack), 0, (int)0xffffffffu, 0x00020000);
    MfmaHead h;
    h.nst = __builtin_amdgcn_readfirstlane(stage_tab[swave].x);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int4 dsc = stages[i];
        const int aoff = dsc.x << 10;
        h.a[i][0] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(arsrc, lane * 16, aoff, 0));
        h.a[i][1] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(arsrc, lane * 16, aoff + 1024, 0));
        h.y[i] = __builtin_amdgcn_readfirstlane(dsc.y);
        h.z[i] = __builtin_amdgcn_readfirstlane(dsc.z);
        h.w[i] = __builtin_amdgcn_readfirstlane(dsc.w);
    }
    const int4 d3 = stages[3];
    h.desc3 = int4{__builtin_amdgcn_readfirstlane(d3.x), __builtin_amdgcn_readfirstlane(d3.y),
                   __builtin_amdgcn_readfirstlane(d3.z), __builtin_amdgcn_readfirstlane(d3.w)};
    return h;
}

// NSL: outputs side by side in the LDS Kstar (NS, or 1); only_d >= 0: the stream holds that output alone, and only its
// partial sum is written.
template <int NS, int D, int NSL = NS>
__device__ __forceinline__ void gp_mfma_phase(const GpConst<NS, D>& gc, const int4* __restrict__ stage_tab,
                                              GpTileLds<NS, D>& lds, int wave, int nw, int lane,
                                              const MfmaHead& head, int stage_cap, int only_d = -1) {
    double ssq[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) ssq[d] = 0.0;

    const int swave = __builtin_amdgcn_readfirstlane(wave);
    // stage_tab is a __restrict__ kernel argument of its own: the loads below are provably unclobbered and uniform,
    // which is what lets the compiler issue them as s_load
    const int4* __restrict__ stages = stage_tab + nw + (size_t)swave * stage_cap;
    const int nst = head.nst;
    // A fragments through a buffer descriptor: address = SGPR base + SGPR stage offset + constant per-lane offset,
    // so a load needs no vector arithmetic at all
    const __amdgpu_buffer_rsrc_t arsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(gc.a_pack), 0, (int)0xffffffffu, 0x00020000);
    const int lane16 = lane * 16;
    const v2d* kbase = reinterpret_cast<const v2d*>(lds.kfrag) + lane;

    // `desc` always holds the descriptor of the next stage to be issued, fetched one issue earlier.
    int4 desc = head.desc3;
    int aoff;
    const v2d* bp;
    // the same, split so that step() can spread it: an f64 MFMA leaves its wave only ~4 free issue slots
    // (tools/mfma_probe2.hip), so the ~13 non-MFMA instructions of a stage go 4 / 4 / 3 / 2 between the four MFMAs
    auto decode_a = [&](MfmaStage& st) {
        aoff = desc.x << 10;
        st.z = desc.z;
        st.w = desc.w;
    };
    auto decode_b = [&](int inext) {
        bp = kbase + desc.y;
        desc = stages[inext + 1];
    };
    auto load_a = [&](int byte_off) -> v2d {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 raw = __builtin_amdgcn_raw_buffer_load_b128(arsrc, lane16, aoff + byte_off, 0);
        return __builtin_bit_cast(v2d, raw);
    };
    auto issue = [&](MfmaStage& st, int i) {  // prologue form: everything but Kstar is resident (gp_mfma_head)
        st.z = head.z[i];
        st.w = head.w[i];
        st.a0 = head.a[i][0];
        st.a1 = head.a[i][1];
        const v2d* b = kbase + head.y[i];
        st.b0 = b[0];
        st.b1 = b[64 * NSL];
    };

    // one accumulator: a dependent chain of this MFMA issues at the full rate (tools/mfma_probe.hip)
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    // compute stage `cur` while putting stage `inext` in flight into `nx`
    auto step = [&](const MfmaStage& cur, MfmaStage& nx, int inext) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.a0.x, cur.b0.x, acc, 0, 0, 0);
        SX_PIN();
        decode_a(nx);
        nx.a0 = load_a(0);
        SX_PIN();
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.a0.y, cur.b0.y, acc, 0, 0, 0);
        SX_PIN();
        nx.a1 = load_a(1024);
        decode_b(inext);
        SX_PIN();
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.a1.x, cur.b1.x, acc, 0, 0, 0);
        SX_PIN();
        nx.b0 = bp[0];
        nx.b1 = bp[64 * NSL];
        SX_PIN();
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.a1.y, cur.b1.y, acc, 0, 0, 0);
        SX_PIN();
        if (__builtin_expect(cur.w & kStageLast, 0)) {  // unlikely: the common path falls through
            const int d = cur.z & 255;
            double s;
            if (cur.w & kStageExtra) {
                // the lane holds rows row0 + 4 r of its query point: rows < N are W rows, rows N .. N + D are the
                // mean / Jacobian rows (to LDS), anything above is zero padding
                const int row0 = (cur.z >> 8) * 16 + (lane >> 4);
                s = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 4 * r;
                    if (row < gc.n_train)
                        s = fma(acc[r], acc[r], s);
                    else if (row - gc.n_train <= D)
                        lds.mj[d * 256 + (row - gc.n_train) * 16 + (lane & 15)] = acc[r];
                }
            } else {
                s = fma(acc[3], acc[3], fma(acc[2], acc[2], fma(acc[1], acc[1], acc[0] * acc[0])));
            }
#pragma unroll
            for (int dd = 0; dd < NS; ++dd) ssq[dd] = fma(s, (d == dd) ? 1.0 : 0.0, ssq[dd]);  // d is scalar: no branch
            acc = v4d{0.0, 0.0, 0.0, 0.0};
        }
    };

    if (nst > 0) {
        MfmaStage s0, s1, s2, s3;
        issue(s0, 0);
        issue(s1, 1);
        issue(s2, 2);
        // steady state without conditional paths, so that the compiler's counted s_waitcnt keep all three
        // prefetched stages in flight; the 1-3 leftover stages are peeled off behind it
        int i = 0;
        for (; i + 4 <= nst; i += 4) {
            step(s0, s3, i + 3);
            step(s1, s0, i + 4);
            step(s2, s1, i + 5);
            step(s3, s2, i + 6);
        }
        if (i < nst) {
            step(s0, s3, i + 3);
            if (i + 1 < nst) {
                step(s1, s0, i + 4);
                if (i + 2 < nst) step(s2, s1, i + 5);
            }
        }
    }
    // The four lanes l, l ^ 16, l ^ 32, l ^ 48 hold the partial sums of one query point.  Lane l's value is exactly the
    // B operand element [k = l >> 4][col = l & 15], so ones(16 x 4) . B puts the column total into every lane: one
    // more MFMA per output instead of two dependent cross-lane shuffles through LDS.
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        if (only_d >= 0 && only_d != d) continue;   // (uniform)
        const v4d tot = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, ssq[d], v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        if (lane < 16) lds.part[(wave * NS + d) * 16 + lane] = tot[0];
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
