// Per-particle ellipsoid algebra: one particle per lane, everything in registers.
// Restates (in our own form) safe_exploration/gp_reachability_pytorch.py:18-243, utils.py:152-194,
// utils_ellipsoid.py:102-140,282-309 of the reference; checked against oracle/reachability.py and the goldens.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/sx_amd.h"

namespace sx {

// Constants of one problem as the kernels see them (kernel argument, lives in SGPRs / constant cache).
template <int NS, int NU>
struct ReachConst {
    double a[NS * NS];
    double b[NS * NU];
    double kfb[NU * NS];
    double l_mu[NS];
    double l_sigma[NS];
    double cholB[NS * NS];  // lower Cholesky factor of B = I + kfb^T kfb (host-computed)
    double beta;
};

template <int M_MAX, int NS, int NU>
struct CostConst {
    double h_mat[M_MAX * NS];
    double h_vec[M_MAX];
    double u_min[NU];
    double u_max[NU];
    double w_abs[NS];
    double target[NS];
    double w_lin[NS];
    int m;
    int obj_mode;
    int con_mode;
};

// _fix_zeros_nans (gp_reachability_pytorch.py:234-243).  The reference decides on the whole batch: a NaN anywhere fails;
// an exact zero ANYWHERE lifts every x <= 0 to 1e-5; without one, a negative goes on to sqrt -> NaN -> fails at the next
// check.  `clamp_nonpos` carries the batch's "an exact zero is present" flag:
//   * sx_onestep_reach computes it over the batch in a pre-pass (batch_zero_flag_kernel) -- the reference's rule exactly;
//   * the fused rollout cannot see the other workgroups' particles of the same step and passes false: exact zeros are
//     lifted, a negative always takes the sqrt -> NaN route.  The two rules differ only when a solve reports BOTH
//     SX_STATUS_NAN and SX_STATUS_ZERO_FIX; FusedCemMpc then repeats the solve through the step-by-step path
//     (sx_gp_predict + sx_onestep_reach), which decides as the reference does (DESIGN.md "Deviations").
__device__ __forceinline__ double fix_zero_nan(double x, int& status, bool clamp_nonpos = false) {
    if (x != x) status |= SX_STATUS_NAN;
    if (x == 0.0 || (clamp_nonpos && x <= 0.0)) {
        status |= SX_STATUS_ZERO_FIX;
        x = 1e-5;
    }
    return x;
}

// Largest eigenvalue of the symmetric matrix S (only the lower triangle is trusted).
template <int NS>
__device__ __forceinline__ double sym_lambda_max(double (&S)[NS][NS]) {
    if constexpr (NS == 1) {
        return S[0][0];
    } else if constexpr (NS == 2) {
        const double h = 0.5 * (S[0][0] - S[1][1]);
        const double o = 0.5 * (S[0][1] + S[1][0]);
        return 0.5 * (S[0][0] + S[1][1]) + sqrt(h * h + o * o);
    } else {
        // cyclic Jacobi, fixed sweep count (quadratic convergence: 8 sweeps reach round-off for n <= 4).  This is the longest
        // dependent chain of a step for n_s >= 3 (round 3: ~20k of the 31k cycles of a cart-pole step at small N), so
        //  * a rotation costs one sqrt, one division and one reciprocal square root: with a = a_qq - a_pp, b = 2 a_pq,
        //    t = sgn(a) b / (|a| + sqrt(a^2 + b^2)), c = rsqrt(t^2 + 1), s = t c  (the textbook form theta = a / b,
        //    t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), c = 1 / sqrt(t^2 + 1) is three divisions and two roots);
        //  * for n = 4 the sweep runs in the round-robin order (0,1)(2,3) | (0,2)(1,3) | (0,3)(1,2): the two rotations of a
        //    round touch disjoint rows and columns, so both angles come from the same matrix and their chains overlap.
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int j = i + 1; j < NS; ++j) {
                const double s = 0.5 * (S[i][j] + S[j][i]);
                S[i][j] = s;
                S[j][i] = s;
            }
        // (c, s) of the rotation that annihilates S[p][q]; identity when it would not change anything.  Branch-free (the chains
        // of a round's two rotations can only overlap in straight-line code): the arithmetic runs on whatever is there and
        // a select takes the identity; `live` says whether any rotation of the sweep was a real one.
        bool live = false;
        double tr = 0.0;     // sum of |diagonal| at the start of the sweep
        auto angle = [&](int p, int q, double& c, double& sn, double& tn) {
            const double apq = S[p][q], app = S[p][p], aqq = S[q][q];
            // An off-diagonal entry below 1e-14 of the TRACE is left alone: all six together move no eigenvalue by more than
            // 6e-14 of the trace (Weyl) -- lambda_max >= trace / n by 2.4e-13 of itself at worst, in practice by the entry's
            // square over the gap.  (The first version compared with 1e-19 of the two diagonal entries involved: below what
            // rounding leaves behind after every sweep, so no sweep ever came out clean and all eight always ran, most of
            // them resolving the SMALL eigenvalues of an elongated ellipsoid, which lambda_max does not need.)
            const bool on = fabs(apq) > 1e-300 && fabs(apq) > 1e-14 * tr;
            const double a = aqq - app, b = 2.0 * apq;
            const double t = (a < 0.0 ? -b : b) / (fabs(a) + sqrt(fma(a, a, b * b)));
            const double cc = rsqrt(fma(t, t, 1.0));
            c = on ? cc : 1.0;
            sn = on ? t * cc : 0.0;
            tn = on ? t : 0.0;
            live = live || on;
        };
        // S <- J^T S J for the rotation J in the (p, q) plane, as the symmetric update it is: the two rows / columns outside
        // the plane (2 entries each for n = 4), the two diagonal entries (a_pp - t a_pq, a_qq + t a_pq) and a_pq = 0 -- a
        // dozen operations where the two-sided product is 64; on one wave per SIMD every f64 operation is ~6 cycles of issue.
        auto rotate = [&](int p, int q, double c, double sn, double tn) {
            const bool on = sn != 0.0;      // (a skipped rotation leaves every entry as it is, a NaN neighbour included)
            const double apq = S[p][q];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (k == p || k == q) continue;
                const double skp = S[k][p], skq = S[k][q];
                const double np_ = on ? c * skp - sn * skq : skp;
                const double nq_ = on ? sn * skp + c * skq : skq;
                S[k][p] = np_;
                S[p][k] = np_;
                S[k][q] = nq_;
                S[q][k] = nq_;
            }
            S[p][p] = on ? S[p][p] - tn * apq : S[p][p];
            S[q][q] = on ? S[q][q] + tn * apq : S[q][q];
            S[p][q] = on ? 0.0 : apq;
            S[q][p] = S[p][q];
        };
        for (int sweep = 0; sweep < 8; ++sweep) {
            tr = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) tr += fabs(S[i][i]);
            if constexpr (NS == 4) {
                constexpr int kRound[3][4] = {{0, 1, 2, 3}, {0, 2, 1, 3}, {0, 3, 1, 2}};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double c0, s0, t0, c1, s1, t1;
                    angle(kRound[r][0], kRound[r][1], c0, s0, t0);
                    angle(kRound[r][2], kRound[r][3], c1, s1, t1);
                    rotate(kRound[r][0], kRound[r][1], c0, s0, t0);
                    rotate(kRound[r][2], kRound[r][3], c1, s1, t1);
                }
            } else {
#pragma unroll
                for (int p = 0; p < NS - 1; ++p)
#pragma unroll
                    for (int q = p + 1; q < NS; ++q) {
                        double c, sn, tn;
                        angle(p, q, c, sn, tn);
                        rotate(p, q, c, sn, tn);
                    }
            }
            if (!live) break;       // a sweep without a real rotation: converged (every later sweep would do nothing either)
            live = false;
        }
        double m = S[0][0];
#pragma unroll
        for (int i = 1; i < NS; ++i) m = fmax(m, S[i][i]);
        return m;
    }
}

// r^2 = lambda_max(Q B), B = I + kfb^T kfb = L L^T  ==  lambda_max(L^T Q L)   (utils.py:175-185)
template <int NS, int NU>
__device__ __forceinline__ double remainder_r2(const ReachConst<NS, NU>& rc, const double (&Q)[NS][NS]) {
    double T[NS][NS];  // Q L
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < NS; ++k) s += Q[i][k] * rc.cholB[k * NS + j];
            T[i][j] = s;
        }
    double S[NS][NS];  // L^T (Q L)
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = i; k < NS; ++k) s += rc.cholB[k * NS + i] * T[k][j];
            S[i][j] = s;
        }
    return sym_lambda_max<NS>(S);
}

// Point branch (gp_reachability_pytorch.py:65-99).  var is fixed up in place and is what the caller reports as sigma.
template <int NS, int NU>
__device__ __forceinline__ void reach_point(const ReachConst<NS, NU>& rc, const double (&p)[NS], const double (&u)[NU],
                                            const double (&mean)[NS], double (&var)[NS], double (&p1)[NS],
                                            double (&Q1)[NS][NS], int& status, bool batch_zero = false) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        var[i] = fix_zero_nan(var[i], status, batch_zero);
        double rk = rc.beta * sqrt(var[i]);
        rk = fix_zero_nan(rk, status);
        if (!(rk > 0.0)) status |= SX_STATUS_UB_NONPOS;
#pragma unroll
        for (int j = 0; j < NS; ++j) Q1[i][j] = 0.0;
        Q1[i][i] = NS * rk * rk;
        double s = mean[i];
#pragma unroll
        for (int j = 0; j < NS; ++j) s += rc.a[i * NS + j] * p[j];
#pragma unroll
        for (int c = 0; c < NU; ++c) s += rc.b[i * NU + c] * u[c];
        p1[i] = s;
    }
}

// Ellipsoid branch (gp_reachability_pytorch.py:100-181).  jac is [NS][NS+NU].
template <int NS, int NU>
__device__ __forceinline__ void reach_ellipsoid(const ReachConst<NS, NU>& rc, const double (&p)[NS],
                                                const double (&Q)[NS][NS], const double (&u)[NU],
                                                const double (&mean)[NS], double (&var)[NS],
                                                const double (&jac)[NS][NS + NU], double (&p1)[NS],
                                                double (&Q1)[NS][NS], int& status, bool batch_zero = false) {
    // H = a + J_x + (J_u + b) k_fb                                             (:131)
    double Hm[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            double s = rc.a[i * NS + j] + jac[i][j];
#pragma unroll
            for (int c = 0; c < NU; ++c) s += (jac[i][NS + c] + rc.b[i * NU + c]) * rc.kfb[c * NS + j];
            Hm[i][j] = s;
        }
    // Q0 = H Q H^T                                                             (:134)
    double T[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NS; ++k) s += Q[i][k] * Hm[j][k];
            T[i][j] = s;  // Q H^T
        }
    double Q0[NS][NS];
    double trQ0 = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NS; ++k) s += Hm[i][k] * T[k][j];
            Q0[i][j] = s;
        }
        trQ0 += Q0[i][i];
    }
    // Lagrange remainder boxes                                                 (:145-162, utils.py:152-194)
    const double r2 = remainder_r2<NS, NU>(rc, Q);
    const double r1 = sqrt(r2);
    double dsig[NS], dmu[NS];
    double trSig = 0.0, trMu = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        var[i] = fix_zero_nan(var[i], status, batch_zero);
        double bs = rc.beta * (sqrt(var[i]) + rc.l_sigma[i] * r1);
        bs = fix_zero_nan(bs, status);
        const double um = rc.l_mu[i] * r2;
        if (!(bs > 0.0) || !(um > 0.0)) status |= SX_STATUS_UB_NONPOS;
        dsig[i] = NS * bs * bs;
        dmu[i] = NS * um * um;
        trSig += dsig[i];
        trMu += dmu[i];
    }
    // (Q_sigma (+) Q_mu) (+) Q0                                                (:169-172, utils_ellipsoid.py:102-140)
    const double c1 = sqrt(trSig / trMu);
    const double f1 = 1.0 + 1.0 / c1, g1 = 1.0 + c1;
    double dsum[NS];
    double trSum = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        dsum[i] = f1 * dsig[i] + g1 * dmu[i];
        trSum += dsum[i];
    }
    const double c2 = sqrt(trSum / trQ0);
    const double f2 = 1.0 + 1.0 / c2, g2 = 1.0 + c2;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int j = 0; j < NS; ++j) Q1[i][j] = g2 * Q0[i][j];
        Q1[i][i] = f2 * dsum[i] + g2 * Q0[i][i];
        // p0 = mean + a p + b u                                                (:132)
        double s = mean[i];
#pragma unroll
        for (int j = 0; j < NS; ++j) s += rc.a[i * NS + j] * p[j];
#pragma unroll
        for (int c = 0; c < NU; ++c) s += rc.b[i * NU + c] * u[c];
        p1[i] = s;
    }
}

// true if the ellipsoid is NOT certified inside: some d_j >= 0 (a NaN distance is "inside", as in the reference's
// `(d >= 0).sum() == 0`, gp_reachability_pytorch.py:229-231).  Optionally stores the distances.
template <int M_MAX, int NS>
__device__ __forceinline__ bool polytope_violated(const double* h_mat, const double* h_vec, int m, double c_safety,
                                                  const double (&p)[NS], const double (&Q)[NS][NS], double* d_out) {
    bool viol = false;
    for (int r = 0; r < m; ++r) {
        double hc = 0.0, hq = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const double hi = h_mat[r * NS + i];
            hc += hi * p[i];
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < NS; ++j) s += Q[i][j] * h_mat[r * NS + j];
            hq += hi * s;
        }
        const double d = hc + c_safety * sqrt(hq) - h_vec[r];
        if (d_out) d_out[r] = d;
        viol = viol || (d >= 0.0);
    }
    return viol;
}

template <int M_MAX, int NS, int NU>
__device__ __forceinline__ double objective_cost(const CostConst<M_MAX, NS, NU>& cc, const double (&p1)[NS],
                                                 const double (&var)[NS]) {
    double o = 0.0;
    if (cc.obj_mode == SX_OBJ_NEG_VARIANCE) {
#pragma unroll
        for (int i = 0; i < NS; ++i) o -= var[i];
    } else {
#pragma unroll
        for (int i = 0; i < NS; ++i) o += cc.w_abs[i] * fabs(cc.target[i] - p1[i]) + cc.w_lin[i] * p1[i];
    }
    return o;
}

}  // namespace sx
