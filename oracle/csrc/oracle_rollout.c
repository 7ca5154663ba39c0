/*
 * Oracle, C restatement: the CEM particle rollout of the reference, one particle at a time, OpenMP over particles.
 * TEST INFRASTRUCTURE (see oracle/__init__.py): used as the checker in tests and as the `cpu_baseline` of bench.py,
 * never by the product path.  Plain C99; build: make -C oracle/csrc  ->  oracle/csrc/liboracle.so
 *
 * Follows (paths relative to the reference root):
 *   dynamics callback            safe_exploration/safempc_cem.py:288-312
 *   onestep_reachability         safe_exploration/gp_reachability_pytorch.py:18-181, _fix_zeros_nans :234-243
 *   remainder over-approximation safe_exploration/utils.py:152-194
 *   ellipsoid sum / from box     safe_exploration/utils_ellipsoid.py:102-140, 282-309
 *   polytope test                safe_exploration/gp_reachability_pytorch.py:184-231
 *   exact GP posterior           closed form of what gp_ssm_cem.py:59-94 asks gpytorch for (noise included)
 * It is pinned the same way as the numpy oracle: against the goldens generated from the reference (tests/golden).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXS 8

typedef struct {
    int n_s, n_u, n, m, obj_mode, con_mode;
    double beta;
    const double *x;      /* [n x D] */
    const double *chol;   /* [n_s x n x n] lower Cholesky factors of K_d + noise_d I */
    const double *alpha;  /* [n_s x n] */
    const double *ls;     /* [n_s x D] lengthscales */
    const double *os;     /* [n_s] outputscale */
    const double *noise;  /* [n_s] */
    const double *a, *b, *kfb, *l_mu, *l_sigma;
    const double *h_mat, *h_vec, *u_min, *u_max;
    const double *w_abs, *target, *w_lin;
} sxo_problem;

/* largest eigenvalue of the symmetric n x n matrix S (cyclic Jacobi); S is destroyed */
static double sym_lambda_max(double *S, int n) {
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += S[p * n + q] * S[p * n + q];
        if (off < 1e-300) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = S[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (S[q * n + q] - S[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double skp = S[k * n + p], skq = S[k * n + q];
                    S[k * n + p] = c * skp - s * skq;
                    S[k * n + q] = s * skp + c * skq;
                }
                for (int k = 0; k < n; ++k) {
                    const double spk = S[p * n + k], sqk = S[q * n + k];
                    S[p * n + k] = c * spk - s * sqk;
                    S[q * n + k] = s * spk + c * sqk;
                }
            }
    }
    double mx = S[0];
    for (int i = 1; i < n; ++i) mx = fmax(mx, S[i * n + i]);
    return mx;
}

/* one particle, H steps.  work: 2 n doubles.  Returns the status bits (1 NaN, 2 zero fix, 4 box bound <= 0). */
static int rollout_one(const sxo_problem *pr, int H, const double *x0, const double *q0, const double *actions,
                       double *traj_p, double *traj_q, double *sigma, double *obj_out, double *con_out, double *work) {
    const int ns = pr->n_s, nu = pr->n_u, D = ns + nu, n = pr->n;
    double p[MAXS], Q[MAXS * MAXS], z[2 * MAXS], mean[MAXS], var[MAXS], jac[MAXS * 2 * MAXS];
    double *ks = work, *v = work + n;
    int have_q = q0 != NULL, status = 0;
    double obj = 0.0, con = 0.0;
    memcpy(p, x0, sizeof(double) * ns);
    if (have_q) memcpy(Q, q0, sizeof(double) * ns * ns);
    /* B = I + kfb^T kfb and its Cholesky factor */
    double Bm[MAXS * MAXS], Lb[MAXS * MAXS];
    for (int i = 0; i < ns; ++i)
        for (int j = 0; j < ns; ++j) {
            double s = (i == j);
            for (int c = 0; c < nu; ++c) s += pr->kfb[c * ns + i] * pr->kfb[c * ns + j];
            Bm[i * ns + j] = s;
        }
    memset(Lb, 0, sizeof(Lb));
    for (int j = 0; j < ns; ++j) {
        double s = Bm[j * ns + j];
        for (int k = 0; k < j; ++k) s -= Lb[j * ns + k] * Lb[j * ns + k];
        Lb[j * ns + j] = sqrt(s);
        for (int i = j + 1; i < ns; ++i) {
            double t = Bm[i * ns + j];
            for (int k = 0; k < j; ++k) t -= Lb[i * ns + k] * Lb[j * ns + k];
            Lb[i * ns + j] = t / Lb[j * ns + j];
        }
    }
    for (int t = 0; t < H; ++t) {
        const double *u = actions + (size_t)t * nu;
        for (int i = 0; i < ns; ++i) z[i] = p[i];
        for (int c = 0; c < nu; ++c) z[ns + c] = u[c];
        /* GP posterior at z */
        for (int d = 0; d < ns; ++d) {
            const double *ls = pr->ls + d * D, *Ld = pr->chol + (size_t)d * n * n, *al = pr->alpha + (size_t)d * n;
            double m = 0.0;
            double jd[2 * MAXS];
            for (int j = 0; j < D; ++j) jd[j] = 0.0;
            for (int k = 0; k < n; ++k) {
                double q = 0.0;
                for (int j = 0; j < D; ++j) {
                    const double df = (z[j] - pr->x[(size_t)k * D + j]) / ls[j];
                    q += df * df;
                }
                ks[k] = pr->os[d] * exp(-0.5 * q);
                const double wk = ks[k] * al[k];
                m += wk;
                for (int j = 0; j < D; ++j) jd[j] += wk * (pr->x[(size_t)k * D + j] - z[j]) / (ls[j] * ls[j]);
            }
            double qf = 0.0;
            for (int i = 0; i < n; ++i) { /* forward substitution L v = k* */
                double s = ks[i];
                const double *row = Ld + (size_t)i * n;
                for (int k = 0; k < i; ++k) s -= row[k] * v[k];
                v[i] = s / row[i];
                qf += v[i] * v[i];
            }
            mean[d] = m;
            var[d] = pr->os[d] - qf + pr->noise[d];
            for (int j = 0; j < D; ++j) jac[d * D + j] = jd[j];
        }
        /* reachability step */
        double p1[MAXS], Q1[MAXS * MAXS];
        for (int i = 0; i < ns; ++i) {
            double s = mean[i];
            for (int j = 0; j < ns; ++j) s += pr->a[i * ns + j] * p[j];
            for (int c = 0; c < nu; ++c) s += pr->b[i * nu + c] * u[c];
            p1[i] = s;
            if (var[i] != var[i]) status |= 1;
            if (var[i] == 0.0) { var[i] = 1e-5; status |= 2; }
        }
        memset(Q1, 0, sizeof(double) * ns * ns);
        if (!have_q) {
            for (int i = 0; i < ns; ++i) {
                double rk = pr->beta * sqrt(var[i]);
                if (rk != rk) status |= 1;
                if (rk == 0.0) { rk = 1e-5; status |= 2; }
                if (!(rk > 0.0)) status |= 4;
                Q1[i * ns + i] = ns * rk * rk;
            }
        } else {
            double Hm[MAXS * MAXS], T[MAXS * MAXS], Q0[MAXS * MAXS], S[MAXS * MAXS], T2[MAXS * MAXS];
            for (int i = 0; i < ns; ++i)
                for (int j = 0; j < ns; ++j) {
                    double s = pr->a[i * ns + j] + jac[i * D + j];
                    for (int c = 0; c < nu; ++c) s += (jac[i * D + ns + c] + pr->b[i * nu + c]) * pr->kfb[c * ns + j];
                    Hm[i * ns + j] = s;
                }
            for (int i = 0; i < ns; ++i)
                for (int j = 0; j < ns; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < ns; ++k) s += Q[i * ns + k] * Hm[j * ns + k];
                    T[i * ns + j] = s;
                }
            double trQ0 = 0.0;
            for (int i = 0; i < ns; ++i)
                for (int j = 0; j < ns; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < ns; ++k) s += Hm[i * ns + k] * T[k * ns + j];
                    Q0[i * ns + j] = s;
                    if (i == j) trQ0 += s;
                }
            /* r^2 = lambda_max(Q B) = lambda_max(Lb^T Q Lb) */
            for (int i = 0; i < ns; ++i)
                for (int j = 0; j < ns; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < ns; ++k) s += Q[i * ns + k] * Lb[k * ns + j];
                    T2[i * ns + j] = s;
                }
            for (int i = 0; i < ns; ++i)
                for (int j = 0; j < ns; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < ns; ++k) s += Lb[k * ns + i] * T2[k * ns + j];
                    S[i * ns + j] = s;
                }
            for (int i = 0; i < ns; ++i)
                for (int j = i + 1; j < ns; ++j) S[i * ns + j] = S[j * ns + i] = 0.5 * (S[i * ns + j] + S[j * ns + i]);
            const double r2 = sym_lambda_max(S, ns), r1 = sqrt(r2);
            double dsig[MAXS], dmu[MAXS], trS = 0.0, trM = 0.0;
            for (int i = 0; i < ns; ++i) {
                double bs = pr->beta * (sqrt(var[i]) + pr->l_sigma[i] * r1);
                if (bs != bs) status |= 1;
                if (bs == 0.0) { bs = 1e-5; status |= 2; }
                const double um = pr->l_mu[i] * r2;
                if (!(bs > 0.0) || !(um > 0.0)) status |= 4;
                dsig[i] = ns * bs * bs;
                dmu[i] = ns * um * um;
                trS += dsig[i];
                trM += dmu[i];
            }
            const double c1 = sqrt(trS / trM);
            double trSum = 0.0, dsum[MAXS];
            for (int i = 0; i < ns; ++i) {
                dsum[i] = (1.0 + 1.0 / c1) * dsig[i] + (1.0 + c1) * dmu[i];
                trSum += dsum[i];
            }
            const double c2 = sqrt(trSum / trQ0);
            for (int i = 0; i < ns; ++i) {
                for (int j = 0; j < ns; ++j) Q1[i * ns + j] = (1.0 + c2) * Q0[i * ns + j];
                Q1[i * ns + i] = (1.0 + 1.0 / c2) * dsum[i] + (1.0 + c2) * Q0[i * ns + i];
            }
        }
        have_q = 1;
        /* costs */
        if (pr->obj_mode == 0) {
            for (int i = 0; i < ns; ++i) obj -= var[i];
        } else {
            for (int i = 0; i < ns; ++i) obj += pr->w_abs[i] * fabs(pr->target[i] - p1[i]) + pr->w_lin[i] * p1[i];
        }
        int uviol = 0;
        for (int c = 0; c < nu; ++c) uviol |= (u[c] < pr->u_min[c]) || (u[c] > pr->u_max[c]);
        if (uviol) con += 3.0;
        if (pr->con_mode == 1 || t == H - 1) {
            int viol = 0;
            for (int r = 0; r < pr->m; ++r) {
                double hc = 0.0, hq = 0.0;
                for (int i = 0; i < ns; ++i) {
                    hc += pr->h_mat[r * ns + i] * p1[i];
                    double s = 0.0;
                    for (int j = 0; j < ns; ++j) s += Q1[i * ns + j] * pr->h_mat[r * ns + j];
                    hq += pr->h_mat[r * ns + i] * s;
                }
                viol |= (hc + sqrt(hq) - pr->h_vec[r] >= 0.0);
            }
            if (viol) con += 10.0;
        }
        if (traj_p) memcpy(traj_p + (size_t)t * ns, p1, sizeof(double) * ns);
        if (traj_q) memcpy(traj_q + (size_t)t * ns * ns, Q1, sizeof(double) * ns * ns);
        if (sigma) memcpy(sigma + (size_t)t * ns, var, sizeof(double) * ns);
        memcpy(p, p1, sizeof(double) * ns);
        memcpy(Q, Q1, sizeof(double) * ns * ns);
    }
    *obj_out = obj;
    *con_out = con;
    return status;
}

/* P particles from the same start state; actions [P x H x n_u]; outputs may be NULL except obj / con. */
int sxo_rollout(const sxo_problem *pr, int P, int H, const double *x0, const double *q0, const double *actions,
                double *traj_p, double *traj_q, double *sigma, double *obj, double *con) {
    if (pr->n_s > MAXS || pr->n_u > MAXS) return -1;
    int status = 0;
    const int ns = pr->n_s, nu = pr->n_u;
#pragma omp parallel reduction(| : status)
    {
        double *work = (double *)malloc(sizeof(double) * 2 * (size_t)pr->n);
#pragma omp for schedule(static)
        for (int i = 0; i < P; ++i)
            status |= rollout_one(pr, H, x0, q0, actions + (size_t)i * H * nu, traj_p ? traj_p + (size_t)i * H * ns : NULL,
                                  traj_q ? traj_q + (size_t)i * H * ns * ns : NULL, sigma ? sigma + (size_t)i * H * ns : NULL,
                                  obj + i, con + i, work);
        free(work);
    }
    return status;
}

int sxo_max_threads(void);
#ifdef _OPENMP
#include <omp.h>
int sxo_max_threads(void) { return omp_get_max_threads(); }
#else
int sxo_max_threads(void) { return 1; }
#endif
