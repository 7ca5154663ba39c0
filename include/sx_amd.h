/*
 * sx_amd.h -- C ABI of libsxamd.so, the MI355X (gfx950) implementation of the CEM safe-MPC hot path of
 * oscarkey/safe-exploration.
 *
 * The reference has no FFI: its boundary is two Python ABCs and one factory branch.  This header is
 * the boundary a maintainer binds with ctypes (INTEGRATION.md shows the stub).  Each entry point names the reference
 * interface it replaces; paths are relative to the reference root.
 *
 * Conventions
 *   - every pointer marked "dev" is a device pointer into HBM, row-major, contiguous, float64 unless noted;
 *     the library borrows it for the duration of the call's kernels and allocates nothing persistent;
 *   - structs are passed by pointer to HOST memory and copied into kernel arguments;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued on it, nothing is
 *     synchronised, the calls are re-entrant per stream;
 *   - return value: SX_OK or an SX_ERR_* code (bad shape, unsupported dimension, HIP launch error);
 *   - numerical trouble is reported through a device status word (SX_STATUS_* bits), read by the host once per solve
 *     and mapped onto the reference's ValueError (safe_exploration/gp_reachability_pytorch.py:76-80,117-121,149-153).
 */
#ifndef SX_AMD_H
#define SX_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SX_MAX_NS 4              /* state dimension: pendulum 2, cart-pole 4 */
#define SX_MAX_NU 2
#define SX_MAX_D (SX_MAX_NS + SX_MAX_NU)
#define SX_MAX_M 16              /* polytope rows: pendulum 4, cart-pole 9 */
#define SX_TILE 16               /* particles per workgroup = one f64 MFMA tile column block */

#define SX_OK 0
#define SX_ERR_ARG 1             /* null pointer / non-positive size / inconsistent shapes */
#define SX_ERR_UNSUPPORTED 2     /* (n_s, n_u) not instantiated, m > SX_MAX_M, n_train too large for the fused path */
#define SX_ERR_LAUNCH 3          /* HIP reported an error on launch */

#define SX_STATUS_NAN 1          /* _fix_zeros_nans saw a NaN: gp_reachability_pytorch.py:234-236 */
#define SX_STATUS_ZERO_FIX 2     /* an exact zero was replaced by 1e-5: gp_reachability_pytorch.py:238-241 */
#define SX_STATUS_UB_NONPOS 4    /* ellipsoid_from_rectangle got u_b <= 0: utils_ellipsoid.py:304 */

#define SX_OBJ_NEG_VARIANCE 0    /* -sum_d sigma_d: safempc_cem.py:308-311 */
#define SX_OBJ_AFFINE_ABS 1      /* sum_j w_abs_j |target_j - p_j| + w_lin_j p_j: environments.py:505-510, lunarlander.py:111-113 */
#define SX_CON_TERMINAL 0        /* EllipsoidTerminalConstraint: safempc_cem.py:102-113 */
#define SX_CON_ALL_STATES 1      /* EllipsoidStateConstraint on every prefix: safempc_cem.py:116-132 */

#define SX_ACTION_VIOLATION_COST 3.0  /* test_safempc_cem.py:59-71 */
#define SX_STATE_VIOLATION_COST 10.0  /* safempc_cem.py:132 */

/* Exact multi-output GP in the form the kernels consume.
 * Replaces: the gpytorch model behind GpCemSSM (ssm_cem/gp_ssm_cem.py:33-57, ssm_pytorch/gaussian_process.py:82-140).
 * Built by sx_gp_pack() from the inverse Cholesky factors W_d = chol(K_d + noise_d I)^-1 and alpha_d. */
typedef struct sx_gp_model {
    int32_t n_s, n_u;            /* outputs, action dims; D = n_s + n_u inputs */
    int32_t n_train;             /* N */
    int32_t n_pad;               /* 16 * ceil((N + 1 + D) / 16): W_d padded, with room for the mean/Jacobian rows */
    double inv_ls2[SX_MAX_NS * SX_MAX_D];  /* [n_s x D] 1 / lengthscale^2 (ARD, per output) */
    double outputscale[SX_MAX_NS];
    double noise[SX_MAX_NS];     /* likelihood noise, added to the predictive variance (gp_ssm_cem.py:93) */
    const double* x_train;       /* dev [N x D] */
    const double* a_pack;        /* dev, sx_gp_pack_sizes() doubles: W_d with the rows alpha_d, alpha_d * X_j / l_dj^2
                                    appended, in MFMA fragment order */
    const int32_t* stage_tab;    /* dev, sx_gp_pack_sizes() int32: the static MFMA operand stream of every wave */
} sx_gp_model;

/* Environment / solver constants of one MPC problem.
 * Replaces: the attributes CemSafeMPC reads from env/conf (safempc_cem.py:166-196) and the constraint objects
 * (safempc_cem.py:135-146). */
typedef struct sx_env {
    int32_t n_s, n_u;
    int32_t m;                   /* polytope rows */
    int32_t obj_mode;            /* SX_OBJ_* */
    int32_t con_mode;            /* SX_CON_* */
    int32_t reserved;
    double beta;                 /* c_safety */
    double a[SX_MAX_NS * SX_MAX_NS];      /* linear prior A (zeros if no prior: safempc_cem.py:291-296) */
    double b[SX_MAX_NS * SX_MAX_NU];
    double k_fb[SX_MAX_NU * SX_MAX_NS];   /* LQR feedback (safempc_simple.py:1105-1129) */
    double l_mu[SX_MAX_NS];
    double l_sigma[SX_MAX_NS];
    double h_mat[SX_MAX_M * SX_MAX_NS];   /* safe polytope h_mat x <= h_vec */
    double h_vec[SX_MAX_M];
    double u_min[SX_MAX_NU];
    double u_max[SX_MAX_NU];
    double obj_w_abs[SX_MAX_NS];
    double obj_target[SX_MAX_NS];
    double obj_w_lin[SX_MAX_NS];
} sx_env;

/* Library / build identification: returns "sxamd <version> gfx950". */
const char* sx_version(void);

/* ---- exact GPs with a degenerate kernel: 'linear' and 'nn' of ssm_cem/gp_ssm_cem.py:45-57,140-185 ----
 * k_d(x, x') = c_d phi(x) . phi(x'), phi = identity (linear kernel) or a small fully connected network with the reference's
 * per-point min/max normalisation (NNFeatureKernel).  Such a GP is Bayesian linear regression on F features: the kernels
 * below work in weight space (A_d = Phi^T Phi + noise_d / c_d I, F x F), one particle per lane, no N x N operand. */
#define SX_FEAT_MAX_WIDTH 32     /* widest layer / feature count F */
#define SX_FEAT_MAX_LAYERS 3     /* linear layers of the feature network (0 = linear kernel) */
typedef struct sx_feat_model {
    int32_t n_s, n_u;
    int32_t n_feat;              /* F: D for the linear kernel, the last layer's width otherwise */
    int32_t n_layers;            /* 0 = phi(z) = z */
    int32_t normalise;           /* 1 = phi = 2 (f - min f) / max f - 1 per point (gp_ssm_cem.py:176-181) */
    int32_t width[SX_FEAT_MAX_LAYERS + 1];   /* width[0] = D, width[l] = outputs of layer l */
    double prelu;                /* slope of the PReLU behind the last layer (ReLU between the layers) */
    double noise[SX_MAX_NS];     /* likelihood noise, included in the predictive variance */
    const double* net;           /* dev: per layer W_l [width[l] x width[l-1]] row-major, then b_l [width[l]] */
    const double* wbar;          /* dev [n_s x F]      posterior weight means (sx_feat_fit) */
    const double* minv;          /* dev [n_s x F x F]  M_d = chol(A_d)^-1, lower triangular (sx_feat_fit) */
} sx_feat_model;

/* phi dev [N x F] = phi(x dev [N x D]).  Uses model->{n_s,n_u,n_feat,n_layers,normalise,width,prelu,net}. */
int sx_feat_features(const sx_feat_model* model, const double* x, int N, double* phi, void* stream);
/* Weight-space fit from phi dev [N x F], y dev [N x n_s], lambda host [n_s] = noise_d / c_d:
 * wbar dev [n_s x F], minv dev [n_s x F x F], stats dev [n_s x 3] = { y^T y, |M Phi^T y|^2, sum log diag chol(A) } (what the
 * exact marginal likelihood needs), status dev int32 (SX_STATUS_NOT_PD).
 * Replaces: gpytorch's ExactGP on set_train_data for these kernels (ssm_cem/gp_ssm_cem.py:96-101). */
int sx_feat_fit(const sx_feat_model* model, const double* phi, const double* y, int N, const double* lambda, double* wbar,
                double* minv, double* stats, int32_t* status, void* stream);
/* Posterior at z dev [P x D]: same outputs as sx_gp_predict.  Replaces GpCemSSM.predict_* for these kernels. */
int sx_feat_predict(const sx_feat_model* model, const double* z, int P, double* mean, double* var, double* jac, void* stream);
/* The CEM particle rollout over such a GP: same arguments and outputs as sx_cem_rollout (no workspace). */
int sx_cem_rollout_feat(const sx_feat_model* model, const sx_env* env, int E, int P, int H, const double* x0, const double* q0,
                        const double* mean, const double* std, const double* noise, double* actions, double* traj,
                        double* sigma, double* obj_cost, double* con_cost, int32_t* status, void* stream);

/* ---- MC-dropout state-space models: ssm_cem/dropout_ssm_cem.py, gal_concrete_dropout.py ----
 * An ensemble of S thinned ReLU networks: dropout masks drawn once per (re)training and held fixed, prediction = mean and
 * unbiased variance over the members, mean Jacobian by reverse sweeps.  One or two hidden layers of <= 64 units run on the
 * f64 matrix cores (csrc/sx_mlp_mfma.hpp: a 16-particle tile per workgroup, a member per wave, activations in registers);
 * other shapes one particle per lane (csrc/sx_mlp.hpp).  SX_MLP_PATH=valu in the environment forces the latter (A/B runs). */
#define SX_MLP_MAX_HIDDEN 4      /* hidden layers */
#define SX_MLP_MAX_WIDTH 64      /* hidden units per layer (the reference's default network is 64 x 64) */
typedef struct sx_mlp_model {
    int32_t n_s, n_u;
    int32_t n_hidden;            /* L */
    int32_t n_out;               /* rows of the output layer (>= n_s; the first n_s are the predicted means) */
    int32_t n_samples;           /* S ensemble members (mc_dropout_num_samples) */
    int32_t predict_std;         /* 1: outputs n_s .. 2 n_s - 1 are log standard deviations; the variance gains the members'
                                    mean exp(2 log std): the expectation of dropout_ssm_cem.py:106-109 over its fresh noise */
    int32_t width[SX_MLP_MAX_HIDDEN + 1];   /* width[0] = D, width[l] = hidden layer l */
    const double* net;           /* dev: W_1 [w1 x D] row-major, b_1, ..., W_L, b_L, W_out [n_out x w_L], b_out */
    const double* masks;         /* dev [S x (width[0] + ... + width[L])]: the multipliers of the input and of every hidden
                                    layer's activations (0 or 1 / keep for Bernoulli dropout, relaxed values for concrete) */
} sx_mlp_model;
/* Posterior at z dev [P x D]: same outputs as sx_gp_predict.  Replaces McDropoutSSM / GalConcreteDropoutSSM.predict_*
 * (dropout_ssm_cem.py:79-112, gal_concrete_dropout.py:164-196). */
int sx_mlp_predict(const sx_mlp_model* model, const double* z, int P, double* mean, double* var, double* jac, void* stream);
/* The CEM particle rollout over the ensemble: same arguments and outputs as sx_cem_rollout (no workspace). */
int sx_cem_rollout_mlp(const sx_mlp_model* model, const sx_env* env, int E, int P, int H, const double* x0, const double* q0,
                       const double* mean, const double* std, const double* noise, double* actions, double* traj,
                       double* sigma, double* obj_cost, double* con_cost, int32_t* status, void* stream);

/* Optional kernel timer -- measurement support, not part of the reference's surface (it has no profiler: SURVEY.md 5).
 * While enabled, the launches of the path's kernels (every `sx_profile_stride`-th of each kind) carry a start and a stop HIP
 * event on the stream the kernel is launched on (hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps, the
 * interval rocprofv3's kernel trace reports; SX_PROF_RECORD=1 in the environment: two hipEventRecord around the launch, as
 * in rounds 1-2, which read ~2.8 us more).  At most `max_launches` launches are recorded; sx_profile_collect synchronises
 * those events and returns the summed elapsed time and the launch count of one kernel class.  bench.py's
 * `roofline.avg_launch_us` comes from here. */
#define SX_PROF_ROLLOUT_FUSED 0  /* cem_rollout_kernel            */
#define SX_PROF_RANK 1           /* cem_rank_kernel               */
#define SX_PROF_KSTAR_BIG 2      /* kstar_big_kernel   (large-N path) */
#define SX_PROF_TRMM_BIG 3       /* trmm_reduce_kernel (large-N path) */
#define SX_PROF_STEP_BIG 4       /* step_big_kernel    (large-N path) */
#define SX_PROF_ROLLOUT_FEAT 5   /* cem_rollout_feat_kernel (degenerate-kernel GPs) */
#define SX_PROF_ROLLOUT_MLP 6    /* cem_rollout_mlp_mfma_kernel / cem_rollout_mlp_kernel (MC-dropout ensembles) */
#define SX_PROF_KINDS 7
int sx_profile_enable(int max_launches);
int sx_profile_stride(int every);   /* time every n-th launch of a kernel class only (default 1): a timed launch costs the
                                       launch path a few microseconds, which a 130 us kernel notices */
int sx_profile_stride_kind(int kind, int every);   /* the same for one kernel class */
int sx_profile_collect(int kind, double* total_ms, int64_t* launches);
int sx_profile_disable(void);

#ifndef SX_WAVES
#define SX_WAVES 8               /* waves per workgroup in the GP kernels (the stage table is laid out for it) */
#endif

/* Sizes of sx_gp_model.a_pack (doubles) and sx_gp_model.stage_tab (int32). */
int sx_gp_pack_sizes(int n_s, int n_u, int n_train, int64_t* a_doubles, int64_t* tab_ints);

#define SX_STATUS_NOT_PD 8       /* sx_gp_fit: K + noise I is not positive definite (gpytorch would raise too) */

/* Exact-GP fit for fixed hyper-parameters: K_d + noise_d I = L_d L_d^T, linv = L_d^-1 (dev [n_s x N x N]),
 * alpha_d = (K_d + noise_d I)^-1 y_d (dev [n_s x N]), logdet_d = sum log diag L_d (dev [n_s]).
 * model->{n_s,n_u,n_train,inv_ls2,outputscale,noise,x_train} must be set; y_train dev [N x n_s];
 * work dev [n_s x N x N] scratch (holds L on return); status dev int32 (SX_STATUS_NOT_PD).  N <= 4096.
 * Replaces: what gpytorch's ExactGP computes when GpCemSSM sets new training data
 * (ssm_cem/gp_ssm_cem.py:96-101, ssm_pytorch/gaussian_process.py:82-140). */
int sx_gp_fit(const sx_gp_model* model, const double* y_train, double* work, double* linv, double* alpha,
              double* logdet, int32_t* status, void* stream);

/* Exact marginal log likelihood per output and its gradient w.r.t. the hyper-parameters, from sx_gp_fit's outputs:
 * mll dev [n_s]; grad dev [n_s x (D + 2)] = d mll_d / d (lengthscale_d[0..D), outputscale_d, noise_d).
 * work dev [n_s x N x N]: scratch, overwritten (sx_gp_fit's `work` may be passed: L is not needed any more).
 * Replaces: the autograd pass of GpCemSSM._train_model (ssm_cem/gp_ssm_cem.py:103-129,
 * gpytorch.ExactMarginalLogLikelihood); the Adam update itself stays on the host. */
int sx_gp_mll_grad(const sx_gp_model* model, const double* y_train, const double* linv, const double* alpha,
                   const double* logdet, double* work, double* mll, double* grad, void* stream);

/* Lays W_d = L_d^-1 (dev [n_s x N x N], lower triangular) and alpha (dev [n_s x N]) out in fragment order.
 * model->{n_s,n_u,n_train,inv_ls2,x_train,a_pack,stage_tab} must be set; n_pad is filled in.
 * Replaces: GpCemSSM._update_model (ssm_cem/gp_ssm_cem.py:96-101) -- where the prediction operands are (re)built. */
int sx_gp_pack(sx_gp_model* model, const double* linv, const double* alpha, void* stream);

/* Posterior at z dev [P x D]: mean dev [P x n_s], var dev [P x n_s] (noise included), jac dev [P x n_s x D] or NULL.
 * Replaces: GpCemSSM.predict_with_jacobians / predict_without_jacobians / _predict (ssm_cem/gp_ssm_cem.py:59-94)
 * and compute_jacobian_fast (ssm_pytorch/utilities.py:54-85). */
int sx_gp_predict(const sx_gp_model* model, const double* z, int P, double* mean, double* var, double* jac,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* d var / d z at z dev [P x D]: jac_var dev [P x n_s x D]; `linv` dev [n_s x N x N] as produced by sx_gp_fit for this model.
 * Not on the CEM path: it completes the numpy adapter (StateSpaceModel.predict(..., jacobians=True) returns it as its
 * fourth output).  Replaces: compute_jacobian(pred_var, inp) in GPyTorchSSM._predict
 * (ssm_pytorch/gaussian_process.py:222-231). */
int sx_gp_predict_var_jac(const sx_gp_model* model, const double* linv, const double* z, int P, double* jac_var,
                          void* stream);

/* d^2 mean / dz dz^T at z dev [P x D]: hess dev [P x n_s x D x D] (symmetric); `alpha` dev [n_s x N] as produced by sx_gp_fit.
 * Not on the CEM path: it completes the numpy adapter's linearize_predict / get_linearize_reverse, which the casadi
 * callback of the reference's other solvers consumes (state_space_models.py:279-304).  Replaces:
 * GPyTorchSSM._compute_hessian_mean (ssm_pytorch/gaussian_process.py:160-187, the `hessian` package over autograd). */
int sx_gp_predict_mean_hessian(const sx_gp_model* model, const double* alpha, const double* z, int P, double* hess,
                               void* stream);

/* Bytes of workspace sx_gp_predict needs (0 while the training set fits the single-launch kernel; < 0 = bad arguments). */
int64_t sx_gp_predict_workspace_bytes(const sx_gp_model* model, int P);

/* One-step ellipsoidal reachability given the GP outputs at (p, u).
 * p dev [P x n_s]; Q dev [P x n_s x n_s] or NULL (point branch); u dev [P x n_u]; mean/var dev [P x n_s];
 * jac dev [P x n_s x D] (ignored in the point branch); outputs p1 dev [P x n_s], Q1 dev [P x n_s x n_s],
 * sigma dev [P x n_s] (the variance after the zero fix-up); status dev int32 (OR-ed; bit 16 is used as scratch during
 * the call and is clear on return).
 * The zero fix-up follows the reference's WHOLE-BATCH rule: an exact zero anywhere in `var` lifts every var <= 0 of the
 * batch to 1e-5 (a one-workgroup pre-pass over `var` finds it); without one a negative variance ends as SX_STATUS_NAN.
 * Uses env->{a,b,k_fb,l_mu,l_sigma,beta}.
 * Replaces: onestep_reachability (gp_reachability_pytorch.py:18-181) with its helpers
 * compute_remainder_overapproximations_pytorch (utils.py:152-194), ellipsoid_from_rectangle_pytorch and
 * sum_two_ellipsoids_pytorch (utils_ellipsoid.py:102-140,282-309), _fix_zeros_nans (:234-243). */
int sx_onestep_reach(const sx_env* env, int P, const double* p, const double* Q, const double* u, const double* mean,
                     const double* var, const double* jac, double* p1, double* Q1, double* sigma, int32_t* status,
                     void* stream);

/* d dev [P x m] = h_mat p + c_safety sqrt(diag(h_mat Q h_mat^T)) - h_vec; inside dev uint8 [P] or NULL.
 * Replaces: lin_ellipsoid_safety_distance / is_ellipsoid_inside_polytope (gp_reachability_pytorch.py:184-231). */
int sx_polytope_distance(const sx_env* env, int P, const double* p, const double* Q, double c_safety, double* d,
                         uint8_t* inside, void* stream);

/* The fused CEM particle rollout: E independent problems x P particles x H steps in ONE launch.
 *   x0      dev [E x n_s]              start states (points; the reference starts every solve from a point,
 *                                      safempc_cem.py:234-235)
 *   q0      dev [E x n_s x n_s] | NULL start shape matrices (NULL = point)
 *   mean,std dev [E x H x n_u]         sampling distribution (ignored when noise == NULL)
 *   noise   dev [E x P x H x n_u]|NULL standard-normal draws; NULL = `actions` is an INPUT
 *   actions dev [E x P x H x n_u]      out: mean + std * noise  (or in, see above)
 *   traj    dev [E x P x H x (n_s + n_s^2)] | NULL   flat states [p | vec_rowmajor(Q)] (PQFlattener, safempc_cem.py:30-76)
 *   sigma   dev [E x P x H x n_s] | NULL
 *   obj_cost, con_cost dev [E x P]     summed objective / constraint cost
 *   status  dev int32                  OR of SX_STATUS_*
 *   workspace dev, sx_cem_rollout_workspace_bytes() bytes (may be NULL when that is 0): training sets whose Kstar tile does
 *                                      not fit in LDS (config 4: N = 2000) take the three-launch-per-step path and keep
 *                                      Kstar, partial sums and particle state there
 * Replaces: the H sequential DynamicsFunc callbacks + per-trajectory Constraint calls the optimiser makes per
 * iteration (safempc_cem.py:102-156,288-312; call sites of the absent constrained-cem-mpc). */
int sx_cem_rollout(const sx_gp_model* model, const sx_env* env, int E, int P, int H, const double* x0, const double* q0,
                   const double* mean, const double* std, const double* noise, double* actions, double* traj,
                   double* sigma, double* obj_cost, double* con_cost, int32_t* status, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* sx_cem_rollout for every CEM iteration after the first: the sampling distribution is not handed in but REFIT from the
 * previous iteration's elite rows inside the kernel's prologue -- mean and unbiased standard deviation (0 when k == 1) over
 *   elite_rows dev [E x k x (2 + H*n_u)]  the [con, obj, actions...] rows sx_cem_rank_refit writes (any order),
 * by every workgroup for itself (a wave per column, fixed summation order: the same numbers in every workgroup and on
 * every GPU), while the kernel's other start-up loads travel.  actions = mean + std * noise as in sx_cem_rollout.
 *   mean_out, std_out dev [E x H*n_u] | both NULL   the refit, for callers that want to see it
 * Only for models on the single-launch path (sx_cem_rollout_workspace_bytes() == 0) with 2 H n_u <= 256 (1 + n_s);
 * SX_ERR_UNSUPPORTED otherwise: use sx_cem_rank_refit's mean / std and sx_cem_rollout there.
 * Replaces: the refit step of ConstrainedCemMpc.get_actions (as sx_cem_rank_refit's mean / std outputs do), moved off the
 * ranking kernel's serial tail. */
int sx_cem_rollout_elites(const sx_gp_model* model, const sx_env* env, int E, int P, int H, const double* x0, const double* q0,
                          const double* elite_rows, int k, const double* noise, double* actions, double* traj, double* sigma,
                          double* obj_cost, double* con_cost, int32_t* status, double* mean_out, double* std_out, void* stream);

/* Bytes of workspace sx_cem_rollout needs for this model and problem size: 0 = the fused single-launch path applies;
 * < 0 = bad arguments. */
int64_t sx_cem_rollout_workspace_bytes(const sx_gp_model* model, int E, int P, int H);

/* Which kernel sx_cem_rollout / sx_cem_rollout_elites launch for this model and horizon (no launch, no device access):
 *   SX_FORM_RH      cem_rollout_rh_kernel: 8 waves, the GP's triangular factors partly resident (registers / LDS / L2)
 *   SX_FORM_RW      cem_rollout_rw_kernel: 4 waves, the factors in the register file
 *   SX_FORM_STREAM  cem_rollout_kernel: the factors streamed from L2, Kstar of all outputs in LDS
 *   SX_FORM_BYOUT   cem_rollout_kernel, one output's Kstar in LDS at a time
 *   SX_FORM_BIG     the three-launch-per-step path (Kstar in HBM)
 * or < 0 for bad arguments.  Reporting only (bench.py names the kernel it timed); the choice itself is the library's.
 * No counterpart in the reference. */
#define SX_FORM_STREAM 0
#define SX_FORM_RW 1
#define SX_FORM_RH 2
#define SX_FORM_BYOUT 3
#define SX_FORM_BIG 4
int sx_cem_rollout_form(const sx_gp_model* model, int H);

/* The ONE device -> host hand-off of a solve, packed by one launch: out dev double [G + E + 1 + E*row_len] =
 *   [status words of the G ranks | best_ok[E] | 1.0 if any of the `q_count` doubles at `q_block` is non-zero | best [E x row_len]]
 * (q_block may be NULL: the flag is 0).  The caller copies `out` to the host once and reads everything from it.
 * Replaces: the synchronisations of CemSafeMPC.get_action (safempc_cem.py:231-263): PQFlattener's `q.nonzero()` scan
 * (:69-71), the optimiser's return value and the failure check of gp_reachability_pytorch.py:149-153, each a trip of its own
 * in the reference. */
int sx_cem_pack_result(int G, int E, int row_len, const int32_t* status, const int32_t* best_ok, const double* q_block,
                       int64_t q_count, const double* best, double* out, void* stream);

/* 1 if sx_cem_rank_refit ranks E problems of P candidates by counting over the whole chip (elite rows in rank order), 0 if
 * by one workgroup per problem (best first, then index order).  Depends on (E, P) only.  A caller that moves the refit into
 * the next rollout (sx_cem_rollout_elites) does so where this returns 1: with many problems at once the one-workgroup
 * kernels refit side by side and the rollout's prologue has nothing to gain. */
int sx_cem_rank_counts(int E, int P);

/* Ranking + elite refit for E problems.  One or two problems of up to 8192 candidates are ranked by counting, spread over
 * the whole chip (csrc/sx_rank_count.hpp: needs elite_rows whenever mean is wanted); anything else by one workgroup per
 * problem (csrc/sx_rank.hpp).  The choice depends on (E, P) only.
 * Candidates c = 0..P-1 of problem e have con = con_cost[(e*P+c)*cost_stride], obj likewise, and an action row of
 * `row_len` doubles at actions + (e*P+c)*act_stride.  Order: lexicographic (con, obj, c); NaN sorts last.
 *   elite_idx  dev int32 [E x k]            elite indices: the best first, the others in a deterministic but unspecified order (may be NULL)
 *   elite_rows dev [E x k x (2 + row_len)]  [con, obj, actions...] of the elites, same order (may be NULL) -- the buffer
 *                                           that is all-reduced across GPUs
 *   mean, std  dev [E x row_len]            refit (unbiased std; 0 when k == 1)  (may be NULL: no refit)
 *   best       dev [E x row_len]            first-ranked action sequence (may be NULL)
 *   best_ok    dev int32 [E]                1 if the first-ranked candidate has con == 0 (may be NULL)
 * Replaces: elite selection / refit / "best feasible or None" of ConstrainedCemMpc.get_actions (safempc_cem.py:235,
 * test_safempc_cem.py:83-148). */
int sx_cem_rank_refit(int E, int P, int k, int row_len, const double* con_cost, const double* obj_cost,
                      int64_t cost_stride, const double* actions, int64_t act_stride, int32_t* elite_idx,
                      double* elite_rows, double* mean, double* std, double* best, int32_t* best_ok, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SX_AMD_H */
