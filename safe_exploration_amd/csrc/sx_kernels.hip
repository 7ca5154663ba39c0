// libsxamd: launchers + C ABI (include/sx_amd.h) over the kernels in sx_*.hpp.  gfx950 only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/sx_amd.h"
#include "sx_gp.hpp"
#include "sx_reach.hpp"
#include "sx_big.hpp"
#include "sx_fit.hpp"
#include "sx_fit_blocked.hpp"
#include "sx_rollout.hpp"
#include "sx_launch.hpp"
#include "sx_rw_launch.hpp"
#include "sx_rank.hpp"
#include "sx_rank_count.hpp"
#include "sx_feat.hpp"
#include "sx_mlp.hpp"
#include "sx_mlp_mfma.hpp"

namespace sx {

#ifdef SX_STAMPS
static unsigned long long* g_stamp_host = nullptr;   // diagnostic build: the phase-stamp buffer (sx_debug_set_stamps)
#endif

constexpr int kPredictThreads = 64 * SX_WAVES;

// ---------------------------------------------------------------------------------------------------------------
// sx_gp_predict: one 16-point tile per workgroup
// ---------------------------------------------------------------------------------------------------------------
// (BYOUT: one output's Kstar in LDS at a time, as in the rollout kernel)
template <int NS, int NU, bool BYOUT = false>
__global__ __launch_bounds__(kPredictThreads) void gp_predict_kernel(GpConst<NS, NS + NU> gc,
                                                                     const int4* __restrict__ stage_tab,
                                                                     const double* __restrict__ z, int P,
                                                                     double* __restrict__ mean, double* __restrict__ var,
                                                                     double* __restrict__ jac) {
    constexpr int D = NS + NU;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    GpTileLds<NS, D> lds;
    const int nw = blockDim.x >> 6;
    lds.carve(smem, gc.n_train, gc.n_pad, nw, BYOUT ? 1 : NS);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    gp_load_xs(gc, lds);
    const MfmaHead head = gp_mfma_head(gc, stage_tab, wave, nw, lane, gc.stage_cap);
    const int4* __restrict__ const tab_one = stage_tab + (size_t)nw * (1 + gc.stage_cap);
    for (int tile = blockIdx.x; tile * SX_TILE < P; tile += gridDim.x) {
        const int g0 = tile * SX_TILE;
        if (tid < SX_TILE * D) {
            const int c = tid / D, j = tid - c * D;
            lds.zs[tid] = (g0 + c < P) ? z[(int64_t)(g0 + c) * D + j] : 0.0;
        }
        __syncthreads();
        int qb, qe;
        kstar_pair_range(gc.n_pad >> 3, wave, 1, nw, qb, qe);
        double zq[D];
#pragma unroll
        for (int j = 0; j < D; ++j) zq[j] = lds.zs[(lane & 15) * D + j];
        if constexpr (BYOUT) {
            auto one_output = [&](auto dtag) {
                constexpr int DD = decltype(dtag)::value;
                if constexpr (DD < NS) {
                    const int4* __restrict__ tab_d = tab_one + (size_t)DD * nw * (1 + gc.stage_cap_one);
                    const MfmaHead head_d = gp_mfma_head(gc, tab_d, wave, nw, lane, gc.stage_cap_one);
                    gp_kstar_phase_one<NS, D, DD>(gc, lds, qb, qe, zq);
                    __syncthreads();
                    gp_mfma_phase<NS, D, 1>(gc, tab_d, lds, wave, nw, lane, head_d, gc.stage_cap_one, DD);
                    __syncthreads();
                }
            };
            one_output(std::integral_constant<int, 0>{});
            one_output(std::integral_constant<int, 1>{});
            one_output(std::integral_constant<int, 2>{});
            one_output(std::integral_constant<int, 3>{});
        } else {
            gp_kstar_phase(gc, lds, qb, qe, zq);
            __syncthreads();
            gp_mfma_phase(gc, stage_tab, lds, wave, nw, lane, head, gc.stage_cap);
            __syncthreads();
        }
        if (tid < SX_TILE && g0 + tid < P) {
            double zz[D], m[NS], v[NS], jc[NS][D];
#pragma unroll
            for (int j = 0; j < D; ++j) zz[j] = lds.zs[tid * D + j];
            if (jac) {
                gp_collect<NS, D, true>(gc, lds, nw, tid, zz, m, v, jc);
            } else {
                gp_collect<NS, D, false>(gc, lds, nw, tid, zz, m, v, jc);
            }
            const int64_t g = g0 + tid;
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                mean[g * NS + d] = m[d];
                var[g * NS + d] = v[d];
                if (jac) {
#pragma unroll
                    for (int j = 0; j < D; ++j) jac[(g * NS + d) * D + j] = jc[d][j];
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sx_onestep_reach / sx_polytope_distance: one particle per lane
// ---------------------------------------------------------------------------------------------------------------
// Pre-pass of sx_onestep_reach: does the variance batch hold an exact zero (gp_reachability_pytorch.py:238)?  ONE
// workgroup, so it is the only writer of the scratch bit: it clears the bit a previous call may have left and sets it
// again if this batch has a zero.  The main kernel reads the bit; both run on the caller's stream, in order.
constexpr int kStatusScratchBatchZero = 0x10000;
__global__ __launch_bounds__(1024) void batch_zero_flag_kernel(const double* __restrict__ var, int64_t n, int* __restrict__ status) {
    int any = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) any |= (var[i] == 0.0) ? 1 : 0;
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) {
        atomicAnd(status, ~kStatusScratchBatchZero);
        if (any) atomicOr(status, kStatusScratchBatchZero);
    }
}

template <int NS, int NU>
__global__ void onestep_reach_kernel(ReachConst<NS, NU> rc, int P, const double* __restrict__ p_in,
                                     const double* __restrict__ q_in, const double* __restrict__ u_in,
                                     const double* __restrict__ mean_in, const double* __restrict__ var_in,
                                     const double* __restrict__ jac_in, double* __restrict__ p_out,
                                     double* __restrict__ q_out, double* __restrict__ sig_out, int* __restrict__ status) {
    constexpr int D = NS + NU;
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= P) return;
    double p[NS], u[NU], mean[NS], var[NS], p1[NS], Q1[NS][NS];
    int st = 0;
    // the whole-batch rule of _fix_zeros_nans: with an exact zero anywhere in the batch, every var <= 0 is lifted
    const bool batch_zero = (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStatusScratchBatchZero) != 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        p[i] = p_in[g * NS + i];
        mean[i] = mean_in[g * NS + i];
        var[i] = var_in[g * NS + i];
    }
#pragma unroll
    for (int c = 0; c < NU; ++c) u[c] = u_in[g * NU + c];
    if (q_in == nullptr) {
        reach_point<NS, NU>(rc, p, u, mean, var, p1, Q1, st, batch_zero);
    } else {
        double Q[NS][NS], jac[NS][D];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
#pragma unroll
            for (int j = 0; j < NS; ++j) Q[i][j] = q_in[(g * NS + i) * NS + j];
#pragma unroll
            for (int j = 0; j < D; ++j) jac[i][j] = jac_in[(g * NS + i) * D + j];
        }
        reach_ellipsoid<NS, NU>(rc, p, Q, u, mean, var, jac, p1, Q1, st, batch_zero);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        p_out[g * NS + i] = p1[i];
        sig_out[g * NS + i] = var[i];
#pragma unroll
        for (int j = 0; j < NS; ++j) q_out[(g * NS + i) * NS + j] = Q1[i][j];
    }
    if (st) atomicOr(status, st);
}

template <int NS>
struct PolyArgs {
    double h_mat[SX_MAX_M * NS];
    double h_vec[SX_MAX_M];
    int m;
};

template <int NS>
__global__ void polytope_kernel(PolyArgs<NS> pa, int P, double c_safety, const double* __restrict__ p_in,
                                const double* __restrict__ q_in, double* __restrict__ d_out,
                                uint8_t* __restrict__ inside) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= P) return;
    double p[NS], Q[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        p[i] = p_in[g * NS + i];
#pragma unroll
        for (int j = 0; j < NS; ++j) Q[i][j] = q_in[(g * NS + i) * NS + j];
    }
    double d[SX_MAX_M];
    const bool viol = polytope_violated<SX_MAX_M, NS>(pa.h_mat, pa.h_vec, pa.m, c_safety, p, Q, d);
    for (int r = 0; r < pa.m; ++r) d_out[g * pa.m + r] = d[r];
    if (inside) inside[g] = viol ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------------------------
template <int NS, int NU>
static GpConst<NS, NS + NU> make_gp_const(const sx_gp_model* m, int nw) {
    constexpr int D = NS + NU;
    GpConst<NS, D> gc;
    for (int d = 0; d < NS; ++d) {
        for (int j = 0; j < D; ++j) {
            gc.inv_ls2[d * D + j] = m->inv_ls2[d * D + j];
            gc.nh_ils2[d * D + j] = -0.5 * m->inv_ls2[d * D + j];
            gc.k_nh_ils2[d * D + j] = gc.nh_ils2[d * D + j] * sx::kExpScale;
        }
        gc.log_os[d] = std::log(m->outputscale[d]);
        gc.k_log_os[d] = gc.log_os[d] * sx::kExpScale;
        gc.outputscale[d] = m->outputscale[d];
        gc.noise[d] = m->noise[d];
    }
    gc.x_train = m->x_train;
    gc.a_pack = m->a_pack;
    gc.stage_tab = reinterpret_cast<const int4*>(m->stage_tab);
    gc.n_train = m->n_train;
    gc.n_pad = m->n_pad;
    gc.stage_cap = gp_stage_cap(NS, m->n_pad, nw);
    gc.stage_cap_one = gp_stage_cap(1, m->n_pad, nw);
    return gc;
}

template <int NS, int NU>
static bool make_reach_const(const sx_env* env, ReachConst<NS, NU>& rc) {
    for (int i = 0; i < NS * NS; ++i) rc.a[i] = env->a[i];
    for (int i = 0; i < NS * NU; ++i) rc.b[i] = env->b[i];
    for (int i = 0; i < NU * NS; ++i) rc.kfb[i] = env->k_fb[i];
    for (int i = 0; i < NS; ++i) {
        rc.l_mu[i] = env->l_mu[i];
        rc.l_sigma[i] = env->l_sigma[i];
    }
    rc.beta = env->beta;
    // B = I + kfb^T kfb is SPD; lower Cholesky on the host
    double B[NS][NS];
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int c = 0; c < NU; ++c) s += env->k_fb[c * NS + i] * env->k_fb[c * NS + j];
            B[i][j] = s;
        }
    for (int i = 0; i < NS * NS; ++i) rc.cholB[i] = 0.0;
    for (int j = 0; j < NS; ++j) {
        double s = B[j][j];
        for (int k = 0; k < j; ++k) s -= rc.cholB[j * NS + k] * rc.cholB[j * NS + k];
        if (!(s > 0.0)) return false;
        const double ljj = std::sqrt(s);
        rc.cholB[j * NS + j] = ljj;
        for (int i = j + 1; i < NS; ++i) {
            double t = B[i][j];
            for (int k = 0; k < j; ++k) t -= rc.cholB[i * NS + k] * rc.cholB[j * NS + k];
            rc.cholB[i * NS + j] = t / ljj;
        }
    }
    return true;
}

template <int NS, int NU>
static void make_cost_const(const sx_env* env, CostConst<SX_MAX_M, NS, NU>& cc) {
    std::memset(&cc, 0, sizeof(cc));
    for (int r = 0; r < env->m; ++r) {
        for (int i = 0; i < NS; ++i) cc.h_mat[r * NS + i] = env->h_mat[r * NS + i];
        cc.h_vec[r] = env->h_vec[r];
    }
    for (int c = 0; c < NU; ++c) {
        cc.u_min[c] = env->u_min[c];
        cc.u_max[c] = env->u_max[c];
    }
    for (int i = 0; i < NS; ++i) {
        cc.w_abs[i] = env->obj_w_abs[i];
        cc.target[i] = env->obj_target[i];
        cc.w_lin[i] = env->obj_w_lin[i];
    }
    cc.m = env->m;
    cc.obj_mode = env->obj_mode;
    cc.con_mode = env->con_mode;
}

// ---------------------------------------------------------------------------------------------------------------
// Optional kernel timer (sx_profile_*): while enabled, every n-th launch of each of the path's kernel classes carries a
// pair of HIP events on the stream the kernel is launched on (sx_launch.hpp); sx_profile_collect adds the elapsed times up
// per kernel class.
// This is how bench.py measures `roofline.avg_launch_us` live, inside its timed region.
// ---------------------------------------------------------------------------------------------------------------
struct ProfEntry {
    int kind;
    hipEvent_t start, stop;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static size_t g_prof_cap = 0;
static int g_prof_stride[SX_PROF_KINDS] = {1, 1, 1, 1, 1, 1, 1};   // every n-th launch of a kernel class is timed
static long g_prof_seen[SX_PROF_KINDS] = {0};
static std::vector<ProfEntry> g_prof_entries;
static std::vector<hipEvent_t> g_prof_pool;

// Takes a (start, stop) event pair for one launch of kernel class `kind`, or returns false (timer off / cap reached).
bool prof_take(int kind, hipEvent_t* start, hipEvent_t* stop) {
    if (!g_prof_on) return false;   // (read without the lock: enabling mid-launch only loses that launch)
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_prof_on || g_prof_entries.size() >= g_prof_cap) return false;
    if ((g_prof_seen[kind]++ % g_prof_stride[kind]) != 0) return false;
    auto take = [&]() {
        hipEvent_t e = nullptr;
        if (!g_prof_pool.empty()) {
            e = g_prof_pool.back();
            g_prof_pool.pop_back();
        } else if (hipEventCreate(&e) != hipSuccess) {
            e = nullptr;
        }
        return e;
    };
    hipEvent_t a = take(), b = take();
    if (!a || !b) return false;
    g_prof_entries.push_back(ProfEntry{kind, a, b});
    *start = a;
    *stop = b;
    return true;
}

// (launch<>() -- every kernel of the path is launched through it -- and allow_lds<>() live in sx_launch.hpp, shared with the
// translation units of the register-resident rollout kernels.)
int check_launch() {
    // SX_DEBUG_SYNC=1: wait for the launch and report an asynchronous failure at the call that caused it (diagnosis only)
    static const bool debug_sync = std::getenv("SX_DEBUG_SYNC") != nullptr;
    if (debug_sync) {
        const hipError_t serr = hipDeviceSynchronize();
        if (serr != hipSuccess) {
            std::fprintf(stderr, "libsxamd: kernel failed: %s\n", hipGetErrorString(serr));
            return SX_ERR_LAUNCH;
        }
    }
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        std::fprintf(stderr, "libsxamd: HIP launch error: %s\n", hipGetErrorString(err));
        return SX_ERR_LAUNCH;
    }
    return SX_OK;
}


// Kernels that need more than 64 KB of dynamic LDS must be granted it once per (device, kernel); the grant is remembered,
// so the hot loop's launches make no runtime call besides the launch itself.
int allow_lds_ptr(const void* kernel, size_t bytes) {
    if (bytes > kMaxLdsBytes) return SX_ERR_UNSUPPORTED;
    if (bytes > 64 * 1024) {
        static std::mutex mu;
        static std::map<std::pair<int, const void*>, size_t> granted;
        int dev = 0;
        (void)hipGetDevice(&dev);
        const auto key = std::make_pair(dev, kernel);
        std::lock_guard<std::mutex> lock(mu);
        auto it = granted.find(key);
        if (it != granted.end() && it->second >= bytes) return SX_OK;
        if (hipFuncSetAttribute(key.second, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            return SX_ERR_UNSUPPORTED;
        }
        granted[key] = bytes;
    }
    return SX_OK;
}

// compute units of the current device (the persistent grids are sized by it)
int device_cus() {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
    return n;
}

static bool predict_fits(int ns, int nu, int n_train, int n_pad, int ns_lds = -1) {
    const int nw = kPredictThreads / 64;
    return gp_tile_lds_doubles(ns, ns + nu, n_train, n_pad, nw, ns_lds) * sizeof(double) <= kMaxLdsBytes && n_pad <= 1024;
}

template <int NS, int NU>
static int launch_predict_big(const sx_gp_model* m, const double* z, int P, double* mean, double* var, double* jac,
                              double* workspace, int64_t workspace_bytes, hipStream_t stream);

template <int NS, int NU>
static int launch_predict(const sx_gp_model* m, const double* z, int P, double* mean, double* var, double* jac,
                          double* workspace, int64_t workspace_bytes, hipStream_t stream) {
    const bool all_at_once = predict_fits(NS, NU, m->n_train, m->n_pad);
    if (!all_at_once && !(NS > 1 && predict_fits(NS, NU, m->n_train, m->n_pad, 1)))
        return launch_predict_big<NS, NU>(m, z, P, mean, var, jac, workspace, workspace_bytes, stream);
    const int nw = kPredictThreads / 64;
    auto gc = make_gp_const<NS, NU>(m, nw);
    const size_t lds = gp_tile_lds_doubles(NS, NS + NU, m->n_train, m->n_pad, nw, all_at_once ? NS : 1) * sizeof(double);
    const int tiles = (P + SX_TILE - 1) / SX_TILE;
    const int grid = tiles < 4096 ? tiles : 4096;
    if (all_at_once) {
        if (int rc = allow_lds(gp_predict_kernel<NS, NU, false>, lds)) return rc;
        hipLaunchKernelGGL((gp_predict_kernel<NS, NU, false>), dim3(grid), dim3(kPredictThreads), lds, stream, gc,
                           gc.stage_tab, z, P, mean, var, jac);
    } else {
        if (int rc = allow_lds(gp_predict_kernel<NS, NU, true>, lds)) return rc;
        hipLaunchKernelGGL((gp_predict_kernel<NS, NU, true>), dim3(grid), dim3(kPredictThreads), lds, stream, gc,
                           gc.stage_tab, z, P, mean, var, jac);
    }
    return check_launch();
}

template <int NS, int NU>
static int launch_reach(const sx_env* env, int P, const double* p, const double* Q, const double* u, const double* mean,
                        const double* var, const double* jac, double* p1, double* Q1, double* sigma, int* status,
                        hipStream_t stream) {
    ReachConst<NS, NU> rc;
    if (!make_reach_const<NS, NU>(env, rc)) return SX_ERR_ARG;
    const int threads = 64;
    hipLaunchKernelGGL(batch_zero_flag_kernel, dim3(1), dim3(1024), 0, stream, var, (int64_t)P * NS, status);
    hipLaunchKernelGGL((onestep_reach_kernel<NS, NU>), dim3((P + threads - 1) / threads), dim3(threads), 0, stream, rc,
                       P, p, Q, u, mean, var, jac, p1, Q1, sigma, status);
    hipLaunchKernelGGL(batch_zero_flag_kernel, dim3(1), dim3(64), 0, stream, var, (int64_t)0, status);  // clears the scratch bit
    return check_launch();
}

template <int NS>
static int launch_polytope(const sx_env* env, int P, const double* p, const double* Q, double c_safety, double* d,
                           uint8_t* inside, hipStream_t stream) {
    PolyArgs<NS> pa;
    std::memset(&pa, 0, sizeof(pa));
    for (int r = 0; r < env->m; ++r) {
        for (int i = 0; i < NS; ++i) pa.h_mat[r * NS + i] = env->h_mat[r * NS + i];
        pa.h_vec[r] = env->h_vec[r];
    }
    pa.m = env->m;
    const int threads = 64;
    hipLaunchKernelGGL((polytope_kernel<NS>), dim3((P + threads - 1) / threads), dim3(threads), 0, stream, pa, P,
                       c_safety, p, Q, d, inside);
    return check_launch();
}

// trmm_reduce_kernel variants, selectable for A/B measurements (tools/cfg4_probe.py):
//   SX_TRMM_ORDER   = tile order bits: 16 paired tiles (default for small grids), 8 longest-first (default otherwise),
//                     1 XCD-contiguous with the row tile fastest, 0 plain;
//                     + 2 / + 4: timing-only diagnostics (every workgroup reads the same Kstar / W tile: no fabric traffic)
//   SX_TRMM_VARIANT = <pairs per chunk><LDS buffers>: 13 (default), 12, 22, 23
// Measured at config 4 (N = 2000, 16 384 particles), per launch: plain order 7.2 ms, XCD-contiguous 4.73 ms, longest-first
// 3.92 ms -- whatever the variant, and the same with the fabric traffic removed (order + 6): the kernel was never
// memory-bound, its tiles differ 16-fold in work and the tail of the launch was what it lost.
static const int g_trmm_order = std::getenv("SX_TRMM_ORDER") ? std::atoi(std::getenv("SX_TRMM_ORDER")) : -1;
static const int g_trmm_variant = std::getenv("SX_TRMM_VARIANT") ? std::atoi(std::getenv("SX_TRMM_VARIANT")) : 13;
//   SX_TRMM_PT      = particle tiles per workgroup: 8 | 4 (default: 4 where 8 would leave a compute unit with at most two
//                     workgroups, see launch_trmm)
static const int g_trmm_pt = std::getenv("SX_TRMM_PT") ? std::atoi(std::getenv("SX_TRMM_PT")) : 0;

template <int NS, int D, int PPC, int NBUF, int PT>
static void launch_trmm_v(int kind, const GpConst<NS, D>& gc, const BigWs& ws, int64_t p128, int row_tiles, hipStream_t stream) {
    constexpr int lds = big_lds_bytes<PPC, NBUF, PT>();
    (void)allow_lds(trmm_reduce_kernel<NS, D, PPC, NBUF, PT>, lds);
    const int64_t pgroups = p128 / (PT * 16);
    const int64_t tiles = pgroups * row_tiles * NS;
    // Tile order.  A large grid runs longest tile first.  A grid of a few rounds is all quantisation: it runs PAIRED tiles
    // (row tile rt and row_tiles - 1 - rt in one workgroup: equal work) when that deals the work out more evenly than
    // longest-first does -- judged by dealing the workgroups round-robin onto the 256 CUs and comparing the fullest CU.
    // SX_TRMM_ORDER overrides.
    int order = g_trmm_order;
    if (order < 0) {
        order = 8;
        if (tiles <= 3 * 768 && row_tiles > 1) {
            const int nrb = gc.n_pad >> 4, groups = (int)pgroups * NS;
            std::vector<double> work(row_tiles);
            for (int rt = 0; rt < row_tiles; ++rt) {
                const int rb0 = rt * kBigRb, rb_end = rb0 + kBigRb < nrb ? rb0 + kBigRb : nrb;
                double w = 0.0;
                for (int rb = rb0; rb < rb0 + kBigRb; ++rb) w += 2 * (rb + 1) < 2 * rb_end ? 2 * (rb + 1) : 2 * rb_end;
                work[rt] = w;
            }
            auto fullest = [&](const std::vector<double>& per_wg) {     // per_wg: work of the workgroups in dispatch order
                double cu[256] = {0.0};
                for (size_t i = 0; i < per_wg.size(); ++i) cu[i & 255] += per_wg[i];
                double m = 0.0;
                for (double v : cu) m = v > m ? v : m;
                return m;
            };
            std::vector<double> plain, paired;
            for (int rt = row_tiles - 1; rt >= 0; --rt) plain.insert(plain.end(), groups, work[rt]);
            for (int j = 0; j < (row_tiles + 1) / 2; ++j)
                paired.insert(paired.end(), groups, work[row_tiles - 1 - j] + (j != row_tiles - 1 - j ? work[j] : 0.0));
            if (fullest(paired) < fullest(plain)) order = 16;
        }
    }
    const dim3 grid((unsigned)((order & 16) ? pgroups * ((row_tiles + 1) / 2) * NS : tiles));
    if (kind >= 0)
        launch(kind, trmm_reduce_kernel<NS, D, PPC, NBUF, PT>, grid, dim3(kBigThreads), lds, stream, gc, ws, p128, row_tiles, order);
    else
        hipLaunchKernelGGL((trmm_reduce_kernel<NS, D, PPC, NBUF, PT>), grid, dim3(kBigThreads), lds, stream, gc, ws, p128, row_tiles,
                           order);
}

template <int NS, int D>
static void launch_trmm(int kind, const GpConst<NS, D>& gc, const BigWs& ws, int64_t p128, int row_tiles, hipStream_t stream) {
    // Small grids take 128 x 64 tiles: with 128 x 128 a grid of up to two workgroups per compute unit (N ~ 1000 .. 1400 at
    // 4096 particles) leaves each SIMD one or two waves that idle through every barrier and DMA wait; twice the workgroups
    // at half the size fill those gaps (tools/n_sweep.sh: the step at the path switch).
    const int64_t wgs8 = (p128 / kBigTile) * row_tiles * NS;
    const bool half = g_trmm_pt ? g_trmm_pt == 4 : wgs8 <= 2 * 768;
    if (half) return launch_trmm_v<NS, D, 1, 3, 4>(kind, gc, ws, p128, row_tiles, stream);
    switch (g_trmm_variant) {
        case 12: return launch_trmm_v<NS, D, 1, 2, 8>(kind, gc, ws, p128, row_tiles, stream);
        case 22: return launch_trmm_v<NS, D, 2, 2, 8>(kind, gc, ws, p128, row_tiles, stream);
        case 23: return launch_trmm_v<NS, D, 2, 3, 8>(kind, gc, ws, p128, row_tiles, stream);
        default: return launch_trmm_v<NS, D, 1, 3, 8>(kind, gc, ws, p128, row_tiles, stream);
    }
}

// ---- sx_gp_predict for training sets beyond the LDS budget: the same Kstar / triangular-product kernels, then collect ----
template <int NS, int D>
__global__ void predict_init_big_kernel(const double* __restrict__ z, int64_t P, int64_t p128, BigWs ws) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= p128 * D) return;
    ws.zs[i] = (i < P * D) ? z[i] : 0.0;
}

template <int NS, int D>
__global__ void predict_collect_big_kernel(GpConst<NS, D> gc, BigWs ws, int64_t P, int64_t p128, int row_parts,
                                           double* __restrict__ mean, double* __restrict__ var, double* __restrict__ jac) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (g >= P) return;
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        double q = 0.0;
        for (int r = 0; r < row_parts; ++r) q += ws.part[((int64_t)d * row_parts + r) * p128 + g];
        var[g * NS + d] = (gc.outputscale[d] - q) + gc.noise[d];
        const double m = ws.mj[((int64_t)d * (D + 1)) * p128 + g];
        mean[g * NS + d] = m;
        if (jac) {
#pragma unroll
            for (int j = 0; j < D; ++j)
                jac[(g * NS + d) * D + j] =
                    ws.mj[((int64_t)d * (D + 1) + 1 + j) * p128 + g] - ws.zs[g * D + j] * gc.inv_ls2[d * D + j] * m;
        }
    }
}

template <int NS, int NU>
static int launch_predict_big(const sx_gp_model* m, const double* z, int P, double* mean, double* var, double* jac,
                              double* workspace, int64_t workspace_bytes, hipStream_t stream) {
    constexpr int D = NS + NU;
    auto gc = make_gp_const<NS, NU>(m, kPredictThreads / 64);
    const int64_t p128 = ((int64_t)P + kBigTile - 1) / kBigTile * kBigTile;
    BigWs ws = big_ws_layout(workspace, NS, D, m->n_pad, P);
    if (!workspace || workspace_bytes < ws.total * (int64_t)sizeof(double)) return SX_ERR_ARG;
    const int row_tiles = (m->n_pad + kBigTile - 1) / kBigTile;
    hipLaunchKernelGGL((predict_init_big_kernel<NS, D>), dim3((unsigned)((p128 * D + 255) / 256)), dim3(256), 0, stream, z,
                       (int64_t)P, p128, ws);
    hipLaunchKernelGGL((kstar_big_kernel<NS, D>), dim3((unsigned)(p128 / 16), (unsigned)((m->n_pad + 255) / 256)), dim3(256),
                       0, stream, gc, ws);
    launch_trmm<NS, D>(-1, gc, ws, p128, row_tiles, stream);
    hipLaunchKernelGGL((predict_collect_big_kernel<NS, D>), dim3((unsigned)((P + 63) / 64)), dim3(64), 0, stream, gc, ws,
                       (int64_t)P, p128, row_tiles * 2, mean, var, jac);
    return check_launch();
}

// does the single-launch kernel's LDS budget hold Kstar for this model -- of all outputs at once (ns_lds = ns), or of one
// output at a time (ns_lds = 1)?
static bool fused_fits(int ns, int nu, int n_train, int n_pad, int H, int ns_lds = -1) {
    const int nw = kRolloutThreads / 64;
    const size_t lds =
        (gp_tile_lds_doubles(ns, ns + nu, n_train, n_pad, nw, ns_lds) + (size_t)SX_TILE * H * nu) * sizeof(double);
    return lds <= kMaxLdsBytes && n_pad <= 1024;
}


template <int NS, int NU>
static int launch_rollout_big(const sx_gp_model* m, const sx_env* env, const RolloutPtrs& rp, double* workspace,
                              int64_t workspace_bytes, hipStream_t stream) {
    constexpr int D = NS + NU;
    auto gc = make_gp_const<NS, NU>(m, kRolloutThreads / 64);
    ReachConst<NS, NU> rc;
    if (!make_reach_const<NS, NU>(env, rc)) return SX_ERR_ARG;
    CostConst<SX_MAX_M, NS, NU> cc;
    make_cost_const<NS, NU>(env, cc);
    const int64_t total = (int64_t)rp.E * rp.P;
    const int64_t p128 = (total + kBigTile - 1) / kBigTile * kBigTile;
    BigWs ws = big_ws_layout(workspace, NS, D, m->n_pad, total);
    if (!workspace || workspace_bytes < ws.total * (int64_t)sizeof(double)) return SX_ERR_ARG;
    const int row_tiles = (m->n_pad + kBigTile - 1) / kBigTile;
    BigInit bi{rp.x0, rp.q0, rp.mean, rp.std, rp.noise, rp.actions, rp.obj_cost, rp.con_cost, rp.P, rp.H};
    hipLaunchKernelGGL((init_big_kernel<NS, NU>), dim3((unsigned)((p128 + 255) / 256)), dim3(256), 0, stream, bi, ws, total,
                       p128);
    for (int t = 0; t < rp.H; ++t) {
        launch(SX_PROF_KSTAR_BIG, kstar_big_kernel<NS, D>, dim3((unsigned)(p128 / 16), (unsigned)((m->n_pad + 255) / 256)),
               dim3(256), 0, stream, gc, ws);
        launch_trmm<NS, D>(SX_PROF_TRMM_BIG, gc, ws, p128, row_tiles, stream);
        BigStep bs{rp.actions, rp.traj, rp.sigma, rp.obj_cost, rp.con_cost, rp.status, rp.H, t, row_tiles * 2,
                   (t > 0 || rp.q0 != nullptr) ? 1 : 0};
        launch(SX_PROF_STEP_BIG, step_big_kernel<NS, NU>, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, stream, gc, rc, cc,
               bs, ws, total, p128);
    }
    return check_launch();
}

template <int NS, int NU>
static int launch_rollout(const sx_gp_model* m, const sx_env* env, const RolloutPtrs& rp, double* workspace,
                          int64_t workspace_bytes, hipStream_t stream) {
    const bool all_at_once = fused_fits(NS, NU, m->n_train, m->n_pad, rp.H);
    if (!all_at_once && !(NS > 1 && fused_fits(NS, NU, m->n_train, m->n_pad, rp.H, 1))) {
        if (rp.elite_rows) return SX_ERR_UNSUPPORTED;   // the refit prologue belongs to the single-launch kernel
        return launch_rollout_big<NS, NU>(m, env, rp, workspace, workspace_bytes, stream);
    }
    // (the refit prologue keeps 2 H n_u doubles in the Kstar / mean-row buffers: at least 256 + 256 NS of them)
    if (rp.elite_rows && 2 * rp.H * NU > 256 + 256 * NS) return SX_ERR_UNSUPPORTED;
    const int nw = kRolloutThreads / 64;
    auto gc = make_gp_const<NS, NU>(m, nw);
    ReachConst<NS, NU> rc;
    if (!make_reach_const<NS, NU>(env, rc)) return SX_ERR_ARG;
    CostConst<SX_MAX_M, NS, NU> cc;
    make_cost_const<NS, NU>(env, cc);
    const size_t lds = (gp_tile_lds_doubles(NS, NS + NU, m->n_train, m->n_pad, nw, all_at_once ? NS : 1) +
                        (size_t)SX_TILE * rp.H * NU) * sizeof(double);
    const int tiles = (rp.P + SX_TILE - 1) / SX_TILE;
    if (all_at_once) {
        // Three forms of the kernel: W partly resident on 8 waves (sx_rollout_rh.hpp: n_s <= 2), all of
        // W in the registers of 4 waves (sx_rollout_rw.hpp: every n_s, smaller N), W streamed from L2 (cem_rollout_kernel: any
        // size that fits the LDS).  By default the first of the three that is instantiated for the shape: the 4-wave form
        // loses to the streaming kernel only where the 8-wave form exists (n_s = 2, n_u = 1: 126.6 against 125.7 us at config
        // 2), and beats it by 7 - 16 % on the shapes the 8-wave form does not cover (n_s = 3, 4; n_s = n_u = 2 beyond N = 128).
        // SX_ROLLOUT=rh|rw|stream picks ONE form for A/B runs (falling back to the streaming kernel where it is not
        // instantiated, unless SX_ROLLOUT_STRICT is set, so that a run knows what it timed).
        static const int form = [] {
            const char* e = std::getenv("SX_ROLLOUT");
            if (e && std::strcmp(e, "stream") == 0) return 0;
            if (e && std::strcmp(e, "rw") == 0) return 1;
            if (e && std::strcmp(e, "rh") == 0) return 2;
            return 3;
        }();
        if (form != 0) {
            static const bool strict = std::getenv("SX_ROLLOUT_STRICT") != nullptr;
            RolloutPtrs rps = rp;
#ifdef SX_STAMPS
            rps.stamps = g_stamp_host;
#endif
            int r = SX_ERR_UNSUPPORTED;
            if (form >= 2) r = launch_rollout_rh<NS, NU>(make_gp_const<NS, NU>(m, 8), rc, cc, rps, stream);
            if (r == SX_ERR_UNSUPPORTED && form != 2)
                r = launch_rollout_rw<NS, NU>(make_gp_const<NS, NU>(m, kRwWaves), rc, cc, rps, stream);
            if (r != SX_ERR_UNSUPPORTED || (strict && form != 3)) return r;
        }
        if (int r = allow_lds(cem_rollout_kernel<NS, NU, false>, lds)) return r;
        launch(SX_PROF_ROLLOUT_FUSED, cem_rollout_kernel<NS, NU, false>, dim3(rp.E * tiles), dim3(kRolloutThreads), lds, stream,
               gc, gc.stage_tab, rc, cc, rp);
    } else {
        if (int r = allow_lds(cem_rollout_kernel<NS, NU, true>, lds)) return r;
        launch(SX_PROF_ROLLOUT_FUSED, cem_rollout_kernel<NS, NU, true>, dim3(rp.E * tiles), dim3(kRolloutThreads), lds, stream,
               gc, gc.stage_tab, rc, cc, rp);
    }
    return check_launch();
}

}  // namespace sx

namespace sx {
template <int NS, int NU>
static int launch_feat_predict(const sx_feat_model* m, const double* z, int P, double* mean, double* var, double* jac,
                               hipStream_t stream) {
    const FeatConst fc = make_feat_const(m);
    const size_t lds = kFeatLdsDoubles * sizeof(double);
    if (int rc = allow_lds(feat_predict_kernel<NS, NU>, lds)) return rc;
    hipLaunchKernelGGL((feat_predict_kernel<NS, NU>), dim3((P + kFeatWave - 1) / kFeatWave), dim3(kFeatWave), lds, stream, fc, z,
                       P, mean, var, jac);
    return check_launch();
}

template <int NS, int NU>
static int launch_rollout_feat(const sx_feat_model* m, const sx_env* env, const FeatRolloutPtrs& rp, hipStream_t stream) {
    const FeatConst fc = make_feat_const(m);
    ReachConst<NS, NU> rc;
    if (!make_reach_const<NS, NU>(env, rc)) return SX_ERR_ARG;
    CostConst<SX_MAX_M, NS, NU> cc;
    make_cost_const<NS, NU>(env, cc);
    const size_t lds = kFeatLdsDoubles * sizeof(double);
    if (int r = allow_lds(cem_rollout_feat_kernel<NS, NU>, lds)) return r;
    const int64_t total = (int64_t)rp.E * rp.P;
    launch(SX_PROF_ROLLOUT_FEAT, cem_rollout_feat_kernel<NS, NU>, dim3((unsigned)((total + kFeatWave - 1) / kFeatWave)),
           dim3(kFeatWave), lds, stream, fc, rc, cc, rp);
    return check_launch();
}
}  // namespace sx

namespace sx {
// SX_MLP_PATH=valu keeps every network on the one-particle-per-lane kernel (A/B runs; the default is the matrix-core
// kernel wherever mlp_mfma_ok() holds)
static bool mlp_use_mfma(const MlpConst& mc) {
    const char* e = getenv("SX_MLP_PATH");   // read per launch, so that a test can switch between the two kernels
    return !(e && strcmp(e, "valu") == 0) && mlp_mfma_ok(mc);
}

template <int NS, int NU, int L, bool FULL>
static int launch_mlp_predict_mfma(const MlpConst& mc, const double* z, int P, double* mean, double* var, double* jac,
                                   hipStream_t stream) {
    const size_t lds = (size_t)MmLds<NS, NS + NU>::total * sizeof(double);
    if (int rc = allow_lds(mlp_predict_mfma_kernel<NS, NU, L, FULL>, lds)) return rc;
    hipLaunchKernelGGL((mlp_predict_mfma_kernel<NS, NU, L, FULL>), dim3((P + kMmTile - 1) / kMmTile), dim3(kMmThreads), lds, stream,
                       mc, z, P, mean, var, jac);
    return check_launch();
}

template <int NS, int NU>
static int launch_mlp_predict(const sx_mlp_model* m, const double* z, int P, double* mean, double* var, double* jac,
                              hipStream_t stream) {
    const MlpConst mc = make_mlp_const(m);
    if (mlp_use_mfma(mc)) {
        const bool full = mlp_mfma_full(mc);
        if (mc.n_hidden == 1)
            return full ? launch_mlp_predict_mfma<NS, NU, 1, true>(mc, z, P, mean, var, jac, stream)
                        : launch_mlp_predict_mfma<NS, NU, 1, false>(mc, z, P, mean, var, jac, stream);
        return full ? launch_mlp_predict_mfma<NS, NU, 2, true>(mc, z, P, mean, var, jac, stream)
                    : launch_mlp_predict_mfma<NS, NU, 2, false>(mc, z, P, mean, var, jac, stream);
    }
    const size_t lds = mlp_lds_doubles(mc.n_hidden, mc.wmax) * sizeof(double);
    if (int rc = allow_lds(mlp_predict_kernel<NS, NU>, lds)) return rc;
    hipLaunchKernelGGL((mlp_predict_kernel<NS, NU>), dim3((P + kMlpLanes - 1) / kMlpLanes), dim3(kMlpLanes), lds, stream, mc, z,
                       P, mean, var, jac);
    return check_launch();
}

template <int NS, int NU, int L, bool FULL>
static int launch_rollout_mlp_mfma(const MlpConst& mc, const ReachConst<NS, NU>& rc, const CostConst<SX_MAX_M, NS, NU>& cc,
                                   const FeatRolloutPtrs& rp, hipStream_t stream) {
    const size_t lds = (size_t)MmLds<NS, NS + NU>::total * sizeof(double);
    if (int r = allow_lds(cem_rollout_mlp_mfma_kernel<NS, NU, L, FULL>, lds)) return r;
    const int64_t total = (int64_t)rp.E * rp.P;
    launch(SX_PROF_ROLLOUT_MLP, cem_rollout_mlp_mfma_kernel<NS, NU, L, FULL>, dim3((unsigned)((total + kMmTile - 1) / kMmTile)),
           dim3(kMmThreads), lds, stream, mc, rc, cc, rp);
    return check_launch();
}

template <int NS, int NU>
static int launch_rollout_mlp(const sx_mlp_model* m, const sx_env* env, const FeatRolloutPtrs& rp, hipStream_t stream) {
    const MlpConst mc = make_mlp_const(m);
    ReachConst<NS, NU> rc;
    if (!make_reach_const<NS, NU>(env, rc)) return SX_ERR_ARG;
    CostConst<SX_MAX_M, NS, NU> cc;
    make_cost_const<NS, NU>(env, cc);
    if (mlp_use_mfma(mc)) {
        const bool full = mlp_mfma_full(mc);
        if (mc.n_hidden == 1)
            return full ? launch_rollout_mlp_mfma<NS, NU, 1, true>(mc, rc, cc, rp, stream)
                        : launch_rollout_mlp_mfma<NS, NU, 1, false>(mc, rc, cc, rp, stream);
        return full ? launch_rollout_mlp_mfma<NS, NU, 2, true>(mc, rc, cc, rp, stream)
                    : launch_rollout_mlp_mfma<NS, NU, 2, false>(mc, rc, cc, rp, stream);
    }
    const size_t lds = mlp_lds_doubles(mc.n_hidden, mc.wmax) * sizeof(double);
    if (int r = allow_lds(cem_rollout_mlp_kernel<NS, NU>, lds)) return r;
    const int64_t total = (int64_t)rp.E * rp.P;
    launch(SX_PROF_ROLLOUT_MLP, cem_rollout_mlp_kernel<NS, NU>, dim3((unsigned)((total + kMlpLanes - 1) / kMlpLanes)),
           dim3(kMlpLanes), lds, stream, mc, rc, cc, rp);
    return check_launch();
}
}  // namespace sx

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
#define SX_DISPATCH(ns, nu, CALL)                              \
    do {                                                       \
        if ((ns) == 2 && (nu) == 1) return CALL(2, 1);         \
        if ((ns) == 4 && (nu) == 1) return CALL(4, 1);         \
        if ((ns) == 2 && (nu) == 2) return CALL(2, 2);         \
        if ((ns) == 4 && (nu) == 2) return CALL(4, 2);         \
        if ((ns) == 3 && (nu) == 1) return CALL(3, 1);         \
        if ((ns) == 1 && (nu) == 1) return CALL(1, 1);         \
        return SX_ERR_UNSUPPORTED;                             \
    } while (0)

extern "C" {

#ifdef SX_STAMPS
int sx_debug_set_stamps(unsigned long long* dev_buf) {
    sx::g_stamp_host = dev_buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(sx::g_stamp_buf), &dev_buf, sizeof(dev_buf)) == hipSuccess ? SX_OK : SX_ERR_LAUNCH;
}
#endif

const char* sx_version(void) { return "sxamd 0.3 gfx950"; }

int sx_profile_enable(int max_launches) {
    if (max_launches <= 0) return SX_ERR_ARG;
    std::lock_guard<std::mutex> lock(sx::g_prof_mu);
    for (auto& e : sx::g_prof_entries) {
        sx::g_prof_pool.push_back(e.start);
        sx::g_prof_pool.push_back(e.stop);
    }
    sx::g_prof_entries.clear();
    sx::g_prof_cap = (size_t)max_launches;
    for (long& n : sx::g_prof_seen) n = 0;
    sx::g_prof_on = true;
    return SX_OK;
}

int sx_profile_stride(int every) {
    if (every <= 0) return SX_ERR_ARG;
    std::lock_guard<std::mutex> lock(sx::g_prof_mu);
    for (int& v : sx::g_prof_stride) v = every;
    return SX_OK;
}

int sx_profile_stride_kind(int kind, int every) {
    if (every <= 0 || kind < 0 || kind >= SX_PROF_KINDS) return SX_ERR_ARG;
    std::lock_guard<std::mutex> lock(sx::g_prof_mu);
    sx::g_prof_stride[kind] = every;
    return SX_OK;
}

int sx_profile_collect(int kind, double* total_ms, int64_t* launches) {
    if (kind < 0 || kind >= SX_PROF_KINDS || !total_ms || !launches) return SX_ERR_ARG;
    std::lock_guard<std::mutex> lock(sx::g_prof_mu);
    double tot = 0.0;
    int64_t n = 0;
    for (auto& e : sx::g_prof_entries) {
        if (e.kind != kind) continue;
        float ms = 0.f;
        if (hipEventSynchronize(e.stop) != hipSuccess || hipEventElapsedTime(&ms, e.start, e.stop) != hipSuccess) {
            (void)hipGetLastError();
            return SX_ERR_LAUNCH;
        }
        tot += ms;
        ++n;
    }
    *total_ms = tot;
    *launches = n;
    return SX_OK;
}

int sx_profile_disable(void) {
    std::lock_guard<std::mutex> lock(sx::g_prof_mu);
    sx::g_prof_on = false;
    for (auto& e : sx::g_prof_entries) {
        sx::g_prof_pool.push_back(e.start);
        sx::g_prof_pool.push_back(e.stop);
    }
    sx::g_prof_entries.clear();
    return SX_OK;
}

int sx_gp_pack_sizes(int n_s, int n_u, int n_train, int64_t* a_doubles, int64_t* tab_ints) {
    if (n_s <= 0 || n_s > SX_MAX_NS || n_u <= 0 || n_u > SX_MAX_NU || n_train <= 0) return SX_ERR_ARG;
    const int n_pad = sx::gp_n_pad(n_train, n_s + n_u);
    if (a_doubles) *a_doubles = sx::a_pack_doubles(n_s, n_pad);
    if (tab_ints) *tab_ints = sx::gp_stage_tab_ints(n_s, n_pad, SX_WAVES);
    return SX_OK;
}

int sx_gp_fit(const sx_gp_model* model, const double* y_train, double* work, double* linv, double* alpha,
              double* logdet, int32_t* status, void* stream) {
    if (!model || !model->x_train || !y_train || !work || !linv || !alpha || !logdet || !status) return SX_ERR_ARG;
    if (model->n_s <= 0 || model->n_s > SX_MAX_NS || model->n_u <= 0 || model->n_u > SX_MAX_NU || model->n_train <= 0)
        return SX_ERR_ARG;
    if (model->n_train > sx::kFitMaxN) return SX_ERR_UNSUPPORTED;
    sx::FitArgs fa;
    std::memset(&fa, 0, sizeof(fa));
    const int D = model->n_s + model->n_u;
    for (int i = 0; i < model->n_s * D; ++i) fa.inv_ls2[i] = model->inv_ls2[i];
    for (int i = 0; i < model->n_s; ++i) {
        fa.outputscale[i] = model->outputscale[i];
        fa.noise[i] = model->noise[i];
    }
    if (model->n_train > sx::kBlockedFitMinN) {
        // blocked multi-workgroup path (sx_fit_blocked.hpp)
        sx::BlockedFitArgs ba;
        std::memset(&ba, 0, sizeof(ba));
        std::memcpy(ba.inv_ls2, fa.inv_ls2, sizeof(ba.inv_ls2));
        std::memcpy(ba.outputscale, fa.outputscale, sizeof(ba.outputscale));
        std::memcpy(ba.noise, fa.noise, sizeof(ba.noise));
        ba.x = model->x_train;
        ba.y = y_train;
        ba.lmat = work;
        ba.linv = linv;
        ba.alpha = alpha;
        ba.logdet = logdet;
        ba.status = status;
        ba.n = model->n_train;
        ba.D = D;
        ba.n_s = model->n_s;
        ba.nblk = (ba.n + sx::kFB - 1) / sx::kFB;
        hipStream_t s = (hipStream_t)stream;
        const int nb = ba.nblk, ns = ba.n_s;
        hipLaunchKernelGGL(sx::fit_kmat_kernel, dim3(nb, nb, ns), dim3(sx::kFThreads), 0, s, ba);
        for (int p = 0; p < nb; ++p) {
            hipLaunchKernelGGL(sx::fit_potrf_diag_kernel, dim3(ns), dim3(sx::kFThreads), 0, s, ba, p);
            const int m = nb - p - 1;
            if (m > 0) {
                hipLaunchKernelGGL(sx::fit_trsm_kernel, dim3(m, ns), dim3(sx::kFThreads), 0, s, ba, p);
                hipLaunchKernelGGL(sx::fit_syrk_kernel, dim3(m, m, ns), dim3(sx::kFThreads), 0, s, ba, p);
            }
        }
        hipLaunchKernelGGL(sx::fit_trtri_kernel, dim3(nb, ns, sx::kFB / 16), dim3(sx::kFThreads), 0, s, ba);
        hipLaunchKernelGGL(sx::fit_alpha_logdet_kernel, dim3(ns), dim3(1024), sizeof(double) * (size_t)ba.n, s, ba);
        return sx::check_launch();
    }
    fa.x = model->x_train;
    fa.y = y_train;
    fa.lmat = work;
    fa.linv = linv;
    fa.alpha = alpha;
    fa.logdet = logdet;
    fa.status = status;
    fa.n = model->n_train;
    fa.D = D;
    fa.n_s = model->n_s;
    // panel width: as wide as the LDS left beside the static arrays allows (vec 32 KB + red 8 KB), at most 32
    const size_t lds_budget = 112 * 1024;
    int nb = (int)(lds_budget / (sizeof(double) * (size_t)fa.n));
    nb = nb > 32 ? 32 : (nb < 1 ? 1 : nb);
    fa.panel_cols = nb;
    const size_t lds = sizeof(double) * (size_t)fa.n * nb;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(sx::gp_fit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return SX_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(sx::gp_fit_kernel, dim3(model->n_s), dim3(sx::kFitThreads), lds, (hipStream_t)stream, fa);
    return sx::check_launch();
}

int sx_gp_mll_grad(const sx_gp_model* model, const double* y_train, const double* linv, const double* alpha,
                   const double* logdet, double* work, double* mll, double* grad, void* stream) {
    if (!model || !model->x_train || !y_train || !linv || !alpha || !logdet || !work || !mll || !grad) return SX_ERR_ARG;
    if (model->n_s <= 0 || model->n_s > SX_MAX_NS || model->n_u <= 0 || model->n_u > SX_MAX_NU || model->n_train <= 0)
        return SX_ERR_ARG;
    if (model->n_train > sx::kBlockedFitMinN) {
        sx::BlockedMllArgs ba;
        std::memset(&ba, 0, sizeof(ba));
        const int Db = model->n_s + model->n_u;
        for (int i = 0; i < model->n_s * Db; ++i) ba.inv_ls2[i] = model->inv_ls2[i];
        for (int i = 0; i < model->n_s; ++i) ba.outputscale[i] = model->outputscale[i];
        ba.x = model->x_train;
        ba.y = y_train;
        ba.linv = linv;
        ba.alpha = alpha;
        ba.logdet = logdet;
        ba.scratch = work;
        ba.mll = mll;
        ba.grad = grad;
        ba.n = model->n_train;
        ba.D = Db;
        ba.n_s = model->n_s;
        ba.nblk = (ba.n + sx::kFB - 1) / sx::kFB;
        hipLaunchKernelGGL(sx::mll_pairs_kernel, dim3(ba.nblk, ba.nblk, ba.n_s), dim3(sx::kFThreads), 0, (hipStream_t)stream, ba);
        hipLaunchKernelGGL(sx::mll_reduce_kernel, dim3(ba.n_s), dim3(256), 0, (hipStream_t)stream, ba);
        return sx::check_launch();
    }
    sx::MllArgs ma;
    std::memset(&ma, 0, sizeof(ma));
    const int D = model->n_s + model->n_u;
    for (int i = 0; i < model->n_s * D; ++i) ma.inv_ls2[i] = model->inv_ls2[i];
    for (int i = 0; i < model->n_s; ++i) {
        ma.outputscale[i] = model->outputscale[i];
        ma.noise[i] = model->noise[i];
    }
    ma.x = model->x_train;
    ma.y = y_train;
    ma.linv = linv;
    ma.alpha = alpha;
    ma.logdet = logdet;
    ma.mll = mll;
    ma.grad = grad;
    ma.n = model->n_train;
    ma.D = D;
    ma.n_s = model->n_s;
    hipLaunchKernelGGL(sx::gp_mll_grad_kernel, dim3(model->n_s), dim3(sx::kFitThreads), 0, (hipStream_t)stream, ma);
    return sx::check_launch();
}

int sx_gp_predict_var_jac(const sx_gp_model* model, const double* linv, const double* z, int P, double* jac_var,
                          void* stream) {
    if (!model || P < 0) return SX_ERR_ARG;
    if (P == 0) return SX_OK;
    if (!model->x_train || !linv || !z || !jac_var) return SX_ERR_ARG;
    if (model->n_s <= 0 || model->n_s > SX_MAX_NS || model->n_u <= 0 || model->n_u > SX_MAX_NU || model->n_train <= 0)
        return SX_ERR_ARG;
    const size_t lds = 2 * (size_t)model->n_train * sizeof(double);
    if (lds > 128 * 1024) return SX_ERR_UNSUPPORTED;  // N <= 8192
    sx::VarJacArgs va;
    std::memset(&va, 0, sizeof(va));
    const int D = model->n_s + model->n_u;
    for (int i = 0; i < model->n_s * D; ++i) va.inv_ls2[i] = model->inv_ls2[i];
    for (int i = 0; i < model->n_s; ++i) va.outputscale[i] = model->outputscale[i];
    va.x = model->x_train;
    va.linv = linv;
    va.z = z;
    va.jac_var = jac_var;
    va.n = model->n_train;
    va.D = D;
    va.n_s = model->n_s;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)sx::gp_var_jac_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SX_ERR_LAUNCH;
    hipLaunchKernelGGL(sx::gp_var_jac_kernel, dim3(P, model->n_s), dim3(256), lds, (hipStream_t)stream, va);
    return sx::check_launch();
}

int sx_gp_predict_mean_hessian(const sx_gp_model* model, const double* alpha, const double* z, int P, double* hess,
                               void* stream) {
    if (!model || P < 0) return SX_ERR_ARG;
    if (P == 0) return SX_OK;
    if (!model->x_train || !alpha || !z || !hess) return SX_ERR_ARG;
    if (model->n_s <= 0 || model->n_s > SX_MAX_NS || model->n_u <= 0 || model->n_u > SX_MAX_NU || model->n_train <= 0)
        return SX_ERR_ARG;
    sx::MeanHessArgs ha;
    std::memset(&ha, 0, sizeof(ha));
    const int D = model->n_s + model->n_u;
    for (int i = 0; i < model->n_s * D; ++i) ha.inv_ls2[i] = model->inv_ls2[i];
    for (int i = 0; i < model->n_s; ++i) ha.outputscale[i] = model->outputscale[i];
    ha.x = model->x_train;
    ha.alpha = alpha;
    ha.z = z;
    ha.hess = hess;
    ha.n = model->n_train;
    ha.D = D;
    ha.n_s = model->n_s;
    hipLaunchKernelGGL(sx::gp_mean_hessian_kernel, dim3(P, model->n_s), dim3(256), 0, (hipStream_t)stream, ha);
    return sx::check_launch();
}

int sx_gp_pack(sx_gp_model* model, const double* linv, const double* alpha, void* stream) {
    if (!model || !linv || !alpha || !model->x_train || !model->a_pack || !model->stage_tab) return SX_ERR_ARG;
    if (model->n_s <= 0 || model->n_s > SX_MAX_NS || model->n_u <= 0 || model->n_u > SX_MAX_NU || model->n_train <= 0)
        return SX_ERR_ARG;
    const int D = model->n_s + model->n_u;
    model->n_pad = sx::gp_n_pad(model->n_train, D);
    const bool has_tab = model->n_pad <= 1024;  // beyond that only the large-training-set path runs (no stage table)
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = sx::a_pack_doubles(model->n_s, model->n_pad);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    sx::PackArgs<SX_MAX_NS, SX_MAX_D> args;
    std::memset(&args, 0, sizeof(args));
    for (int i = 0; i < model->n_s * D; ++i) args.inv_ls2[i] = model->inv_ls2[i];
    hipLaunchKernelGGL(sx::pack_a_kernel, dim3(grid), dim3(256), 0, s, linv, alpha, model->x_train, args, model->n_s, D,
                       model->n_train, model->n_pad, const_cast<double*>(model->a_pack));
    if (has_tab)
        hipLaunchKernelGGL(sx::build_stage_tab_kernel, dim3(1), dim3(64), 0, s,
                           reinterpret_cast<int4*>(const_cast<int32_t*>(model->stage_tab)), model->n_s, model->n_train,
                           model->n_pad, SX_WAVES, sx::gp_stage_cap(model->n_s, model->n_pad, SX_WAVES),
                           sx::gp_stage_cap(1, model->n_pad, SX_WAVES));
    return sx::check_launch();
}

int64_t sx_gp_predict_workspace_bytes(const sx_gp_model* model, int P) {
    if (!model || P < 0) return -1;
    if (sx::predict_fits(model->n_s, model->n_u, model->n_train, model->n_pad)) return 0;
    if (model->n_s > 1 && sx::predict_fits(model->n_s, model->n_u, model->n_train, model->n_pad, 1)) return 0;
    return sx::big_ws_layout(nullptr, model->n_s, model->n_s + model->n_u, model->n_pad, P).total * (int64_t)sizeof(double);
}

int sx_gp_predict(const sx_gp_model* model, const double* z, int P, double* mean, double* var, double* jac,
                  void* workspace, int64_t workspace_bytes, void* stream) {
    if (!model || P < 0) return SX_ERR_ARG;
    if (P == 0) return SX_OK;   // an empty batch is not an error (its pointers may be NULL)
    if (!z || !mean || !var) return SX_ERR_ARG;
#define CALL(NS, NU) \
    sx::launch_predict<NS, NU>(model, z, P, mean, var, jac, (double*)workspace, workspace_bytes, (hipStream_t)stream)
    SX_DISPATCH(model->n_s, model->n_u, CALL);
#undef CALL
}

int sx_onestep_reach(const sx_env* env, int P, const double* p, const double* Q, const double* u, const double* mean,
                     const double* var, const double* jac, double* p1, double* Q1, double* sigma, int32_t* status,
                     void* stream) {
    if (!env || !p || !u || !mean || !var || !p1 || !Q1 || !sigma || !status || P < 0) return SX_ERR_ARG;
    if (Q && !jac) return SX_ERR_ARG;
    if (P == 0) return SX_OK;
#define CALL(NS, NU) \
    sx::launch_reach<NS, NU>(env, P, p, Q, u, mean, var, jac, p1, Q1, sigma, status, (hipStream_t)stream)
    SX_DISPATCH(env->n_s, env->n_u, CALL);
#undef CALL
}

int sx_polytope_distance(const sx_env* env, int P, const double* p, const double* Q, double c_safety, double* d,
                         uint8_t* inside, void* stream) {
    if (!env || !p || !Q || !d || P < 0) return SX_ERR_ARG;
    if (env->m <= 0 || env->m > SX_MAX_M) return SX_ERR_UNSUPPORTED;
    if (P == 0) return SX_OK;
    switch (env->n_s) {
        case 1: return sx::launch_polytope<1>(env, P, p, Q, c_safety, d, inside, (hipStream_t)stream);
        case 2: return sx::launch_polytope<2>(env, P, p, Q, c_safety, d, inside, (hipStream_t)stream);
        case 3: return sx::launch_polytope<3>(env, P, p, Q, c_safety, d, inside, (hipStream_t)stream);
        case 4: return sx::launch_polytope<4>(env, P, p, Q, c_safety, d, inside, (hipStream_t)stream);
        default: return SX_ERR_UNSUPPORTED;
    }
}

int64_t sx_cem_rollout_workspace_bytes(const sx_gp_model* model, int E, int P, int H) {
    if (!model || E <= 0 || P <= 0 || H <= 0) return -1;
    if (sx::fused_fits(model->n_s, model->n_u, model->n_train, model->n_pad, H)) return 0;
    if (model->n_s > 1 && sx::fused_fits(model->n_s, model->n_u, model->n_train, model->n_pad, H, 1)) return 0;
    return sx::big_ws_layout(nullptr, model->n_s, model->n_s + model->n_u, model->n_pad, (int64_t)E * P).total *
           (int64_t)sizeof(double);
}

int sx_cem_rollout_form(const sx_gp_model* model, int H) {
    if (!model || H <= 0) return -1;
    const int ns = model->n_s, nu = model->n_u;
    if (!sx::fused_fits(ns, nu, model->n_train, model->n_pad, H))
        return (ns > 1 && sx::fused_fits(ns, nu, model->n_train, model->n_pad, H, 1)) ? SX_FORM_BYOUT : SX_FORM_BIG;
    const char* e = std::getenv("SX_ROLLOUT");
    if (e && std::strcmp(e, "stream") == 0) return SX_FORM_STREAM;
    const bool only_rw = e && std::strcmp(e, "rw") == 0, only_rh = e && std::strcmp(e, "rh") == 0;
#define CALL(NS, NU) \
    ((!only_rw && sx::rollout_rh_applies<NS, NU>(model->n_train, model->n_pad, H)) ? SX_FORM_RH \
     : ((!only_rh && sx::rollout_rw_applies<NS, NU>(model->n_train, model->n_pad, H)) ? SX_FORM_RW : SX_FORM_STREAM))
    SX_DISPATCH(ns, nu, CALL);
#undef CALL
}

int sx_cem_rollout(const sx_gp_model* model, const sx_env* env, int E, int P, int H, const double* x0, const double* q0,
                   const double* mean, const double* std, const double* noise, double* actions, double* traj,
                   double* sigma, double* obj_cost, double* con_cost, int32_t* status, void* workspace,
                   int64_t workspace_bytes, void* stream) {
    if (!model || !env || !x0 || !actions || !obj_cost || !con_cost || !status) return SX_ERR_ARG;
    if (E <= 0 || P <= 0 || H <= 0) return SX_ERR_ARG;
    if (noise && (!mean || !std)) return SX_ERR_ARG;
    if (model->n_s != env->n_s || model->n_u != env->n_u) return SX_ERR_ARG;
    if (env->m <= 0 || env->m > SX_MAX_M) return SX_ERR_UNSUPPORTED;
    sx::RolloutPtrs rp{x0, q0, mean, std, noise, actions, traj, sigma, obj_cost, con_cost, status, E, P, H};
#define CALL(NS, NU) \
    sx::launch_rollout<NS, NU>(model, env, rp, (double*)workspace, workspace_bytes, (hipStream_t)stream)
    SX_DISPATCH(model->n_s, model->n_u, CALL);
#undef CALL
}

int sx_cem_rollout_elites(const sx_gp_model* model, const sx_env* env, int E, int P, int H, const double* x0, const double* q0,
                          const double* elite_rows, int k, const double* noise, double* actions, double* traj, double* sigma,
                          double* obj_cost, double* con_cost, int32_t* status, double* mean_out, double* std_out, void* stream) {
    if (!model || !env || !x0 || !actions || !obj_cost || !con_cost || !status || !elite_rows || !noise) return SX_ERR_ARG;
    if (E <= 0 || P <= 0 || H <= 0 || k <= 0 || (mean_out == nullptr) != (std_out == nullptr)) return SX_ERR_ARG;
    if (model->n_s != env->n_s || model->n_u != env->n_u) return SX_ERR_ARG;
    if (env->m <= 0 || env->m > SX_MAX_M) return SX_ERR_UNSUPPORTED;
    sx::RolloutPtrs rp{x0, q0, nullptr, nullptr, noise, actions, traj, sigma, obj_cost, con_cost, status, E, P, H};
    rp.elite_rows = elite_rows;
    rp.elite_k = k;
    rp.mean_out = mean_out;
    rp.std_out = std_out;
#define CALL(NS, NU) sx::launch_rollout<NS, NU>(model, env, rp, nullptr, 0, (hipStream_t)stream)
    SX_DISPATCH(model->n_s, model->n_u, CALL);
#undef CALL
}

static bool feat_model_ok(const sx_feat_model* m) {
    if (!m || m->n_s <= 0 || m->n_s > SX_MAX_NS || m->n_u <= 0 || m->n_u > SX_MAX_NU) return false;
    if (m->n_layers < 0 || m->n_layers > SX_FEAT_MAX_LAYERS || m->n_feat <= 0 || m->n_feat > SX_FEAT_MAX_WIDTH) return false;
    if (m->width[0] != m->n_s + m->n_u) return false;
    for (int l = 1; l <= m->n_layers; ++l)
        if (m->width[l] <= 0 || m->width[l] > SX_FEAT_MAX_WIDTH) return false;
    if (m->n_layers == 0 ? m->n_feat != m->n_s + m->n_u : (m->n_feat != m->width[m->n_layers] || !m->net)) return false;
    return true;
}

int sx_feat_features(const sx_feat_model* model, const double* x, int N, double* phi, void* stream) {
    if (!feat_model_ok(model) || N < 0) return SX_ERR_ARG;
    if (N == 0) return SX_OK;
    if (!x || !phi) return SX_ERR_ARG;
    const sx::FeatConst fc = sx::make_feat_const(model);
    const size_t lds = sx::kFeatLdsDoubles * sizeof(double);
    const dim3 grid((N + sx::kFeatWave - 1) / sx::kFeatWave);
#define FEAT_D(DD)                                                                                                          \
    if (fc.d_in == DD) {                                                                                                   \
        if (int rc = sx::allow_lds(sx::feat_features_kernel<DD>, lds)) return rc;                                          \
        hipLaunchKernelGGL(sx::feat_features_kernel<DD>, grid, dim3(sx::kFeatWave), lds, (hipStream_t)stream, fc, x, N, phi); \
        return sx::check_launch();                                                                                         \
    }
    FEAT_D(2) FEAT_D(3) FEAT_D(4) FEAT_D(5) FEAT_D(6)
#undef FEAT_D
    return SX_ERR_UNSUPPORTED;
}

int sx_feat_fit(const sx_feat_model* model, const double* phi, const double* y, int N, const double* lambda, double* wbar,
                double* minv, double* stats, int32_t* status, void* stream) {
    if (!feat_model_ok(model) || !phi || !y || N <= 0 || !lambda || !wbar || !minv || !stats || !status) return SX_ERR_ARG;
    sx::FeatFitArgs fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.phi = phi;
    fa.y = y;
    for (int d = 0; d < model->n_s; ++d) fa.lambda[d] = lambda[d];
    fa.wbar = wbar;
    fa.minv = minv;
    fa.stats = stats;
    fa.status = status;
    fa.n = N;
    fa.F = model->n_feat;
    fa.n_s = model->n_s;
    hipLaunchKernelGGL(sx::feat_fit_kernel, dim3(model->n_s), dim3(1024), 0, (hipStream_t)stream, fa);
    return sx::check_launch();
}

int sx_feat_predict(const sx_feat_model* model, const double* z, int P, double* mean, double* var, double* jac, void* stream) {
    if (!feat_model_ok(model) || P < 0) return SX_ERR_ARG;
    if (P == 0) return SX_OK;
    if (!z || !mean || !var || !model->wbar || !model->minv) return SX_ERR_ARG;
#define CALL(NS, NU) sx::launch_feat_predict<NS, NU>(model, z, P, mean, var, jac, (hipStream_t)stream)
    SX_DISPATCH(model->n_s, model->n_u, CALL);
#undef CALL
}

int sx_cem_rollout_feat(const sx_feat_model* model, const sx_env* env, int E, int P, int H, const double* x0, const double* q0,
                        const double* mean, const double* std, const double* noise, double* actions, double* traj,
                        double* sigma, double* obj_cost, double* con_cost, int32_t* status, void* stream) {
    if (!feat_model_ok(model) || !env || !x0 || !actions || !obj_cost || !con_cost || !status) return SX_ERR_ARG;
    if (!model->wbar || !model->minv || E <= 0 || P <= 0 || H <= 0) return SX_ERR_ARG;
    if (noise && (!mean || !std)) return SX_ERR_ARG;
    if (model->n_s != env->n_s || model->n_u != env->n_u) return SX_ERR_ARG;
    if (env->m <= 0 || env->m > SX_MAX_M) return SX_ERR_UNSUPPORTED;
    sx::FeatRolloutPtrs rp{x0, q0, mean, std, noise, actions, traj, sigma, obj_cost, con_cost, status, E, P, H};
#define CALL(NS, NU) sx::launch_rollout_feat<NS, NU>(model, env, rp, (hipStream_t)stream)
    SX_DISPATCH(model->n_s, model->n_u, CALL);
#undef CALL
}

static bool mlp_model_ok(const sx_mlp_model* m) {
    if (!m || m->n_s <= 0 || m->n_s > SX_MAX_NS || m->n_u <= 0 || m->n_u > SX_MAX_NU) return false;
    if (m->n_hidden < 0 || m->n_hidden > SX_MLP_MAX_HIDDEN || m->n_out < m->n_s || m->n_samples <= 0) return false;
    if (m->predict_std && m->n_out < 2 * m->n_s) return false;
    if (m->width[0] != m->n_s + m->n_u || !m->net || !m->masks) return false;
    for (int l = 1; l <= m->n_hidden; ++l)
        if (m->width[l] <= 0 || m->width[l] > SX_MLP_MAX_WIDTH) return false;
    return true;
}

int sx_mlp_predict(const sx_mlp_model* model, const double* z, int P, double* mean, double* var, double* jac, void* stream) {
    if (!mlp_model_ok(model) || P < 0) return SX_ERR_ARG;
    if (P == 0) return SX_OK;
    if (!z || !mean || !var) return SX_ERR_ARG;
#define CALL(NS, NU) sx::launch_mlp_predict<NS, NU>(model, z, P, mean, var, jac, (hipStream_t)stream)
    SX_DISPATCH(model->n_s, model->n_u, CALL);
#undef CALL
}

int sx_cem_rollout_mlp(const sx_mlp_model* model, const sx_env* env, int E, int P, int H, const double* x0, const double* q0,
                       const double* mean, const double* std, const double* noise, double* actions, double* traj,
                       double* sigma, double* obj_cost, double* con_cost, int32_t* status, void* stream) {
    if (!mlp_model_ok(model) || !env || !x0 || !actions || !obj_cost || !con_cost || !status) return SX_ERR_ARG;
    if (E <= 0 || P <= 0 || H <= 0) return SX_ERR_ARG;
    if (noise && (!mean || !std)) return SX_ERR_ARG;
    if (model->n_s != env->n_s || model->n_u != env->n_u) return SX_ERR_ARG;
    if (env->m <= 0 || env->m > SX_MAX_M) return SX_ERR_UNSUPPORTED;
    sx::FeatRolloutPtrs rp{x0, q0, mean, std, noise, actions, traj, sigma, obj_cost, con_cost, status, E, P, H};
#define CALL(NS, NU) sx::launch_rollout_mlp<NS, NU>(model, env, rp, (hipStream_t)stream)
    SX_DISPATCH(model->n_s, model->n_u, CALL);
#undef CALL
}

namespace sx {
// sx_cem_pack_result: one small workgroup
__global__ __launch_bounds__(256) void pack_result_kernel(int G, int E, int L, const int* __restrict__ status,
                                                          const int* __restrict__ best_ok, const double* __restrict__ q_block,
                                                          long long q_count, const double* __restrict__ best,
                                                          double* __restrict__ out) {
    __shared__ int any_nz;
    const int tid = threadIdx.x;
    if (tid == 0) any_nz = 0;
    __syncthreads();
    int nz = 0;
    if (q_block)
        for (long long i = tid; i < q_count; i += blockDim.x) nz |= (q_block[i] != 0.0) ? 1 : 0;   // (NaN counts as non-zero)
    if (nz) atomicOr(&any_nz, 1);
    for (int i = tid; i < G; i += blockDim.x) out[i] = (double)status[i];
    for (int i = tid; i < E; i += blockDim.x) out[G + i] = (double)best_ok[i];
    for (int i = tid; i < E * L; i += blockDim.x) out[G + E + 1 + i] = best[i];
    __syncthreads();
    if (tid == 0) out[G + E] = any_nz ? 1.0 : 0.0;
}
}  // namespace sx

int sx_cem_pack_result(int G, int E, int row_len, const int32_t* status, const int32_t* best_ok, const double* q_block,
                       int64_t q_count, const double* best, double* out, void* stream) {
    if (G <= 0 || E <= 0 || row_len <= 0 || !status || !best_ok || !best || !out || q_count < 0) return SX_ERR_ARG;
    hipLaunchKernelGGL(sx::pack_result_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, G, E, row_len, status, best_ok,
                       q_count > 0 ? q_block : nullptr, (long long)q_count, best, out);
    return sx::check_launch();
}

// Which ranking kernel: by shape only (every rank of a multi-GPU solve must take the same one: the elite order differs).
// Counting spreads one or two problems over the chip (E P / 16 workgroups, all keys in each one's LDS); many problems at
// once already fill it with the one-workgroup kernel (SX_RANK_PATH = count | select overrides, for A/B measurements).
int sx_cem_rank_counts(int E, int P) {
    if (E <= 0 || P <= 0 || P > sx::kCountMaxP || E > sx::kCountMaxE) return 0;
    static const char* const forced = std::getenv("SX_RANK_PATH");
    if (forced && forced[0] == 's') return 0;
    if (forced && forced[0] == 'c') return 1;
    return (long long)E * ((P + 15) / 16) <= sx::kCountMaxGrid ? 1 : 0;
}

int sx_cem_rank_refit(int E, int P, int k, int row_len, const double* con_cost, const double* obj_cost,
                      int64_t cost_stride, const double* actions, int64_t act_stride, int32_t* elite_idx,
                      double* elite_rows, double* mean, double* std, double* best, int32_t* best_ok, void* stream) {
    if (!con_cost || !obj_cost || !actions || E <= 0 || P <= 0 || k <= 0 || row_len <= 0) return SX_ERR_ARG;
    if (k > P) return SX_ERR_ARG;
    if (k > sx::kRankMaxK) return SX_ERR_UNSUPPORTED;
    sx::RankArgs ra{P,      k,          row_len,    con_cost, obj_cost, (long long)cost_stride,
                    actions, (long long)act_stride, elite_idx, elite_rows, mean,     std,
                    best,   best_ok};
    const bool count = sx_cem_rank_counts(E, P) != 0 && (elite_rows || !mean);
    const int tiles = (P + 15) / 16;
    if (count) {
        const size_t lds = (size_t)((P + 127) & ~127) * sizeof(sx::CountKey);
        if (int rc = sx::allow_lds(sx::cem_rank_count_kernel, lds)) return rc;
        // the refit's ticket cells: one set per launch, kCountTicketSlots sets in rotation (launches whose refits overlap in
        // time must be fewer than that); the symbol's address is looked up once per device
        static std::atomic<unsigned int> seq{0};
        unsigned int* tickets = nullptr;
        if (mean) {
            static std::mutex mu;
            static std::map<int, unsigned int*> base;
            int dev = 0;
            (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> lock(mu);
            auto it = base.find(dev);
            if (it == base.end()) {
                unsigned int* p = nullptr;
                if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(sx::g_rank_tickets)) != hipSuccess) return SX_ERR_LAUNCH;
                it = base.emplace(dev, p).first;
            }
            tickets = it->second + (size_t)(seq.fetch_add(1) % sx::kCountTicketSlots) * sx::kCountMaxE;
        }
        sx::launch(SX_PROF_RANK, sx::cem_rank_count_kernel, dim3((unsigned)tiles, (unsigned)E), dim3(sx::kCountThreads), lds,
                   (hipStream_t)stream, ra, tickets);
        return sx::check_launch();
    }
    if (P > sx::kRankThreads * sx::kRankSlots) return SX_ERR_UNSUPPORTED;
    const int slots = (P + sx::kRankThreads - 1) / sx::kRankThreads;
    if (slots <= 4)
        sx::launch(SX_PROF_RANK, sx::cem_rank_kernel<4>, dim3(E), dim3(sx::kRankThreads), 0, (hipStream_t)stream, ra);
    else if (slots <= 8)
        sx::launch(SX_PROF_RANK, sx::cem_rank_kernel<8>, dim3(E), dim3(sx::kRankThreads), 0, (hipStream_t)stream, ra);
    else
        sx::launch(SX_PROF_RANK, sx::cem_rank_kernel<sx::kRankSlots>, dim3(E), dim3(sx::kRankThreads), 0, (hipStream_t)stream, ra);
    return sx::check_launch();
}

}  // extern "C"
