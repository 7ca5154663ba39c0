// The fused CEM particle rollout kernel (sx_cem_rollout, training sets that fit the LDS budget).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sx_amd.h"
#include "sx_gp.hpp"
#include "sx_reach.hpp"
#include "sx_refit.hpp"

namespace sx {

#ifndef SX_ROLLOUT_THREADS
#define SX_ROLLOUT_THREADS (64 * SX_WAVES)
#endif
constexpr int kRolloutThreads = SX_ROLLOUT_THREADS;  // waves of the CU that owns the 16-particle tile

// ---------------------------------------------------------------------------------------------------------------
// sx_cem_rollout: the fused H-step particle rollout.  One workgroup = 16 particles of one problem for all H steps.
// ---------------------------------------------------------------------------------------------------------------
#ifdef SX_STAMPS
// Diagnostic build only (tools/phase_stamps.py): per-workgroup cycle sums of the three phases of a step.
static __device__ unsigned long long* g_stamp_buf = nullptr;   // (one per translation unit)
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#endif

struct RolloutPtrs {
    const double* x0;
    const double* q0;
    const double* mean;
    const double* std;
    const double* noise;
    double* actions;
    double* traj;
    double* sigma;
    double* obj_cost;
    double* con_cost;
    int* status;
    int E, P, H;
    // sx_cem_rollout_elites: the sampling distribution is refit from the previous iteration's elite rows
    // [E x elite_k x (2 + H n_u)] in the kernel's prologue (mean / std above are then unused)
    const double* elite_rows = nullptr;
    int elite_k = 0;
    double* mean_out = nullptr;   // [E x H n_u] the refit, written by the first workgroup of each problem (may be null)
    double* std_out = nullptr;
#ifdef SX_STAMPS
    unsigned long long* stamps = nullptr;   // the register-resident kernels take the stamp buffer as an argument
#endif
};

// BYOUT = false: Kstar of all outputs in LDS at once -- two barriers per step (the kernel measured throughout DESIGN.md).
// BYOUT = true: for training sets whose n_s Kstar buffers do not fit in LDS together, one output at a time
// (Kstar_d | MFMA_d for d = 0 .. n_s - 1: 2 n_s barriers per step, per-output stage streams), still ONE launch for the
// whole rollout and no Kstar in HBM.
template <int NS, int NU, bool BYOUT = false>
__global__ __launch_bounds__(kRolloutThreads) void cem_rollout_kernel(GpConst<NS, NS + NU> gc,
                                                                      const int4* __restrict__ stage_tab,
                                                                      ReachConst<NS, NU> rc,
                                                                      CostConst<SX_MAX_M, NS, NU> cc, RolloutPtrs rp) {
    constexpr int D = NS + NU;
    constexpr int S = NS + NS * NS;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    GpTileLds<NS, D> lds;
    const int nw = blockDim.x >> 6;
    double* acts = lds.carve(smem, gc.n_train, gc.n_pad, nw, BYOUT ? 1 : NS);  // [16][H][NU]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int H = rp.H;
    const int tiles_per_problem = (rp.P + SX_TILE - 1) / SX_TILE;
    const int e = blockIdx.x / tiles_per_problem;
    const int c0 = (blockIdx.x - e * tiles_per_problem) * SX_TILE;  // first particle of the tile within problem e

    // The head of this wave's MFMA stream (stage count, first descriptors, first W fragments: two dependent round trips to
    // L2) is requested first of all, so that it travels while X, the exp table and the actions are loaded: the launch's
    // fixed cost is ~8 us of its 127 (tools: bench.py --horizon 1 .. 15), all of it dependent loads like these.
    // (in the output-by-output mode every phase fetches the head of its own stream)
    const MfmaHead head = gp_mfma_head(gc, stage_tab, wave, nw, lane, gc.stage_cap);
    const int4* __restrict__ const tab_one = stage_tab + (size_t)nw * (1 + gc.stage_cap);
    gp_load_xs(gc, lds);
    // The sampling distribution.  Either given (mean, std), or refit here from the elite rows the ranking kernel left behind
    // (sx_cem_rollout_elites): every workgroup computes the same H n_u means and standard deviations for itself, a wave per
    // group of columns and no barrier inside (wave_refit_columns), while the loads above are still travelling -- the refit used to be
    // the serial tail of the ranking kernel, ~5 us per CEM iteration on one compute unit.  The result sits in the (still
    // unused) Kstar buffer until the actions are sampled.
    const double* dist_mean = rp.mean + (int64_t)e * H * NU;
    const double* dist_std = rp.std + (int64_t)e * H * NU;
    // (the first trip's noise draw / given action is requested before the refit, so that it travels meanwhile)
    double first_in = 0.0;
    if (tid < SX_TILE * H * NU && c0 + tid / (H * NU) < rp.P) {
        const int64_t gi = ((int64_t)e * rp.P + c0) * (H * NU) + tid;
        first_in = rp.noise ? rp.noise[gi] : rp.actions[gi];
    }
    if (rp.elite_rows) {
        const int L = H * NU, W = 2 + L;
        double* const ms = lds.kfrag;   // [2][L]
        const double* rows = rp.elite_rows + (int64_t)e * rp.elite_k * W + 2;
        const bool publish = (blockIdx.x - e * tiles_per_problem) == 0 && rp.mean_out;
        // 2^cshift adjacent columns per wave and trip: all columns in one trip when they fit (L <= 64 nw).
        // (This code runs once per launch, but its registers are part of the whole kernel's allocation problem, and the step
        // loop's spills depend on its spelling.  A/B on one box, config 2 / config 5 launch: 8 loads in flight per lane
        // 129.9 / 974 us, 16 with the uniform guards 135.6 / 1030, 16 without 130.7 / 990, as a non-inlined function
        // 147.7 / 1116 -- config 5 does not even execute it.)
        int cshift = 0;
        while ((nw << cshift) < L && cshift < 6) ++cshift;
        const int cc = lane & ((1 << cshift) - 1);
        for (int c0 = wave << cshift; c0 < L; c0 += nw << cshift) {
            const int col = c0 + cc;
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
    // sample (or load) this tile's action sequences: a = mean + std * eps
    for (int i = tid; i < SX_TILE * H * NU; i += blockDim.x) {
        const int c = i / (H * NU);
        const int r = i - c * (H * NU);
        double a = 0.0;
        if (c0 + c < rp.P) {
            const int64_t gi = ((int64_t)e * rp.P + c0 + c) * (H * NU) + r;
            if (rp.noise) {
                a = dist_mean[r] + dist_std[r] * (i == tid ? first_in : rp.noise[gi]);
                rp.actions[gi] = a;
            } else {
                a = (i == tid) ? first_in : rp.actions[gi];
            }
        }
        acts[i] = a;
    }
    // per-particle state lives in the registers of thread c (tid < 16) for the whole rollout
    const bool owner = tid < SX_TILE;
    const bool valid = owner && (c0 + tid < rp.P);
    double p[NS], Q[NS][NS];
    bool have_q = rp.q0 != nullptr;
    double obj = 0.0, con = 0.0;
    int st = 0;
    if (owner) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            p[i] = rp.x0[(int64_t)e * NS + i];
#pragma unroll
            for (int j = 0; j < NS; ++j) Q[i][j] = have_q ? rp.q0[((int64_t)e * NS + i) * NS + j] : 0.0;
        }
    }
    __syncthreads();
    if (owner) {
#pragma unroll
        for (int i = 0; i < NS; ++i) lds.zs[tid * D + i] = p[i];
#pragma unroll
        for (int cidx = 0; cidx < NU; ++cidx) lds.zs[tid * D + NS + cidx] = acts[(tid * H + 0) * NU + cidx];
    }
    __syncthreads();

    // Step t:   Kstar(t)  |sync|  MFMA(t)  |sync|
    // The next centre p_{t+1} = mean + a p + b u is not a phase of its own: every Kstar thread derives the centre of
    // ITS query point at the start of Kstar(t+1) from z_t (LDS) and the posterior mean MFMA(t) left in LDS -- a dozen
    // FMAs, redundantly, in the slack the Kstar waves have against finish().  z lives in two LDS buffers: buffer t & 1
    // holds z_t = (p_t, u_t); finish(t-1), which derives the same p_t (same fma chain, bit-identical) during
    // Kstar(t), writes z_t into it for the threads of step t + 1.
    // The rest of step t (variance, Jacobian, ellipsoid algebra, costs: ~4.2k cycles on 16 lanes) does not feed
    // Kstar(t+1), so wave 0 runs it DURING Kstar(t+1) while waves 1..7 compute the kernel rows.
    double* const zs_base = lds.zs;
    // centre of particle c at step t >= 1 from z_{t-1} and the means of step t - 1
    auto next_centre = [&](int c, const double* z_prev, double (&out)[NS]) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            double s = lds.mj[i * 256 + c];  // posterior mean of output i
#pragma unroll
            for (int j = 0; j < NS; ++j) s = fma(rc.a[i * NS + j], z_prev[j], s);
#pragma unroll
            for (int cidx = 0; cidx < NU; ++cidx) s = fma(rc.b[i * NU + cidx], z_prev[NS + cidx], s);
            out[i] = s;
        }
    };
    auto finish = [&](int t) {
        double z[D], u[NU], mean[NS], var[NS], jac[NS][D], p1[NS], Q1[NS][NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) z[j] = p[j];
#pragma unroll
        for (int cidx = 0; cidx < NU; ++cidx) {
            u[cidx] = acts[(tid * H + t) * NU + cidx];
            z[NS + cidx] = u[cidx];
        }
        int st_step = 0;
        if (have_q) {
            gp_collect<NS, D, true>(gc, lds, nw, tid, z, mean, var, jac);
            reach_ellipsoid<NS, NU>(rc, p, Q, u, mean, var, jac, p1, Q1, st_step);
        } else {
            gp_collect<NS, D, false>(gc, lds, nw, tid, z, mean, var, jac);
            reach_point<NS, NU>(rc, p, u, mean, var, p1, Q1, st_step);
        }
        have_q = true;
        {
            // exactly the centre the next GP query uses (same chain as next_centre), published for the step after it
            double zt[D];
#pragma unroll
            for (int j = 0; j < NS; ++j) zt[j] = p[j];
#pragma unroll
            for (int cidx = 0; cidx < NU; ++cidx) zt[NS + cidx] = u[cidx];
            next_centre(tid, zt, p1);
            if (t + 1 < H) {
                double* zn = zs_base + ((t + 1) & 1) * 16 * D + tid * D;
#pragma unroll
                for (int i = 0; i < NS; ++i) zn[i] = p1[i];
#pragma unroll
                for (int cidx = 0; cidx < NU; ++cidx) zn[NS + cidx] = acts[(tid * H + t + 1) * NU + cidx];
            }
        }
        if (valid) st |= st_step;
        // costs (safempc_cem.py:102-132,304-312; action constraint: test_safempc_cem.py:59-71)
        obj += objective_cost<SX_MAX_M, NS, NU>(cc, p1, var);
        bool uviol = false;
#pragma unroll
        for (int cidx = 0; cidx < NU; ++cidx) uviol = uviol || (u[cidx] < cc.u_min[cidx]) || (u[cidx] > cc.u_max[cidx]);
        if (uviol) con += SX_ACTION_VIOLATION_COST;
        if (cc.con_mode == SX_CON_ALL_STATES || t == H - 1) {
            if (polytope_violated<SX_MAX_M, NS>(cc.h_mat, cc.h_vec, cc.m, 1.0, p1, Q1, nullptr))
                con += SX_STATE_VIOLATION_COST;
        }
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
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            p[i] = p1[i];
#pragma unroll
            for (int j = 0; j < NS; ++j) Q[i][j] = Q1[i][j];
        }
    };

#ifdef SX_STAMPS
    unsigned long long c_k = 0, c_kb = 0, c_m = 0, c_mb = 0, c_e = 0, c_eb = 0;
    // shader clock against the constant 100 MHz reference: is the chip holding its clock under this kernel?
    unsigned long long rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
    const unsigned long long ct0 = stamp();
#endif
    // Kstar shares (pairs of fragments).  Step 0: all waves alike.  From step 1 on wave 0 runs finish(); waves w and w + 4
    // share a SIMD, so wave 4 competes with finish() for its pipe and gets half a share (weight 1 against 2).
    int q0_begin, q0_end, q_begin = 0, q_end = 0;
    kstar_pair_range(gc.n_pad >> 3, wave, 1, nw, q0_begin, q0_end);
    if (wave > 0) {
        const bool shares = nw > 4;
        const int before = 2 * (wave - 1) - ((shares && wave > 4) ? 1 : 0);
        const int weight = (shares && wave == 4) ? 1 : 2;
        kstar_pair_range(gc.n_pad >> 3, before, weight, 2 * (nw - 1) - (shares ? 1 : 0), q_begin, q_end);
    }
    for (int t = 0; t < H; ++t) {
#ifdef SX_STAMPS
        const unsigned long long t0 = stamp();
#endif
        // the query point of this thread's particle (every thread of a Kstar wave; see next_centre above)
        double zq[D];
        const bool kstar_wave = !(t > 0 && wave == 0);
        if (kstar_wave || BYOUT) {   // (output by output, wave 0 rejoins the Kstar waves after finish())
            const int c = lane & 15;
            if (t == 0) {
#pragma unroll
                for (int j = 0; j < D; ++j) zq[j] = zs_base[c * D + j];
            } else {
                double pc[NS];
                next_centre(c, zs_base + ((t - 1) & 1) * 16 * D + c * D, pc);
#pragma unroll
                for (int i = 0; i < NS; ++i) zq[i] = pc[i];
#pragma unroll
                for (int cidx = 0; cidx < NU; ++cidx) zq[NS + cidx] = acts[(c * H + t) * NU + cidx];
            }
        }
        if constexpr (BYOUT) {
            // output by output; finish(t-1) rides on the first Kstar phase.  (z was derived above, before MFMA_0
            // overwrites the means of the previous step.)
            auto one_output = [&](auto dtag) {
                constexpr int DD = decltype(dtag)::value;
                if constexpr (DD < NS) {
                    const int4* __restrict__ tab_d = tab_one + (size_t)DD * nw * (1 + gc.stage_cap_one);
                    const MfmaHead head_d = gp_mfma_head(gc, tab_d, wave, nw, lane, gc.stage_cap_one);
                    if (DD == 0 && t > 0) {
                        if (wave == 0) {
                            if (owner) finish(t - 1);
                        } else {
                            gp_kstar_phase_one<NS, D, DD>(gc, lds, q_begin, q_end, zq);
                        }
                    } else {
                        gp_kstar_phase_one<NS, D, DD>(gc, lds, q0_begin, q0_end, zq);   // all waves, equal shares
                    }
                    __syncthreads();
                    gp_mfma_phase<NS, D, 1>(gc, tab_d, lds, wave, nw, lane, head_d, gc.stage_cap_one, DD);
                    __syncthreads();
                }
            };
            one_output(std::integral_constant<int, 0>{});
            one_output(std::integral_constant<int, 1>{});
            one_output(std::integral_constant<int, 2>{});
            one_output(std::integral_constant<int, 3>{});
            static_assert(NS <= 4, "one_output is spelled out for up to four outputs");
            continue;
        }
        if (!kstar_wave) {
            if (owner) finish(t - 1);
        } else if (t == 0) {
            gp_kstar_phase(gc, lds, q0_begin, q0_end, zq);
        } else {
            gp_kstar_phase(gc, lds, q_begin, q_end, zq);
        }
#ifdef SX_STAMPS
        const unsigned long long t1 = stamp();
#endif
        __syncthreads();
#ifdef SX_STAMPS
        const unsigned long long t2 = stamp();
#endif
        gp_mfma_phase(gc, stage_tab, lds, wave, nw, lane, head, gc.stage_cap);
#ifdef SX_STAMPS
        const unsigned long long t3 = stamp();
#endif
        __syncthreads();
#ifdef SX_STAMPS
        const unsigned long long t4 = stamp();
        c_k += t1 - t0; c_kb += t2 - t1; c_m += t3 - t2; c_mb += t4 - t3;
#endif
    }
    if (owner) finish(H - 1);
#ifdef SX_STAMPS
    if (g_stamp_buf && lane == 0) {
        unsigned long long* o = g_stamp_buf + ((size_t)blockIdx.x * nw + wave) * 8;
        o[0] = c_k; o[1] = c_kb; o[2] = c_m; o[3] = c_mb; o[4] = c_e; o[5] = c_eb;
        unsigned long long rt1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
        o[6] = stamp() - ct0;   // shader cycles of the step loop
        o[7] = rt1 - rt0;       // the same span in 10 ns ticks
    }
#endif
    if (valid) {
        const int64_t g = (int64_t)e * rp.P + c0 + tid;
        rp.obj_cost[g] = obj;
        rp.con_cost[g] = con;
        if (st) atomicOr(rp.status, st);
    }
}

}  // namespace sx
