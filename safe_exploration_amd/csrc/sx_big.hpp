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
// grid (P128 / 16, chunks of 32 fragment pairs = 256 k); thread = (query point c = tid & 15, kk = (tid >> 4) & 3, pair
// q = 32 chunk + 4 i + (tid >> 6)).  A thread computes BOTH slots of one 16-byte fragment element (rows 8 q + kk and
// 8 q + kk + 4), so a wave stores 64 consecutive elements -- 1 KB, whole lines -- per output and trip (round 1 wrote the two
// slots of an element from different trips: half-filled 16-byte elements, 4.0 TB/s; the kernel is HBM-write bound).
template <int NS, int D>
__global__ __launch_bounds__(256) void kstar_big_kernel(GpConst<NS, D> gc, BigWs ws) {
    const int tile = blockIdx.x, c = threadIdx.x & 15, kk = (threadIdx.x >> 4) & 3, qw = threadIdx.x >> 6;
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
    const int npairs = gc.n_pad >> 3;
    const int64_t tstride = (int64_t)npairs * 64;                 // 16-byte elements per particle tile
    const int64_t dstride = (int64_t)gridDim.x * tstride;
    v2d* out = reinterpret_cast<v2d*>(ws.ks) + (int64_t)tile * tstride + (kk << 4) + c;
    for (int i = 0; i < 8; ++i) {
        const int q = blockIdx.y * 32 + 4 * i + qw;
        if (q >= npairs) break;                                   // (wave-uniform)
        double arg[2 * NS], val[2 * NS];
        int ks[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 8 * q + kk + 4 * h;
            ks[h] = k;
            const int kr = k < gc.n_train ? k : 0;
            double sq[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double df = z[j] - gc.x_train[(int64_t)kr * D + j];
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
        for (int d = 0; d < NS; ++d)
            out[d * dstride + (int64_t)q * 64] = v2d{ks[0] < gc.n_train ? val[d] : 0.0, ks[1] < gc.n_train ? val[NS + d] : 0.0};
    }
}

// ---- 2. triangular product + row reduction ---------------------------------------------------------------------------
// T = W_d . Kstar_d^T, 128 rows x 128 particles per workgroup; 4 waves, wave w owns row-blocks 4 (w >> 1) .. +3 and particle
// tiles 4 (w & 1) .. +3 (4 x 4 accumulators of 16 x 16).
//
// The f64 MFMA runs on the SIMD's f64 FMA lanes: every VALU instruction takes cycles from it (tools/overlap_probe.hip).
// Round 1's kernel spent 1.8 VALU instructions per MFMA on 64-bit address arithmetic and bounds selects, staged both
// operands through registers and needed two barriers per 64 MFMAs (profiles/r02_pmc_cfg4.json: matrix pipe 75 % busy).
// Here a K-chunk (2 fragment pairs = 16 k: 16 KB of W, 16 KB of Kstar) goes global -> LDS by 32 `buffer_load ... lds`
// pieces of 1 KB -- one fragment each, the fragment order of both operands IS the lane-linear image the DMA writes --,
// 8 per wave, addressed by SGPR offsets only (a constant per piece + 1 KB per pair); two LDS buffers, ONE barrier per chunk,
// the DMA of chunk c + 1 in flight under the 64 MFMAs per wave of chunk c.  W fragments beyond a row-block's diagonal are
// whatever follows in memory: the MFMAs that would read them are skipped (wave-uniformly); row-blocks beyond the matrix
// read zeros (buffer bounds).
//
// Grid: 1-D, LONGEST TILE FIRST.  W is triangular: row tile rt has (rt + 1) / row_tiles of the last tile's K extent, a
// 16-fold spread at N = 2000.  Dealt out in ascending order the long tiles start last and the launch ends in a tail of a
// few busy CUs (7.2 ms at config 4); in descending order the short tiles fill the gaps the long ones leave (3.9 ms).  With
// every workgroup reading ONE W tile and ONE Kstar tile (no fabric traffic at all) the times are the same: the kernel is
// bound by MFMA issue and by how evenly the tiles pack, not by HBM (8 - 11 GB per launch at ~2 TB/s).
// PPC = fragment pairs per K-chunk (1 or 2: 8 or 16 k), NBUF = LDS buffers (NBUF - 1 chunks of DMA in flight).
// PT = particle tiles (of 16) per workgroup: 8 (128 rows x 128 particles), or 4 for small grids -- twice the workgroups, so that
// a grid of one round of 128 x 128 tiles (N ~ 1000 at 4096 particles: ONE workgroup of 4 waves per compute unit, every
// barrier and DMA wait idling its SIMDs) becomes two workgroups per compute unit that fill each other's gaps.
template <int PPC, int NBUF, int PT = kBigRb>
constexpr int big_lds_bytes() { return NBUF * (kBigRb + PT) * PPC * 64 * 16; }   // [buffer][W blocks | Kstar tiles][pair][lane] x 16 B

template <int NS, int D, int PPC, int NBUF, int PT>
__global__ __launch_bounds__(kBigThreads, (PPC * NBUF <= 3) ? 3 : (PPC * NBUF <= 4 ? 2 : 1)) void trmm_reduce_kernel(GpConst<NS, D> gc, BigWs ws, int64_t p128,
                                                                      int row_tiles, int xcd_aware) {
    static_assert(PT == 8 || PT == 4, "a wave column holds PT / 2 particle tiles");
    constexpr int NT = PT / 2;                                // particle tiles per wave column
    extern __shared__ __attribute__((aligned(16))) double big_smem[];
    typedef __attribute__((address_space(3))) v2d lds_v2d;
    lds_v2d* const lds = (lds_v2d*)big_smem;                 // [NBUF][8 W blocks + PT Kstar tiles][PPC][64]
    constexpr int kFragsW = kBigRb * PPC, kFragsK = PT * PPC;   // fragments of the two operands per chunk
    constexpr int kFragsBuf = kFragsW + kFragsK;
    constexpr int kPiecesW = kFragsW / 2, kPiecesK = kFragsK / 2;   // DMA pieces per wave and chunk (waves 0, 1: W; 2, 3: Kstar)
    constexpr int kPieces = kPiecesW;                         // (the larger of the two: array bound)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int pgroups = (int)(p128 / (PT * 16));
    // XCD-aware decode (see above); any grid whose size is not a multiple of 8 keeps the plain order
    int id = blockIdx.x;
    const int total = gridDim.x;
    int rt, pg, d, rt_second = -1;
    if (xcd_aware & 16) {
        // PAIRED tiles (small grids): this workgroup does row tile row_tiles - 1 - j and then row tile j -- every workgroup
        // the same work, so a launch of one or two rounds has no tail however the dispatcher places it (longest-first
        // cannot pack 512 workgroups of 8 : 1 unequal work onto 768 slots: the N ~ 1000 step of tools/n_sweep.sh)
        const int half = (row_tiles + 1) >> 1;
        pg = id % pgroups;
        d = (id / pgroups) % NS;
        const int j = id / (pgroups * NS);
        rt = row_tiles - 1 - j;
        if (j != rt) rt_second = j;
        (void)half;
    } else if (xcd_aware & 8) {
        // longest first: row tile rt does (rt + 1) / row_tiles of the longest tile's work, so the tiles are dealt out in
        // descending rt (particle group fastest: the workgroups that run together share their W row tile through L2)
        pg = id % pgroups;
        d = (id / pgroups) % NS;
        rt = row_tiles - 1 - id / (pgroups * NS);
    } else {
        if ((xcd_aware & 1) && (total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
        rt = id % row_tiles;
        pg = (id / row_tiles) % pgroups;
        d = id / (row_tiles * pgroups);
    }
    const int pg_src = (xcd_aware & 2) ? 0 : pg;     // (diagnostic, timing only: every workgroup reads particle group 0's Kstar)
    const int nrb = gc.n_pad >> 4;
    const int64_t wpo = w_pairs_per_output(nrb);
    const int64_t tstride = (int64_t)(gc.n_pad >> 3) * 64;           // 16-byte elements per particle tile
    // buffer descriptors: W of output d (reads beyond it return 0), the 8 Kstar particle tiles of this workgroup
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(gc.a_pack) + (int64_t)d * wpo * 128, 0, (int)(wpo * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc(
        ws.ks + (((int64_t)d * (p128 / 16) + (int64_t)pg_src * PT) * tstride) * 2, 0, (int)(PT * tstride * 16), 0x00020000);
    const bool is_k = wave >= 2;
    const int lane16 = lane * 16;
  for (int rep = 0; rep < 2; ++rep) {
    if (rep == 1) {
        if (rt_second < 0) break;
        rt = rt_second;
        __syncthreads();     // everyone is done with the first tile's LDS buffers before they are refilled
    }
    const int rt_src = (xcd_aware & 4) ? 0 : rt;     // (diagnostic, timing only: every workgroup reads row tile 0's W)
    const int nrb = gc.n_pad >> 4;
    const int rb0 = rt * kBigRb;
    const int rb_end = (rb0 + kBigRb < nrb) ? rb0 + kBigRb : nrb;   // exclusive
    const int npairs = 2 * rb_end;                                   // K extent of the tile's longest row-block
    const int nchunks = (npairs + PPC - 1) / PPC;
    // this wave's DMA pieces of a chunk: waves 0, 1 fetch the W fragments, waves 2, 3 the Kstar fragments;
    // piece i = fragment f = kPieces (wave & 1) + i -> (block f / PPC, pair f % PPC)
    int piece_off[kPieces];     // byte offset of the fragment at q0 = 0 (SGPRs)
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int f = (is_k ? kPiecesK : kPiecesW) * (wave & 1) + i, blk = f / PPC, pr = f % PPC;
        const int rb = rt_src * kBigRb + blk;
        piece_off[i] = is_k ? (int)((blk * tstride + pr * 64) * 16) : (rb * (rb + 1) + pr) * 1024;
    }
    // (ONE loop with the piece count tested inside: written as two loops under `if (is_k) ... else ...`, the host compiler
    // of ROCm 7.2 silently emitted no launch stub for this kernel -- the library then failed to load with an undefined
    // symbol; tests/test_abi.py loads it on the CPU and catches that)
    auto issue_chunk = [&](int chunk, int buf) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            if (i < (is_k ? kPiecesK : kPiecesW)) {
                const int f = (is_k ? kPiecesK : kPiecesW) * (wave & 1) + i;
                lds_v2d* dst = lds + (buf * kFragsBuf + (is_k ? kFragsW : 0) + f) * 64;      // 1 KB per fragment
                const int soff = piece_off[i] + chunk * (PPC * 1024);
                if (is_k)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, (__attribute__((address_space(3))) void*)dst, 16, lane16, soff, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)dst, 16, lane16, soff, 0, 0);
            }
        }
    };

    v4d acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};

    auto compute = [&](int chunk, int buf) {
        const lds_v2d* sa = lds + buf * kFragsBuf * 64 + lane;
        const lds_v2d* sb = lds + (buf * kFragsBuf + kFragsW) * 64 + lane;
#pragma unroll
        for (int pr = 0; pr < PPC; ++pr) {
            v2d a[4], b[NT];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = sa[((4 * wr + m) * PPC + pr) * 64];
#pragma unroll
            for (int n = 0; n < NT; ++n) b[n] = sb[((NT * wc + n) * PPC + pr) * 64];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                // beyond the diagonal of row-block rb0 + 4 wr + m its W fragments are not its own: nothing to add
                // (wave-uniform; only the last chunks of a tile's K extent are affected)
                if (chunk * PPC + pr >= 2 * (rb0 + 4 * wr + m + 1)) continue;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b[n].x, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b[n].y, acc[m][n], 0, 0, 0);
                }
            }
        }
    };

    // Pipeline: NBUF - 1 chunks of DMA in flight.  Iteration c: wait until this wave's pieces of chunk c have landed (all
    // but the NBUF - 2 younger chunks' pieces), barrier -- now chunk c is in LDS for everyone and everyone is done reading
    // buffer (c - 1) % NBUF --, refill that buffer with chunk c + NBUF - 1, compute chunk c.  The barrier is a raw
    // s_barrier: __syncthreads() would drain the DMA in flight (vmcnt(0)).
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c)
        if (c < nchunks) issue_chunk(c, c);
    auto step = [&](int c, auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
        // outstanding after this wait: the pieces of the chunks younger than c that have been issued
        const int younger = (nchunks - 1 - c) < (NBUF - 2) ? (nchunks - 1 - c) : (NBUF - 2);
        if (younger >= 1 && NBUF >= 3) {
            // (counted wait: this wave's pieces of ONE younger chunk may stay in flight)
            if (is_k)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPiecesK) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPiecesW) : "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + NBUF - 1 < nchunks) issue_chunk(c + NBUF - 1, (BUF + NBUF - 1) % NBUF);
        compute(c, BUF);
    };
    static_assert(NBUF == 2 || NBUF == 3, "the wait counts are written for one or two chunks in flight");
    for (int c = 0; c < nchunks; c += NBUF) {
        step(c, std::integral_constant<int, 0>{});
        if (c + 1 < nchunks) step(c + 1, std::integral_constant<int, 1>{});
        if (NBUF == 3 && c + 2 < nchunks) step(c + 2, std::integral_constant<int, 2 % NBUF>{});
    }
    // epilogue: rows < N are squared and summed, rows N .. N + D are the mean / Jacobian rows
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int64_t p = ((int64_t)pg * PT + NT * wc + n) * 16 + (lane & 15);
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
        if (lane < 16) ws.part[((int64_t)d * row_tiles * 2 + rt * 2 + wr) * p128 + p] = s;
    }
  }   // rep
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
        // (partial sums in a FIXED order; the loads of 8 of them in flight together: each is an L2 / fabric round trip)
        double q = 0.0;
        int r = 0;
        for (; r + 8 <= bs.row_parts; r += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws.part[((int64_t)d * bs.row_parts + r + u) * p128 + g];
#pragma unroll
            for (int u = 0; u < 8; ++u) q += v[u];
        }
        for (; r < bs.row_parts; ++r) q += ws.part[((int64_t)d * bs.row_parts + r) * p128 + g];
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
