// Elite ranking + refit kernel (sx_cem_rank_refit).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sx_amd.h"
#include "sx_rollout.hpp"  // stamp() in diagnostic builds

namespace sx {

// ---------------------------------------------------------------------------------------------------------------
// sx_cem_rank_refit: one workgroup (16 waves) per problem.
//   1. every thread keeps its candidates' 128-bit keys (con, obj) in registers: element i lives in slot i / 1024 of
//      thread i % 1024, so (slot, thread) order is index order;
//   2. MSB-first radix select of the k-th key, 8 bits per pass: wave-aggregated LDS histogram (one atomic per wave when
//      all lanes agree -- the common case in the high bytes), bin scan by one wave, early exit as soon as the bin
//      holding the k-th key is wholly selected;
//   3. ballot compaction in index order (ties broken by the lower index); the best survivor is moved to the front,
//      the others stay where the compaction put them (nothing downstream needs them sorted);
//   4. refit: mean / unbiased std over the elites, rows spread over the whole workgroup.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRankThreads = 1024;
constexpr int kRankWaves = kRankThreads / 64;
constexpr int kRankMaxK = 2048;
constexpr int kRankSlots = 16;  // candidates per thread held in registers: P <= 16384

__device__ __forceinline__ unsigned long long sortable_key(double x) {
    if (x != x) return ~0ull;  // NaN last
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

struct RankArgs {
    int P, k, row_len;
    const double* con;
    const double* obj;
    long long cost_stride;
    const double* actions;
    long long act_stride;
    int* elite_idx;
    double* elite_rows;
    double* mean;
    double* std;
    double* best;
    int* best_ok;
};

template <int SLOTS>
__global__ __launch_bounds__(kRankThreads) void cem_rank_kernel(RankArgs ra) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long sel_hi[kRankMaxK], sel_lo[kRankMaxK];
    __shared__ int sel_idx[kRankMaxK];
    __shared__ double red[kRankThreads];
    __shared__ double col_mean[256];
    __shared__ int wave_cnt[kRankWaves][2];
    __shared__ unsigned long long red_u64[kRankWaves], red_lo[kRankWaves];
    __shared__ int sh_digit, sh_need, sh_done;

    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int P = ra.P, k = ra.k;
    const double* con = ra.con + (long long)e * P * ra.cost_stride;
    const double* obj = ra.obj + (long long)e * P * ra.cost_stride;
    const double* act = ra.actions + (long long)e * P * ra.act_stride;

    unsigned long long kh[SLOTS], kl[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = s * kRankThreads + tid;
        kh[s] = ~0ull;
        kl[s] = ~0ull;
        if (i < P) {
            kh[s] = sortable_key(con[(long long)i * ra.cost_stride]);
            kl[s] = sortable_key(obj[(long long)i * ra.cost_stride]);
        }
    }

#ifdef SX_STAMPS
    const unsigned long long ts0 = stamp();
#endif
    // ---- radix select ----
    unsigned long long ph = 0, pl = 0;   // prefix of the k-th key found so far (uniform)
    unsigned long long mh = 0, ml = 0;   // mask of the prefix bits
    int need = k;                        // rank of the k-th key among the candidates matching the prefix
    bool done = false;
    int first_pass = 0;
    {
        // The constraint word takes few distinct values (0 for every feasible particle, then 3 a + 10 b), so its
        // k-th smallest value is found by walking up the distinct values: one (min, multiplicity) reduction per value,
        // at most 8 of them, instead of eight radix passes.  (Beyond 8 the general passes below take over.)
        unsigned long long floor_key = 0;  // only keys >= floor_key are still in play
        int acc = 0;                       // candidates below floor_key
        for (int it = 0; it < 8; ++it) {
            unsigned long long mn = ~0ull;
            int cnt = 0;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const bool in_play = (s * kRankThreads + tid < P) && kh[s] >= floor_key;
                if (in_play) {
                    if (kh[s] < mn) { mn = kh[s]; cnt = 1; } else if (kh[s] == mn) { ++cnt; }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long om = __shfl_xor(mn, off);
                const int oc = __shfl_xor(cnt, off);
                if (om < mn) { mn = om; cnt = oc; } else if (om == mn) { cnt += oc; }
            }
            if (lane == 0) { red_u64[wave] = mn; wave_cnt[wave][0] = cnt; }
            __syncthreads();
            mn = red_u64[0];
            cnt = wave_cnt[0][0];
#pragma unroll
            for (int w = 1; w < kRankWaves; ++w) {
                const unsigned long long om = red_u64[w];
                const int oc = wave_cnt[w][0];
                if (om < mn) { mn = om; cnt = oc; } else if (om == mn) { cnt += oc; }
            }
            __syncthreads();
            if (acc + cnt >= k) {   // the k-th key has this constraint word
                ph = mn;
                mh = ~0ull;
                need = k - acc;
                first_pass = 8;
                break;
            }
            acc += cnt;
            floor_key = mn + 1;
        }
    }
    for (int pass = first_pass; pass < 16 && !done; ++pass) {
        const int shift = 56 - 8 * (pass & 7);
        const bool in_hi = pass < 8;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int i = s * kRankThreads + tid;
            const bool match = (i < P) && ((kh[s] & mh) == ph) && ((kl[s] & ml) == pl);
            const unsigned int digit = match ? (unsigned int)(((in_hi ? kh[s] : kl[s]) >> shift) & 255ull) : 0xffffffffu;
            const unsigned int first = __builtin_amdgcn_readfirstlane(digit);
            if (__all(digit == first)) {
                if (first != 0xffffffffu && lane == 0) atomicAdd(&hist[first], 64u);
            } else if (match) {
                atomicAdd(&hist[digit], 1u);
            }
        }
        __syncthreads();
        if (wave == 0) {
            // lane l owns bins 4l .. 4l+3
            const unsigned int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
            const int mine = (int)(c0 + c1 + c2 + c3);
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            const int before = incl - mine;
            if (need > before && need <= incl) {
                int rem = need - before;
                int dsel = 4 * lane;
                unsigned int cnt = c0;
                if (rem > (int)c0) { rem -= c0; dsel++; cnt = c1;
                    if (rem > (int)c1) { rem -= c1; dsel++; cnt = c2;
                        if (rem > (int)c2) { rem -= c2; dsel++; cnt = c3; } } }
                sh_digit = dsel;
                sh_need = rem;
                sh_done = (rem == (int)cnt) ? 1 : 0;  // the whole bin is selected: no need to look at lower digits
            }
        }
        __syncthreads();
        const unsigned long long dg = (unsigned long long)sh_digit << shift, mk = 255ull << shift;
        if (in_hi) { ph |= dg; mh |= mk; } else { pl |= dg; ml |= mk; }
        need = sh_need;
        done = sh_done != 0;
    }
#ifdef SX_STAMPS
    const unsigned long long ts1 = stamp();
#endif
    // Candidates whose masked key is below the prefix are selected; of those equal to it, the first `need` in index
    // order (all of them after an early exit).
    const int n_less_total = k - need;
    int base_less = 0, base_tie = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        if (s * kRankThreads >= P) break;
        const int i = s * kRankThreads + tid;
        const unsigned long long a_h = kh[s] & mh, a_l = kl[s] & ml;
        const bool valid = i < P;
        const bool less = valid && (a_h < ph || (a_h == ph && a_l < pl));
        const bool tie = valid && a_h == ph && a_l == pl;
        const unsigned long long bl = __ballot(less), bt = __ballot(tie);
        if (lane == 0) {
            wave_cnt[wave][0] = __popcll(bl);
            wave_cnt[wave][1] = __popcll(bt);
        }
        __syncthreads();
        int off_less = base_less, off_tie = base_tie, tot_less = 0, tot_tie = 0;
#pragma unroll
        for (int w = 0; w < kRankWaves; ++w) {
            const int cl = wave_cnt[w][0], ct = wave_cnt[w][1];
            if (w < wave) { off_less += cl; off_tie += ct; }
            tot_less += cl;
            tot_tie += ct;
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (less) {
            const int slot = off_less + __popcll(bl & below);
            sel_hi[slot] = kh[s]; sel_lo[slot] = kl[s]; sel_idx[slot] = i;
        } else if (tie) {
            const int r = off_tie + __popcll(bt & below);
            if (r < need) {
                const int slot = n_less_total + r;
                sel_hi[slot] = kh[s]; sel_lo[slot] = kl[s]; sel_idx[slot] = i;
            }
        }
        base_less += tot_less;
        base_tie += tot_tie;
        __syncthreads();
    }
#ifdef SX_STAMPS
    const unsigned long long ts2 = stamp();
#endif
    // ---- the elites stay where the compaction put them; only the best one is moved to the front ----
    {
        unsigned long long bh = ~0ull, bl = ~0ull;
        int bi = 0x7fffffff, bslot = 0;
        for (int i = tid; i < k; i += kRankThreads) {
            const unsigned long long h = sel_hi[i], l = sel_lo[i];
            const int ix = sel_idx[i];
            if (h < bh || (h == bh && (l < bl || (l == bl && ix < bi)))) { bh = h; bl = l; bi = ix; bslot = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long oh = __shfl_xor(bh, off), ol = __shfl_xor(bl, off);
            const int oi = __shfl_xor(bi, off), os = __shfl_xor(bslot, off);
            if (oh < bh || (oh == bh && (ol < bl || (ol == bl && oi < bi)))) { bh = oh; bl = ol; bi = oi; bslot = os; }
        }
        if (lane == 0) { red_u64[wave] = bh; red_lo[wave] = bl; wave_cnt[wave][0] = bi; wave_cnt[wave][1] = bslot; }
        __syncthreads();
        if (tid == 0) {
            int best_w = 0;
            for (int w = 1; w < kRankWaves; ++w) {
                const unsigned long long oh = red_u64[w], ol = red_lo[w], ch = red_u64[best_w], cl = red_lo[best_w];
                if (oh < ch || (oh == ch && (ol < cl || (ol == cl && wave_cnt[w][0] < wave_cnt[best_w][0])))) best_w = w;
            }
            const int s = wave_cnt[best_w][1];
            const unsigned long long th = sel_hi[0], tl = sel_lo[0];
            const int ti = sel_idx[0];
            sel_hi[0] = sel_hi[s]; sel_lo[0] = sel_lo[s]; sel_idx[0] = sel_idx[s];
            sel_hi[s] = th; sel_lo[s] = tl; sel_idx[s] = ti;
        }
        __syncthreads();
    }
#ifdef SX_STAMPS
    const unsigned long long ts3 = stamp();
#endif
    // ---- outputs ----
    const int L = ra.row_len;
    if (ra.elite_idx)
        for (int i = tid; i < k; i += kRankThreads) ra.elite_idx[(long long)e * k + i] = sel_idx[i];
    if (ra.elite_rows) {
        const int W = 2 + L;
        for (int i = tid; i < k * W; i += kRankThreads) {
            const int r = i / W, c = i - r * W;
            const int src = sel_idx[r];
            double v;
            if (c == 0)
                v = con[(long long)src * ra.cost_stride];
            else if (c == 1)
                v = obj[(long long)src * ra.cost_stride];
            else
                v = act[(long long)src * ra.act_stride + (c - 2)];
            ra.elite_rows[((long long)e * k + r) * W + c] = v;
        }
    }
    if (ra.best)
        for (int c = tid; c < L; c += kRankThreads) ra.best[(long long)e * L + c] = act[(long long)sel_idx[0] * ra.act_stride + c];
    if (ra.best_ok && tid == 0) ra.best_ok[e] = (con[(long long)sel_idx[0] * ra.cost_stride] == 0.0) ? 1 : 0;
    if (ra.mean) {
        // columns in chunks of up to 256; thread t sums rows t / Lc, t / Lc + R, ... of column t % Lc
        for (int c0 = 0; c0 < L; c0 += 256) {
            const int Lc = (L - c0) < 256 ? (L - c0) : 256;
            const int R = kRankThreads / Lc;  // row groups
            const int c = tid % Lc, r0 = tid / Lc;
            const bool active = r0 < R;
            double s = 0.0;
            if (active)
                for (int r = r0; r < k; r += R) s += act[(long long)sel_idx[r] * ra.act_stride + c0 + c];
            red[tid] = s;
            __syncthreads();
            if (tid < Lc) {
                double t = 0.0;
                for (int g = 0; g < R; ++g) t += red[g * Lc + tid];
                col_mean[tid] = t / k;
            }
            __syncthreads();
            const double mu = col_mean[c];
            double ss = 0.0;
            if (active)
                for (int r = r0; r < k; r += R) {
                    const double dv = act[(long long)sel_idx[r] * ra.act_stride + c0 + c] - mu;
                    ss += dv * dv;
                }
            red[tid] = ss;
            __syncthreads();
            if (tid < Lc) {
                double t = 0.0;
                for (int g = 0; g < R; ++g) t += red[g * Lc + tid];
                ra.mean[(long long)e * L + c0 + tid] = col_mean[tid];
                if (ra.std) ra.std[(long long)e * L + c0 + tid] = (k > 1) ? sqrt(t / (k - 1)) : 0.0;
            }
            __syncthreads();
        }
    }
#ifdef SX_STAMPS
    const unsigned long long ts4 = stamp();
    if (g_stamp_buf && tid == 0 && e == 0) {
        g_stamp_buf[0] = ts1 - ts0; g_stamp_buf[1] = ts2 - ts1; g_stamp_buf[2] = ts3 - ts2; g_stamp_buf[3] = ts4 - ts3;
    }
#endif
}

}  // namespace sx
