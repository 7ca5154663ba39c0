// Elite ranking + refit kernel (sx_cem_rank_refit).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "../../include/sx_amd.h"
#include "sx_refit.hpp"
#include "sx_rollout.hpp"  // stamp() in diagnostic builds

namespace sx {

// ---------------------------------------------------------------------------------------------------------------
// sx_cem_rank_refit: one workgroup (16 waves) per problem.  The kernel is a chain of dependent steps on ONE compute
// unit, so what it costs is workgroup barriers and LDS round trips, not arithmetic; every step below is built to
// need one barrier.
//   1. every thread keeps its candidates' 128-bit keys (con, obj) in registers: element i lives in slot i / 1024 of
//      thread i % 1024, so (slot, thread) order is index order;
//   2. the best candidate (smallest (con, obj, index)) by one workgroup reduction;
//   3. the k-th key: the constraint word takes few distinct values, so it is found by walking up the distinct values
//      (counted with ballots); then MSB-first radix select on the objective word, 8 bits per pass, wave-aggregated LDS
//      histograms in three rotating buffers (no clearing barrier), the bin scan done redundantly by every wave (no
//      broadcast barrier), early exit as soon as the bin holding the k-th key is wholly selected;
//   4. ballot compaction in index order (ties broken by the lower index) from one table of per-(slot, wave) counts
//      that every wave scans itself; the best survivor goes to slot 0, the others keep their index order;
//   5. refit: mean / unbiased std over the elites, rows spread over the whole workgroup, values kept in registers
//      between the two passes.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRankThreads = 1024;
constexpr int kRankWaves = kRankThreads / 64;
constexpr int kRankMaxK = 2048;
constexpr int kRankSlots = 16;  // candidates per thread held in registers: P <= 16384

__device__ __forceinline__ unsigned long long sortable_key(double x) {
    if (x != x) return ~0ull;  // NaN last, behind +inf
    x += 0.0;                  // -0.0 -> +0.0: the two compare equal, so they must share a key (ties go by index)
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

// Cross-lane moves on the DPP path of the VALU (no LDS round trip, unlike __shfl_*): row_shr:n shifts within a row of
// 16 lanes, row_bcast:15 / :31 hand the last lane of a row / of the first half on to the following row(s).
template <int CTRL, int ROW_MASK, bool BOUND_ZERO>
__device__ __forceinline__ int dpp_move(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, BOUND_ZERO);
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

// inclusive prefix sum over the 64 lanes of a wave: Hillis-Steele inside the rows, then the row totals
__device__ __forceinline__ int wave_incl_scan(int v, int /*lane*/) {
    v += dpp_move<kDppRowShr1, 0xf, true>(0, v);
    v += dpp_move<kDppRowShr2, 0xf, true>(0, v);
    v += dpp_move<kDppRowShr4, 0xf, true>(0, v);
    v += dpp_move<kDppRowShr8, 0xf, true>(0, v);
    v += dpp_move<kDppRowBcast15, 0xa, false>(0, v);   // rows 1 and 3 += total of the row before
    v += dpp_move<kDppRowBcast31, 0xc, false>(0, v);   // rows 2 and 3 += total of rows 0-1
    return v;
}

// sum over each row of 16 lanes, valid in the row's last lane
__device__ __forceinline__ double row16_sum(double v) {
    auto shr = [](double x, auto ctrl) {
        constexpr int C = decltype(ctrl)::value;
        const long long b = __double_as_longlong(x);
        const int lo = dpp_move<C, 0xf, true>(0, (int)(b & 0xffffffffll)), hi = dpp_move<C, 0xf, true>(0, (int)(b >> 32));
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);   // lanes without a source get +0.0
    };
    v += shr(v, std::integral_constant<int, kDppRowShr1>{});
    v += shr(v, std::integral_constant<int, kDppRowShr2>{});
    v += shr(v, std::integral_constant<int, kDppRowShr4>{});
    v += shr(v, std::integral_constant<int, kDppRowShr8>{});
    return v;
}

// column totals of red[R][Lc] (row-major): 16 lanes per column and a row reduction when the columns fit
// (16 Lc <= NT threads), else one thread per column; `store(c, total)` is called once per column
template <int NT, class Store>
__device__ __forceinline__ void column_totals(const double* red, int Lc, int R, int tid, Store&& store) {
    if (16 * Lc <= NT) {
        const int c = tid >> 4, j = tid & 15;
        double t = 0.0;
        if (c < Lc)
            for (int g = j; g < R; g += 16) t += red[g * Lc + c];
        t = row16_sum(t);
        if (c < Lc && j == 15) store(c, t);
    } else if (tid < Lc) {
        double t = 0.0;
        for (int g = 0; g < R; ++g) t += red[g * Lc + tid];
        store(tid, t);
    }
}

// minimum of (hi, lo, idx) triples over the wave, lexicographic; the result is valid in lane 63
struct RankKey {
    unsigned long long h, l;
    int i;
};
__device__ __forceinline__ bool key_less(const RankKey& a, const RankKey& b) {
    return a.h < b.h || (a.h == b.h && (a.l < b.l || (a.l == b.l && a.i < b.i)));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ RankKey key_min_step(const RankKey& k) {
    // lanes without a source (row edge, masked rows) see the key itself: the minimum is unchanged there
    RankKey o;
    const int hl = (int)(k.h & 0xffffffffu), hh = (int)(k.h >> 32), ll = (int)(k.l & 0xffffffffu), lh = (int)(k.l >> 32);
    const unsigned int ohl = (unsigned int)dpp_move<CTRL, ROW_MASK, false>(hl, hl), ohh = (unsigned int)dpp_move<CTRL, ROW_MASK, false>(hh, hh);
    const unsigned int oll = (unsigned int)dpp_move<CTRL, ROW_MASK, false>(ll, ll), olh = (unsigned int)dpp_move<CTRL, ROW_MASK, false>(lh, lh);
    o.h = ((unsigned long long)ohh << 32) | ohl;
    o.l = ((unsigned long long)olh << 32) | oll;
    o.i = dpp_move<CTRL, ROW_MASK, false>(k.i, k.i);
    return key_less(o, k) ? o : k;
}
__device__ __forceinline__ RankKey wave_min_key(RankKey k) {
    k = key_min_step<kDppRowShr1, 0xf>(k);
    k = key_min_step<kDppRowShr2, 0xf>(k);
    k = key_min_step<kDppRowShr4, 0xf>(k);
    k = key_min_step<kDppRowShr8, 0xf>(k);     // lane 15 of every row: the row's minimum
    k = key_min_step<kDppRowBcast15, 0xa>(k);  // lane 31: rows 0-1, lane 63: rows 2-3
    k = key_min_step<kDppRowBcast31, 0xc>(k);  // lane 63: the wave
    return k;
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
    constexpr int kEntries = SLOTS * kRankWaves;  // (slot, wave) pairs, in index order
    constexpr int EPL = kEntries / 64;            // table entries per lane when a wave scans the table
    static_assert(SLOTS % 4 == 0, "a wave scans the (slot, wave) table with SLOTS / 4 entries per lane");
    __shared__ unsigned int hist[3][256];
    __shared__ unsigned long long red_h[kRankWaves], red_l[kRankWaves];
    __shared__ int red_i[kRankWaves];
    __shared__ unsigned long long walk_min[8];
    __shared__ unsigned long long hdiff_cell;   // bits in which the constraint words of the candidates differ at all
    __shared__ int walk_cnt[8];
    __shared__ int cnt_less[kEntries], cnt_tie[kEntries];
    __shared__ unsigned long long ball_less[kEntries], ball_tie[kEntries];
    __shared__ int sel_idx[kRankMaxK];
    __shared__ double red[kRankThreads];
    __shared__ double col_mean[256];

    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int P = ra.P, k = ra.k;
    const double* con = ra.con + (long long)e * P * ra.cost_stride;
    const double* obj = ra.obj + (long long)e * P * ra.cost_stride;
    const double* act = ra.actions + (long long)e * P * ra.act_stride;

    unsigned long long kh[SLOTS], kl[SLOTS];
    {
        // all 2 SLOTS loads of a thread in flight together: the costs were just written by the rollout's workgroups on
        // other XCDs, so each load is a trip to the fabric, and a load under its own `if (i < P)` waits for the one before
        // (measured at P = 8192: 20 us of the kernel's 52 went here).  Out-of-range slots read the last candidate instead.
        double cv[SLOTS], ov[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int i = s * kRankThreads + tid;
            const long long ii = (long long)(i < P ? i : P - 1) * ra.cost_stride;
            cv[s] = con[ii];
            ov[s] = obj[ii];
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const bool in = s * kRankThreads + tid < P;
            kh[s] = in ? sortable_key(cv[s]) : ~0ull;
            kl[s] = in ? sortable_key(ov[s]) : ~0ull;
        }
    }
    if (tid < 256) {
        hist[0][tid] = 0;
        hist[1][tid] = 0;
        hist[2][tid] = 0;
    }
    if (tid < 8) {
        walk_min[tid] = ~0ull;
        walk_cnt[tid] = 0;
    }
    if (tid == 0) hdiff_cell = 0ull;

#ifdef SX_STAMPS
    const unsigned long long ts0 = stamp();
#endif
    // ---- the best candidate: smallest (con, obj, index) ----
    unsigned long long bh = ~0ull, bl = ~0ull;
    int bi = 0x7fffffff;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = s * kRankThreads + tid;
        // (slots ascend in i, so an equal key never replaces an earlier one; the first valid candidate always enters,
        // which keeps best_idx valid when every key is the NaN key)
        if (i < P && (bi == 0x7fffffff || kh[s] < bh || (kh[s] == bh && kl[s] < bl))) { bh = kh[s]; bl = kl[s]; bi = i; }
    }
    {
        const RankKey wk = wave_min_key(RankKey{bh, bl, bi});
        if (lane == 63) { red_h[wave] = wk.h; red_l[wave] = wk.l; red_i[wave] = wk.i; }
    }
    __syncthreads();  // (also publishes the cleared histograms / walk cells)
#pragma unroll
    for (int w = 0; w < kRankWaves; ++w) {
        const unsigned long long oh = red_h[w], ol = red_l[w];
        const int oi = red_i[w];
        if (oh < bh || (oh == bh && (ol < bl || (ol == bl && oi < bi)))) { bh = oh; bl = ol; bi = oi; }
    }
    const int best_idx = bi;  // uniform
#ifdef SX_STAMPS
    const unsigned long long tsa = stamp();
#endif

    // ---- the k-th key ----
    unsigned long long ph = 0, pl = 0;   // prefix of the k-th key found so far (uniform)
    unsigned long long mh = 0, ml = 0;   // mask of the prefix bits
    int need = k;                        // rank of the k-th key among the candidates matching the prefix
    bool done = false;
    int first_pass = 0;
    {
        // The constraint word takes few distinct values (0 for every feasible particle, then 3 a + 10 b), so its
        // k-th smallest value is found by walking up the distinct values, at most kWalk of them (beyond that the general
        // passes below take over: a long horizon has dozens of values).  The smallest one is the best candidate's; its
        // multiplicity comes from ballots.  The first iteration's barrier also publishes in which BYTES the constraint
        // words differ at all: small integers stored as doubles share their low mantissa bytes, and a radix pass over
        // a byte in which no two candidates differ would only confirm that (6 of 8 passes at config 3).
        constexpr int kWalk = 5;
        unsigned long long cur = bh;
        int acc = 0;  // candidates below `cur`
        {
            unsigned long long dif = 0ull;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (s * kRankThreads + tid < P) dif |= kh[s] ^ bh;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dif |= __shfl_xor(dif, off);
            if (lane == 0 && dif) atomicOr(&hdiff_cell, dif);
        }
        for (int it = 0; it < kWalk; ++it) {
            // ONE barrier per distinct value: the multiplicity of `cur` and the next value above it in the same round
            int cnt = 0;
            unsigned long long mn = ~0ull;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const bool in = s * kRankThreads + tid < P;
                cnt += __popcll(__ballot(in && kh[s] == cur));
                if (in && kh[s] > cur && kh[s] < mn) mn = kh[s];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long om = __shfl_xor(mn, off);
                if (om < mn) mn = om;
            }
            if (lane == 0) {
                atomicAdd(&walk_cnt[it], cnt);
                if (mn != ~0ull) atomicMin(&walk_min[it], mn);
            }
            __syncthreads();
            cnt = walk_cnt[it];
            if (acc + cnt >= k) {   // the k-th key has this constraint word
                ph = cur;
                mh = ~0ull;
                need = k - acc;
                first_pass = 8;
                break;
            }
            acc += cnt;
            cur = walk_min[it];
            if (cur == ~0ull) break;  // no further value (cannot happen while acc < k <= P, kept for safety)
        }
    }
#ifdef SX_STAMPS
    const unsigned long long tsb = stamp();
    int npass = 0;
    unsigned long long pass_t[3] = {0, 0, 0};
#endif
    const unsigned long long hdiff = hdiff_cell;   // (published by the walk's first barrier)
    int hp = 0;                                    // executed passes: the histogram buffers rotate with it
    for (int pass = first_pass; pass < 16 && !done; ++pass) {
        const int shift = 56 - 8 * (pass & 7);
        const bool in_hi = pass < 8;
        if (in_hi && ((hdiff >> shift) & 255ull) == 0ull) {
            // every candidate has the best candidate's byte here: it joins the prefix without a pass
            ph |= bh & (255ull << shift);
            mh |= 255ull << shift;
            continue;
        }
#ifdef SX_STAMPS
        ++npass;
#endif
        unsigned int* h = hist[hp % 3];
        // the buffer of the next pass was last read two passes ago: clear it now, behind this pass's barrier
        if (tid < 256) hist[(hp + 1) % 3][tid] = 0;
        ++hp;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int i = s * kRankThreads + tid;
            const bool match = (i < P) && ((kh[s] & mh) == ph) && ((kl[s] & ml) == pl);
            const unsigned int digit = match ? (unsigned int)(((in_hi ? kh[s] : kl[s]) >> shift) & 255ull) : 0xffffffffu;
            // (a ballot-per-distinct-digit aggregation was measured three times, round 2 for the constraint word alone with
            // its handful of digits per wave: slower than the plain atomics it saves)
            const unsigned int first = __builtin_amdgcn_readfirstlane(digit);
            if (__all(digit == first)) {
                if (first != 0xffffffffu && lane == 0) atomicAdd(&h[first], 64u);
            } else if (match) {
                atomicAdd(&h[digit], 1u);
            }
        }
#ifdef SX_STAMPS
        const unsigned long long tp1 = stamp();
#endif
        __syncthreads();
#ifdef SX_STAMPS
        const unsigned long long tp2 = stamp();
#endif
        // every wave scans the 256 bins itself: lane l owns bins 4l .. 4l+3
        const unsigned int c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
        const int mine = (int)(c0 + c1 + c2 + c3);
        const int incl = wave_incl_scan(mine, lane);
        const int before = incl - mine;
        const bool owner = need > before && need <= incl;  // exactly one lane
        int rem = need - before;
        int dsel = 4 * lane;
        unsigned int cnt = c0;
        if (rem > (int)c0) { rem -= c0; dsel++; cnt = c1;
            if (rem > (int)c1) { rem -= c1; dsel++; cnt = c2;
                if (rem > (int)c2) { rem -= c2; dsel++; cnt = c3; } } }
        const int src = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)__ballot(owner)) - 1);
        const int digit_sel = __builtin_amdgcn_readlane(dsel, src);
        need = __builtin_amdgcn_readlane(rem, src);
        done = __builtin_amdgcn_readlane((int)(rem == (int)cnt), src) != 0;  // whole bin selected: lower digits do not matter
        const unsigned long long dg = (unsigned long long)digit_sel << shift, mk = 255ull << shift;
        if (in_hi) { ph |= dg; mh |= mk; } else { pl |= dg; ml |= mk; }
#ifdef SX_STAMPS
        const unsigned long long tp3 = stamp();
        if (npass == 1) { pass_t[0] = tp1 - tsb; pass_t[1] = tp2 - tp1; pass_t[2] = tp3 - tp2; }
#endif
    }
#ifdef SX_STAMPS
    const unsigned long long ts1 = stamp();
#endif
    // ---- compaction ----
    // Candidates whose masked key is below the prefix are selected; of those equal to it, the first `need` in index
    // order (all of them after an early exit).
    const int n_less_total = k - need;
    unsigned long long my_bl[SLOTS], my_bt[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = s * kRankThreads + tid;
        const unsigned long long a_h = kh[s] & mh, a_l = kl[s] & ml;
        const bool valid = i < P;
        const bool less = valid && (a_h < ph || (a_h == ph && a_l < pl));
        const bool tie = valid && a_h == ph && a_l == pl;
        my_bl[s] = __ballot(less);
        my_bt[s] = __ballot(tie);
        if (lane == 0) {
            cnt_less[s * kRankWaves + wave] = __popcll(my_bl[s]);
            cnt_tie[s * kRankWaves + wave] = __popcll(my_bt[s]);
            ball_less[s * kRankWaves + wave] = my_bl[s];
            ball_tie[s * kRankWaves + wave] = my_bt[s];
        }
    }
    __syncthreads();
    // exclusive prefix over the (slot, wave) table, by every wave for itself: lane l holds entries EPL l .. EPL l + EPL-1
    int tl[EPL], tt[EPL], sum_l = 0, sum_t = 0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        tl[j] = cnt_less[EPL * lane + j];
        tt[j] = cnt_tie[EPL * lane + j];
        sum_l += tl[j];
        sum_t += tt[j];
    }
    const int ex_l = wave_incl_scan(sum_l, lane) - sum_l, ex_t = wave_incl_scan(sum_t, lane) - sum_t;
    auto table_offset = [&](const int (&cell)[EPL], int ex, int entry) {  // entry is wave-uniform
        const int src = __builtin_amdgcn_readfirstlane(entry / EPL), sub = entry % EPL;
        int off = __builtin_amdgcn_readlane(ex, src);
#pragma unroll
        for (int j = 0; j < EPL; ++j)
            if (j < sub) off += __builtin_amdgcn_readlane(cell[j], src);
        return off;
    };
    // where the best candidate would land: it goes to slot 0 instead, those in front of it move up by one
    int best_slot;
    {
        const bool best_less = ((bh & mh) < ph) || ((bh & mh) == ph && (bl & ml) < pl);
        const int bs = best_idx / kRankThreads, bt = best_idx % kRankThreads;
        const int entry = bs * kRankWaves + (bt >> 6);
        const unsigned long long before = (1ull << (bt & 63)) - 1ull;
        if (best_less)
            best_slot = table_offset(tl, ex_l, entry) + __popcll(ball_less[entry] & before);
        else  // it matches the prefix (after an early exit the ties differ in their lower bits: it need not be the first)
            best_slot = n_less_total + table_offset(tt, ex_t, entry) + __popcll(ball_tie[entry] & before);
    }
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        if (s * kRankThreads >= P) break;
        const int i = s * kRankThreads + tid;
        const int entry = s * kRankWaves + wave;
        // The table offsets are read with v_readlane from lanes that hold no selected candidate themselves: they MUST be
        // taken here, in wave-uniform control flow.  Inside the `if` below the source lane may be inactive, and the
        // value an inactive lane holds is undefined (the compiler is free to sink its computation into the branch) --
        // round 1 did that, and at config 3 (8 slots, many distinct constraint costs) one elite in ~300 rankings landed
        // in a neighbour's slot, leaving a stale index behind (tests/golden/rank_case_r02.npz).
        const int off_less = table_offset(tl, ex_l, entry);
        const int off_tie = table_offset(tt, ex_t, entry);
        int slot = -1;
        if ((my_bl[s] >> lane) & 1ull) {
            slot = off_less + __popcll(my_bl[s] & below);
        } else if ((my_bt[s] >> lane) & 1ull) {
            const int r = off_tie + __popcll(my_bt[s] & below);
            if (r < need) slot = n_less_total + r;
        }
        if (slot >= 0) sel_idx[slot == best_slot ? 0 : (slot < best_slot ? slot + 1 : slot)] = i;
    }
    __syncthreads();
#ifdef SX_STAMPS
    const unsigned long long ts2 = stamp();
    const unsigned long long ts3 = ts2;
#endif
    // ---- outputs ----
    const int L = ra.row_len;
    if (ra.elite_idx)
        for (int i = tid; i < k; i += kRankThreads) ra.elite_idx[(long long)e * k + i] = sel_idx[i];
    if (ra.elite_rows) {
        // k x (2 + L) gathered elements, 4 per thread and trip with their loads in flight together (the output is a
        // different buffer: without the restrict qualifiers every load would wait for the store before it)
        const int W = 2 + L;
        double* __restrict__ rows_out = ra.elite_rows + (long long)e * k * W;
        const double* __restrict__ con_r = con;
        const double* __restrict__ obj_r = obj;
        const double* __restrict__ act_r = act;
        const int total = k * W;
        for (int i0 = tid; i0 < total; i0 += 4 * kRankThreads) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * kRankThreads;
                v[u] = 0.0;
                if (i < total) {
                    const int r = i / W, c = i - r * W;
                    const int src = sel_idx[r];
                    v[u] = (c == 0) ? con_r[(long long)src * ra.cost_stride]
                                    : (c == 1) ? obj_r[(long long)src * ra.cost_stride] : act_r[(long long)src * ra.act_stride + (c - 2)];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * kRankThreads;
                if (i < total) rows_out[i] = v[u];
            }
        }
    }
    if (ra.best)
        for (int c = tid; c < L; c += kRankThreads) ra.best[(long long)e * L + c] = act[(long long)best_idx * ra.act_stride + c];
    if (ra.best_ok && tid == 0) ra.best_ok[e] = (con[(long long)best_idx * ra.cost_stride] == 0.0) ? 1 : 0;
    if (ra.mean) {
        // columns in chunks of up to 256; thread t sums rows t / Lc, t / Lc + R, ... of column t % Lc.  The rows are
        // gathered kKeep at a time with all loads of a batch in flight together (a row is reached through sel_idx, so a
        // plain loop would pay one dependent L2 round trip per row: 24 of them at k = 819, L = 30); the first batch stays
        // in registers for the second pass, the others are read again (L2 hits).
        constexpr int kKeep = 8;
        for (int c0 = 0; c0 < L; c0 += 256) {
            const int Lc = (L - c0) < 256 ? (L - c0) : 256;
            const int R = kRankThreads / Lc;  // row groups
            const int c = tid % Lc, r0 = tid / Lc;
            const bool active = r0 < R;
            auto gather = [&](int base, double (&v)[kKeep]) {
#pragma unroll
                for (int j = 0; j < kKeep; ++j) {
                    const int r = base + j * R;
                    v[j] = (r < k) ? act[(long long)sel_idx[r] * ra.act_stride + c0 + c] : 0.0;
                }
            };
            double keep[kKeep];
            double s = 0.0;
            if (active) {
                gather(r0, keep);
#pragma unroll
                for (int j = 0; j < kKeep; ++j) s += keep[j];
                for (int base = r0 + kKeep * R; base < k; base += kKeep * R) {
                    double v[kKeep];
                    gather(base, v);
#pragma unroll
                    for (int j = 0; j < kKeep; ++j) s += v[j];
                }
            }
            red[tid] = s;
            __syncthreads();
            column_totals<kRankThreads>(red, Lc, R, tid, [&](int cc, double t) { col_mean[cc] = t / k; });
            __syncthreads();
            const double mu = col_mean[c];
            double ss = 0.0;
            if (active) {
#pragma unroll
                for (int j = 0; j < kKeep; ++j) {
                    const double dv = keep[j] - mu;
                    if (r0 + j * R < k) ss += dv * dv;
                }
                for (int base = r0 + kKeep * R; base < k; base += kKeep * R) {
                    double v[kKeep];
                    gather(base, v);
#pragma unroll
                    for (int j = 0; j < kKeep; ++j) {
                        const double dv = v[j] - mu;
                        if (base + j * R < k) ss += dv * dv;
                    }
                }
            }
            red[tid] = ss;  // (the column owners finished reading red before the barrier above)
            __syncthreads();
            column_totals<kRankThreads>(red, Lc, R, tid, [&](int cc, double t) {
                ra.mean[(long long)e * L + c0 + cc] = col_mean[cc];
                if (ra.std) ra.std[(long long)e * L + c0 + cc] = (k > 1) ? sqrt(t / (k - 1)) : 0.0;
            });
            __syncthreads();
        }
    }
#ifdef SX_STAMPS
    const unsigned long long ts4 = stamp();
    if (g_stamp_buf && tid == 0 && e == 0) {
        g_stamp_buf[0] = ts1 - ts0; g_stamp_buf[1] = ts2 - ts1; g_stamp_buf[2] = ts3 - ts2; g_stamp_buf[3] = ts4 - ts3;
        g_stamp_buf[4] = tsa - ts0; g_stamp_buf[5] = tsb - tsa; g_stamp_buf[6] = ts1 - tsb; g_stamp_buf[7] = npass;
        g_stamp_buf[8] = pass_t[0]; g_stamp_buf[9] = pass_t[1]; g_stamp_buf[10] = pass_t[2];
    }
#endif
}

}  // namespace sx
