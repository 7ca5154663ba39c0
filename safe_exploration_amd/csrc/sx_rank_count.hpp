// Elite ranking by counting, spread over the whole chip (sx_cem_rank_refit for one or two problems of up to 8192 candidates).
#pragma once
#include <hip/hip_runtime.h>

#include "sx_rank.hpp"

namespace sx {

// ---------------------------------------------------------------------------------------------------------------
// cem_rank_kernel (sx_rank.hpp) selects the k-th key with a chain of dependent steps on ONE compute unit: ~15 us of
// barriers and LDS round trips at P = 4096 while 255 CUs idle, 13 % of a config-2 solve.  Here
// the rank of every candidate is COUNTED instead -- rank(a) = #{b : (con, obj, index)_b < (con, obj, index)_a} -- which
// has no dependent steps at all and spreads over P / 16 workgroups:
//   1. every workgroup stages all P keys in LDS as 128-bit sortable integers (16 B each; the costs were just written
//      by the rollout, one pass over them per workgroup out of L2) -- each wave the eighth it will walk itself, so no
//      workgroup barrier separates staging from counting;
//   2. the workgroup owns 16 candidates ("a").  Lane l of wave w holds a = l & 15 and walks the candidates
//      b = w P/8 + 4 i + (l >> 4): one ds_read_b128 (a broadcast within the 16 lanes) and one 128-bit compare per trip.
//      Ties go to the lower index: b counts as smaller when key_b < key_a + [b < a's index].  The quarter-waves'
//      b differ by less than 4 and the workgroup's candidates are 16 aligned ones, so `b < a's index` is WAVE-UNIFORM
//      except inside the workgroup's own 16 candidates: the walk is two plain loops (before / after them) with a
//      loop-invariant comparand each, and the 16 x 16 pairs of the own tile take the general form once;
//   3. the 32 partial counts of a candidate are added (two shuffles + one LDS round); a candidate of rank r < k writes
//      its row [con, obj, actions...] to slot r: the elite rows come out SORTED, the best in slot 0;
//   4. refit: the workgroup that finishes last (one atomic ticket per problem, self-resetting) reads the k rows back --
//      contiguous, no gather -- and computes mean and unbiased std in two passes, values kept in registers.
// Work is O(P^2 / 64) wave-compares: 262 k at P = 4096, 2.6 us over 256 CUs (the kernel: 7.8 us without the refit, 14 us with
// it); the one-workgroup kernel stays for larger or many simultaneous problems (the launcher chooses by shape
// only -- sx_cem_rank_counts -- so every rank of a multi-GPU solve takes the same one).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kCountThreads = 512;
constexpr int kCountWaves = kCountThreads / 64;
constexpr int kCountMaxP = 8192;       // 128 KB of keys in LDS
constexpr int kCountMaxE = 64;         // ticket cells per launch slot
constexpr int kCountMaxGrid = 768;      // workgroups beyond which the one-workgroup-per-problem kernel is taken (measured)
constexpr int kCountTicketSlots = 64;  // launches in flight at once before two could share a ticket cell

// one ticket per (launch slot, problem): counts the workgroups that have delivered their rows; atomicInc wraps it back
// to zero with the last one, so it never needs clearing
__device__ unsigned int g_rank_tickets[kCountTicketSlots * kCountMaxE];

struct CountKey {
    unsigned long long h, l;
};

// keys are (sortable(con), sortable(obj)); the objective word is kept below 2^64 - 1 so that `l + 1` cannot wrap (every
// NaN maps to the one largest key, which loses nothing: NaNs rank last, among themselves by index)
__device__ __forceinline__ CountKey count_key(double con, double obj) {
    CountKey k;
    k.h = sortable_key(con);
    const unsigned long long l = sortable_key(obj);
    k.l = l == ~0ull ? ~0ull - 1ull : l;
    return k;
}

// cnt += [key_b < (h, l)] as 128-bit integers: the borrow of b - a through four full-rate 32-bit subtractions, added with
// a fifth (three v_cmp_*_u64 at half rate + the select logic took 32 cycles per trip, this takes 20)
__device__ __forceinline__ void count_below(int& cnt, const CountKey& b, unsigned long long h, unsigned long long l) {
    unsigned int junk;
    asm("v_sub_co_u32_e32 %1, vcc, %2, %6\n\t"
        "v_subb_co_u32_e32 %1, vcc, %3, %7, vcc\n\t"
        "v_subb_co_u32_e32 %1, vcc, %4, %8, vcc\n\t"
        "v_subb_co_u32_e32 %1, vcc, %5, %9, vcc\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc"
        : "+v"(cnt), "=&v"(junk)
        : "v"((unsigned int)b.l), "v"((unsigned int)(b.l >> 32)), "v"((unsigned int)b.h), "v"((unsigned int)(b.h >> 32)),
          "v"((unsigned int)l), "v"((unsigned int)(l >> 32)), "v"((unsigned int)h), "v"((unsigned int)(h >> 32))
        : "vcc");
}

// the same on 96 bits, for waves whose constraint words all have a zero low half (small integers stored as doubles: every
// cost the solver itself produces) -- one instruction less per trip
__device__ __forceinline__ void count_below96(int& cnt, const CountKey& b, unsigned long long h, unsigned long long l) {
    unsigned int junk;
    asm("v_sub_co_u32_e32 %1, vcc, %2, %5\n\t"
        "v_subb_co_u32_e32 %1, vcc, %3, %6, vcc\n\t"
        "v_subb_co_u32_e32 %1, vcc, %4, %7, vcc\n\t"
        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc"
        : "+v"(cnt), "=&v"(junk)
        : "v"((unsigned int)b.l), "v"((unsigned int)(b.l >> 32)), "v"((unsigned int)(b.h >> 32)), "v"((unsigned int)l),
          "v"((unsigned int)(l >> 32)), "v"((unsigned int)(h >> 32))
        : "vcc");
}

__global__ __launch_bounds__(kCountThreads) void cem_rank_count_kernel(RankArgs ra, unsigned int* __restrict__ tickets) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long count_smem[];
    CountKey* const keys = reinterpret_cast<CountKey*>(count_smem);   // [p_pad]
    __shared__ int part[kCountWaves][16];
    __shared__ int last_flag;
    __shared__ double red[kCountThreads];
    __shared__ double col_mean[256];

    const int tile = blockIdx.x, e = blockIdx.y;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // (uniform: scalar loop control below)
    const int P = ra.P, k = ra.k, L = ra.row_len;
    const int p_pad = (P + 127) & ~127;   // 8 waves x 16-aligned ranges
    const double* __restrict__ con = ra.con + (long long)e * P * ra.cost_stride;
    const double* __restrict__ obj = ra.obj + (long long)e * P * ra.cost_stride;
    const double* __restrict__ act = ra.actions + (long long)e * P * ra.act_stride;
#ifdef SX_STAMPS
    unsigned long long cs[10];
    cs[0] = stamp();
#endif

    // ---- 1. keys to LDS.  A wave walks only ITS eighth of the candidates, so it stages that eighth itself and waits for
    // nobody: no workgroup barrier between staging and counting.  The 16 own candidates come straight from memory. ----
    const int a = lane & 15, g = lane >> 4;
    const int n_w = p_pad >> 3;     // candidates per wave, a multiple of 16
    const int base = wave * n_w;
    const int ia = tile * 16 + a;   // < p_pad
    const long long iac = (long long)(ia < P ? ia : P - 1) * ra.cost_stride;
    const double a_con = con[iac], a_obj = obj[iac];
    // the row of candidate a2 = tid >> 5, column tid & 31: requested now, stored once its rank is known
    const int a2 = tid >> 5, c = tid & 31;
    const int i2 = tile * 16 + a2;
    const int W = 2 + L;
    double row_v = 0.0;
    if (i2 < P && c < W)
        row_v = (c == 0) ? con[(long long)i2 * ra.cost_stride]
                         : (c == 1) ? obj[(long long)i2 * ra.cost_stride] : act[(long long)i2 * ra.act_stride + (c - 2)];
    unsigned int low_bits = 0;   // OR of the low halves of the constraint words this wave compares
    for (int i0 = lane; i0 < n_w; i0 += 8 * 64) {
        double cv[8], ov[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + i0 + u * 64;
            const long long ii = (long long)(i < P ? i : P - 1) * ra.cost_stride;
            cv[u] = con[ii];
            ov[u] = obj[ii];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + i0 + u * 64;
            if (i0 + u * 64 < n_w) {
                CountKey kk = count_key(cv[u], ov[u]);
                if (i >= P) kk.h = kk.l = ~0ull;   // padding: never below anything
                else low_bits |= (unsigned int)kk.h;
                keys[i] = kk;
            }
        }
    }
    CountKey ka = count_key(a_con, a_obj);
    if (ia >= P) ka.h = ka.l = ~0ull;
    else low_bits |= (unsigned int)ka.h;
    const bool narrow = !__any(low_bits != 0u);   // (uniform)
#ifdef SX_STAMPS
    cs[1] = stamp();
#endif
    // the wave's own LDS writes, read back by its other lanes: wave-level ordering is all it takes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef SX_STAMPS
    cs[2] = stamp();
#endif

    // ---- 2. count ----
    const int n_it = n_w >> 2;
    int it0 = (tile * 16 - base) / 4;   // first trip that touches the own tile (exact: both are multiples of 16)
    int it1 = it0 + 4;
    const bool own_here = it0 >= 0 && it0 < n_it;   // (uniform) the own tile lies in this wave's range
    it0 = it0 < 0 ? 0 : (it0 > n_it ? n_it : it0);
    it1 = it1 < 0 ? 0 : (it1 > n_it ? n_it : it1);
    const CountKey* const kb = keys + base + g;
    int cnt = 0;
    auto walk = [&](auto narrow_tag) {
        auto below = [&](const CountKey& b, unsigned long long l) {
            if constexpr (decltype(narrow_tag)::value) count_below96(cnt, b, ka.h, l);
            else count_below(cnt, b, ka.h, l);
        };
        auto span = [&](int i, int end, unsigned long long l) {
            for (; i + 8 <= end; i += 8) {
                CountKey b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) b[u] = kb[4 * (i + u)];
#pragma unroll
                for (int u = 0; u < 8; ++u) below(b[u], l);
            }
            for (; i < end; ++i) below(kb[4 * i], l);
        };
        span(0, it0, ka.l + 1ull);   // candidates in front of the tile win ties
        span(it1, n_it, ka.l);
        if (own_here) {
            // the own tile (staged by this wave): candidates 4 j + g of it against a, ties by index
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int bb = 4 * j + g;
                below(keys[tile * 16 + bb], ka.l + (bb < a ? 1ull : 0ull));
            }
        }
    };
    if (narrow) walk(std::true_type{});
    else walk(std::false_type{});
#ifdef SX_STAMPS
    cs[3] = stamp();
#endif
    cnt += __shfl_xor(cnt, 16);
    cnt += __shfl_xor(cnt, 32);
    if (lane < 16) part[wave][lane] = cnt;
    __syncthreads();
    int r = 0;   // rank of candidate a2, summed by each of its 32 threads
#pragma unroll
    for (int w = 0; w < kCountWaves; ++w) r += part[w][a2];
    if (i2 >= P) r = 0x7fffffff;

#ifdef SX_STAMPS
    cs[4] = stamp();
#endif
    // ---- 3. the elites' rows, 32 threads per candidate ----
    if (r < k) {
        if (ra.elite_rows) {
            // device-scope stores (sc1: written through this XCD's L2): the refit below reads them from another
            // workgroup, on another XCD in general, without a full L2 write-back / invalidate per workgroup
            double* row = ra.elite_rows + ((long long)e * k + r) * W;
            if (c < W) __hip_atomic_store(row + c, row_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int col = c + 32; col < W; col += 32)
                __hip_atomic_store(row + col, act[(long long)i2 * ra.act_stride + (col - 2)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (c == 0 && ra.elite_idx) ra.elite_idx[(long long)e * k + r] = i2;
        if (r == 0) {
            if (ra.best) {
                if (c >= 2 && c < W) ra.best[(long long)e * L + c - 2] = row_v;
                for (int col = c + 32; col < W; col += 32) ra.best[(long long)e * L + col - 2] = act[(long long)i2 * ra.act_stride + (col - 2)];
            }
            if (c == 0 && ra.best_ok) ra.best_ok[e] = (row_v == 0.0) ? 1 : 0;
        }
    }
#ifdef SX_STAMPS
    cs[5] = stamp();
    if (g_stamp_buf && tid == 0 && tile == 0 && e == 0)
        for (int j = 0; j < 5; ++j) g_stamp_buf[16 + j] = cs[j + 1] - cs[j];
#endif
    if (!ra.mean) return;

    // ---- 4. refit by the workgroup that delivers last ----
    // The rows were stored at device scope (sc1: written through to the L2 all XCDs share).  Every wave first DRAINS its own
    // stores -- s_waitcnt vmcnt(0): they are acknowledged, i.e. visible at device scope -- and only then joins the barrier
    // behind which thread 0 takes the ticket.  (Round 2 relied on the barrier for that; gfx950's s_barrier does not wait for
    // outstanding stores -- the built code was store, s_barrier, atomic with no wait in between -- so the last workgroup
    // could in principle have read rows still in flight: ADVICE r2.)  The reader side pairs it with an acquire fence in the
    // one workgroup that takes the last ticket.  No __threadfence per thread: an L2 write-back + invalidate per wave,
    // 2048 of them, took 30 us of this kernel's first version at P = 4096.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last_flag = (atomicInc(tickets + e, gridDim.x - 1) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
#ifdef SX_STAMPS
    cs[6] = stamp();
#endif
    if (!last_flag) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (one workgroup per problem gets here)
    const double* rows = ra.elite_rows + (long long)e * k * W;   // written by this launch: device-scope loads (sc1)
    constexpr int kKeep = 8;
    for (int c0 = 0; c0 < L; c0 += 256) {
        const int Lc = (L - c0) < 256 ? (L - c0) : 256;
        const int R = kCountThreads / Lc;   // row groups
        const int c = tid % Lc, r0 = tid / Lc;
        const bool active = r0 < R;
        const double* const col = rows + 2 + c0 + c;
        auto fetch = [&](int first, double (&v)[kKeep]) {   // all loads of a batch in flight together; rows past k read row k - 1
#pragma unroll
            for (int j = 0; j < kKeep; ++j) {
                const int r = first + j * R;
                v[j] = __hip_atomic_load(col + (r < k ? r : k - 1) * W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int j = 0; j < kKeep; ++j)
                if (first + j * R >= k) v[j] = 0.0;
        };
        double keep[kKeep];
        double s = 0.0;
        if (active) {
            fetch(r0, keep);
#pragma unroll
            for (int j = 0; j < kKeep; ++j) s += keep[j];
            for (int first = r0 + kKeep * R; first < k; first += kKeep * R) {
                double v[kKeep];
                fetch(first, v);
#pragma unroll
                for (int j = 0; j < kKeep; ++j) s += v[j];
            }
        }
        red[tid] = s;
        __syncthreads();
        column_totals<kCountThreads>(red, Lc, R, tid, [&](int cc, double t) { col_mean[cc] = t / k; });
        __syncthreads();
        const double mu = col_mean[c];
        double ss = 0.0;
        if (active) {
#pragma unroll
            for (int j = 0; j < kKeep; ++j) {
                const double dv = keep[j] - mu;
                if (r0 + j * R < k) ss += dv * dv;
            }
            for (int first = r0 + kKeep * R; first < k; first += kKeep * R) {
                double v[kKeep];
                fetch(first, v);
#pragma unroll
                for (int j = 0; j < kKeep; ++j) {
                    const double dv = v[j] - mu;
                    if (first + j * R < k) ss += dv * dv;
                }
            }
        }
        red[tid] = ss;
        __syncthreads();
        column_totals<kCountThreads>(red, Lc, R, tid, [&](int cc, double t) {
            ra.mean[(long long)e * L + c0 + cc] = col_mean[cc];
            if (ra.std) ra.std[(long long)e * L + c0 + cc] = (k > 1) ? sqrt(t / (k - 1)) : 0.0;
        });
        __syncthreads();
    }
#ifdef SX_STAMPS
    cs[7] = stamp();
    if (g_stamp_buf && tid == 0 && e == 0) {
        g_stamp_buf[21] = cs[6] - cs[5];   // barrier + ticket, in the workgroup that delivered last
        g_stamp_buf[22] = cs[7] - cs[6];   // refit
        g_stamp_buf[23] = cs[7] - cs[0];   // the last workgroup's whole life
    }
#endif
}

}  // namespace sx
