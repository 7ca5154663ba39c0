// The fused CEM particle rollout, third form (round 3): EIGHT waves with the GP's triangular factors PARTLY resident.
//
// What the first two forms taught (profiles/r03_*):
//   * cem_rollout_kernel (sx_rollout.hpp; 8 waves, W streamed from L2 by a table-driven stage loop): the matrix phase costs
//     14.2k cycles per step against 11.7k of pure MFMA issue -- per-row-block epilogues on a single accumulator tuple,
//     descriptor loads, four rotating register sets.  Its VALU phase is fine: two waves per SIMD hide each other's LDS
//     round trips and pipeline fill.
//   * cem_rollout_rw_kernel (sx_rollout_rw.hpp; 4 waves x 512 registers, all of W resident): the matrix phase is
//     straight-line code, one accumulator per row-block, 12.8k cycles.  But a LONE wave per SIMD issues an f64 VALU
//     instruction every 5.8 cycles instead of 5.1 and pays every pipeline fill itself: the Kstar / finish() phase grew from
//     4.3k to 5.1k cycles and ate the gain.
// This kernel keeps two waves per SIMD for the VALU phases AND the static matrix phase.  Eight waves have 256 registers
// each; a wave's share of W (<= 48 fragment pairs at n_s = 2, N = 200) is consumed in a fixed order, pair-major, and by
// position in that order a pair lives
//     [0, REGP)              in registers, for the whole launch (the phase opens on them: nothing to wait for),
//     [REGP, REGP + LDSP)    in LDS, next to Kstar (read like a Kstar fragment, two pairs ahead),
//     the rest               in L2, requested PFG pairs ahead of use into rotating registers (in flight while the resident
//                            pairs multiply: a quarter of cem_rollout_kernel's L2 traffic).
// The instruction stream of a wave's matrix phase is generated at compile time as a PROGRAM (RhProgram: load Kstar pair,
// load W pair from LDS / L2, two MFMAs, epilogue) and emitted by a compile-time loop: no descriptor, no address
// arithmetic, no branch, one accumulator tuple per row-block.
// The VALU phases are the register-resident kernel's: the diet Kstar phase (rw_kstar_phase), finish() with its constants
// in LDS and the polytope rows on four lane groups, per-particle state in LDS.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sx_amd.h"
#include "sx_rollout_rw.hpp"

#ifndef SX_RH_REGP
#define SX_RH_REGP 26      // W fragment pairs per wave kept in registers (104 of the wave's 256)
#endif
#ifndef SX_RH_LEAD
// Where the L2-streamed part of a wave's stream sits in its program: behind this many register-resident pairs, and BEFORE the
// LDS and the other register pairs -- so the phase's tail, which one wave of a SIMD runs alone, needs no loads at all
// (-1: at the end of the program, as first built).
#define SX_RH_LEAD -1
#endif
#ifndef SX_RH_LDSP
#define SX_RH_LDSP 8       // W fragment pairs per wave kept in LDS (8 waves x 8 KB)
#endif
#ifndef SX_RH_PFG
#define SX_RH_PFG 6        // streamed pairs in flight ahead of their MFMAs (L2 round trip ~ 6 x 128 cycles)
#endif
#ifndef SX_RH_PFL
#define SX_RH_PFL 2        // LDS-resident pairs requested ahead of their MFMAs
#endif
#ifndef SX_RH_RBMAJOR
#define SX_RH_RBMAJOR 1    // matrix phase row-block by row-block (unbroken accumulator chains) instead of pair-major
#endif
#ifndef SX_RH_PRIO
#define SX_RH_PRIO 0       // matrix phase: waves 0 .. 3 at raised priority (measured: 126.1 against 125.3 us without -- off)
#endif
#ifndef SX_RH_WFIRST
#define SX_RH_WFIRST 0     // 1: request the resident W before anything else (the first version; A/B switch)
#endif
#ifndef SX_RH_SHARES
#define SX_RH_SHARES 4, 4, 4, 1, 4, 4, 4   // Kstar phase: relative shares of waves 1 .. 7 (25 pairs of 8 rows at N = 200; wave 4 also runs finish_costs)
#endif

namespace sx {

constexpr int kRhWaves = 8;
constexpr int kRhThreads = 64 * kRhWaves;

// Static assignment of the (output d, row-block rb) tasks to the 8 waves.  Waves w and w + 4 share SIMD w, and a phase ends
// when a SIMD's TOTAL is done, so the sets of one output go to its wave group in descending order of size and the next
// output's in ascending order: heavy meets light on a SIMD.
template <int NS, int NRB>
struct RhPlan {
    int wave_of[NS][NRB];
    int pairs[kRhWaves];
    constexpr RhPlan() : wave_of{}, pairs{} {
        for (int w = 0; w < kRhWaves; ++w) pairs[w] = 0;
        for (int d = 0; d < NS; ++d) {
            int lo = 0, n = kRhWaves;
            if (kRhWaves % NS == 0) {
                n = kRhWaves / NS;
                lo = d * n;
            }
            // LPT into n sets (on top of what the waves already hold when the outputs share all waves); set s goes to
            // wave lo + s, in reverse for every other output of a grouped plan
            const bool rev = (kRhWaves % NS == 0) && (d & 1);
            int set_of[NRB] = {};
            int load[kRhWaves] = {};
            for (int rb = NRB - 1; rb >= 0; --rb) {
                int best = 0;
                for (int s = 1; s < n; ++s)
                    if (load[s] + pairs[lo + (rev ? n - 1 - s : s)] < load[best] + pairs[lo + (rev ? n - 1 - best : best)]) best = s;
                set_of[rb] = best;
                load[best] += 2 * (rb + 1);
            }
            for (int rb = 0; rb < NRB; ++rb) wave_of[d][rb] = lo + (rev ? n - 1 - set_of[rb] : set_of[rb]);
            for (int s = 0; s < n; ++s) pairs[lo + (rev ? n - 1 - s : s)] += load[s];
        }
    }
    constexpr int max_pairs() const {
        int m = 0;
        for (int w = 0; w < kRhWaves; ++w) m = pairs[w] > m ? pairs[w] : m;
        return m;
    }
    constexpr bool has(int w, int d) const {
        for (int rb = 0; rb < NRB; ++rb)
            if (wave_of[d][rb] == w) return true;
        return false;
    }
    constexpr int qmax(int w, int d) const {
        for (int rb = NRB - 1; rb >= 0; --rb)
            if (wave_of[d][rb] == w) return 2 * (rb + 1);
        return 0;
    }
};

// A wave's MFMA pairs (d, rb, q) in CONSUMPTION order.  SX_RH_RBMAJOR: row-block by row-block -- one accumulator tuple is
// fed by an unbroken chain of dependent MFMAs, which issue every 64.5 cycles; pair-major order (every Kstar pair read once,
// the wave's row-blocks taking turns) switches the accumulator after every second MFMA and costs 67.3 cycles per MFMA
// (tools/mfma_probe3.hip: chain / pairs7).  The price is one LDS read of the Kstar pair per MFMA pair instead of one per
// pair and wave: 46 instead of 26 ds_read_b128 per wave and step, far from the LDS's limit.
template <int NS, int NRB>
struct RhElements {
    static constexpr int kMax = NS * NRB * (NRB + 1);
    short d[kRhWaves][kMax], rb[kRhWaves][kMax], q[kRhWaves][kMax];
    int n[kRhWaves];
    constexpr RhElements() : d{}, rb{}, q{}, n{} {
        const RhPlan<NS, NRB> plan{};
        for (int w = 0; w < kRhWaves; ++w) {
            int idx = 0;
            for (int dd = 0; dd < NS; ++dd) {
#if SX_RH_RBMAJOR
                for (int r = NRB - 1; r >= 0; --r)
                    for (int qq = 0; qq < 2 * (r + 1); ++qq)
                        if (plan.wave_of[dd][r] == w) { d[w][idx] = dd; rb[w][idx] = r; q[w][idx] = qq; ++idx; }
#else
                for (int qq = 0; qq < 2 * NRB; ++qq)
                    for (int r = NRB - 1; r >= 0; --r)
                        if (plan.wave_of[dd][r] == w && qq < 2 * (r + 1)) { d[w][idx] = dd; rb[w][idx] = r; q[w][idx] = qq; ++idx; }
#endif
            }
            n[w] = idx;
        }
    }
};

// Where the A operand of a wave's e-th MFMA pair (consumption order) is kept.  STORAGE index s: s < REGP a register pair,
// REGP <= s < REGP + LDSP a pair in the wave's LDS block, above that a pair streamed from L2 every step.  With `total`
// pairs in the wave's stream: registers take min(REGP, total), LDS the next min(LDSP, rest), L2 what is left; consumption
// order is [lead register pairs][L2][LDS][the other register pairs] (SX_RH_LEAD), or [registers][LDS][L2] (lead < 0).
struct RhLayout {
    int nr, nl, ng, lead;
    constexpr RhLayout(int total, int regp, int ldsp)
        : nr(total < regp ? total : regp), nl(0), ng(0), lead(0) {
        nl = total - nr < ldsp ? total - nr : ldsp;
        ng = total - nr - nl;
        lead = SX_RH_LEAD < 0 || ng == 0 ? nr : (SX_RH_LEAD < nr ? SX_RH_LEAD : nr);
    }
    constexpr int storage(int e, int regp, int ldsp) const {
        if (SX_RH_LEAD < 0 || ng == 0) return e;                              // [registers][LDS][L2]: storage order
        if (e < lead) return e;
        if (e < lead + ng) return regp + ldsp + (e - lead);
        if (e < lead + ng + nl) return regp + (e - lead - ng);
        return lead + (e - lead - ng - nl);
    }
};

// pair offsets into a_pack by STORAGE index; entries past the share repeat pair 0 (a valid address, never used)
template <int NS, int NRB, int MAXP, int REGP, int LDSP>
struct RhStream {
    int pair[kRhWaves][MAXP];
    constexpr RhStream() : pair{} {
        const RhElements<NS, NRB> el{};
        constexpr int wpo = NRB * (NRB + 1);
        for (int w = 0; w < kRhWaves; ++w) {
            const RhLayout lay(el.n[w], REGP, LDSP);
            for (int i = 0; i < MAXP; ++i) pair[w][i] = 0;
            for (int e = 0; e < el.n[w]; ++e)
                pair[w][lay.storage(e, REGP, LDSP)] = el.d[w][e] * wpo + el.rb[w][e] * (el.rb[w][e] + 1) + el.q[w][e];
        }
    }
};

// The matrix phase of one wave as a program.
enum RhOpKind { kRhLoadB = 0, kRhLoadWL = 1, kRhLoadWG = 2, kRhMfma = 3, kRhRowEpi = 4, kRhFinalEpi = 5 };
struct RhOp {
    int kind;
    int i;      // storage index of the W pair (W loads, MFMA)
    int rb;     // row-block (MFMA, row epilogue)
    int q;      // Kstar pair (Kstar load, MFMA)
    int d;      // output
    int slot;   // Kstar register slot (Kstar load, MFMA)
    int e;      // position in the wave's consumption order (MFMA)
};
template <int NS, int NRB, int WAVE, int REGP, int LDSP, int PF, int PFL, int PFG>
struct RhProgram {
    static constexpr int kMaxOps = 4 * NS * NRB * (NRB + 1) / 2 + 8 * NRB * NS + 16;
    RhOp ops[kMaxOps];
    int n;
    int total;   // pairs in this wave's stream
    constexpr RhProgram() : ops{}, n(0), total(0) {
        const RhElements<NS, NRB> el{};
        total = el.n[WAVE];
        const RhLayout lay(total, REGP, LDSP);
        int next_l = 0;                  // next LDS-resident pair to request (count within the LDS part)
        int next_g = 0;                  // next streamed pair to request (count within the L2 part)
        const int e_g0 = (SX_RH_LEAD < 0 || lay.ng == 0) ? lay.nr + lay.nl : lay.lead;             // element of the first L2 pair
        const int e_l0 = (SX_RH_LEAD < 0 || lay.ng == 0) ? lay.nr : lay.lead + lay.ng;             // ... first LDS pair
        // Kstar loads: one per MFMA pair (row-block-major) or one per (d, q) group (pair-major); `bl` = loads issued so far
        // in element order, each element knows the slot its Kstar pair sits in
        int bslot[RhElements<NS, NRB>::kMax] = {};
        int bfirst[RhElements<NS, NRB>::kMax] = {};   // 1 if the element opens a new Kstar load
        int nloads = 0;
        for (int e = 0; e < total; ++e) {
            const bool fresh = SX_RH_RBMAJOR || e == 0 || el.q[WAVE][e] != el.q[WAVE][e - 1] || el.d[WAVE][e] != el.d[WAVE][e - 1];
            if (fresh) ++nloads;
            bfirst[e] = fresh ? 1 : 0;
            bslot[e] = (nloads - 1) % (PF + 1);
        }
        // pending row epilogues: emitted kDelay MFMA pairs after the row-block's last pair (its accumulator has drained by
        // then), or at the end
        constexpr int kDelay = 2;
        int pend_rb[NRB * NS] = {}, pend_d[NRB * NS] = {}, pend_at[NRB * NS] = {};
        int npend = 0, done = 0;
        int issued_b = 0;   // Kstar loads issued (in element order of their first users)
        int eb = 0;         // next element whose Kstar load is to be issued
        auto issue_b_upto = [&](int loads_wanted) {
            while (issued_b < loads_wanted && eb < total) {
                if (bfirst[eb]) {
                    ops[n++] = RhOp{kRhLoadB, 0, 0, el.q[WAVE][eb], el.d[WAVE][eb], bslot[eb]};
                    ++issued_b;
                }
                ++eb;
            }
        };
        int loads_seen = 0;
        for (int e = 0; e < total; ++e) {
            if (bfirst[e]) ++loads_seen;
            issue_b_upto(loads_seen + PF);
            while (next_l < lay.nl && e_l0 + next_l <= e + PFL) ops[n++] = RhOp{kRhLoadWL, REGP + next_l++, 0, 0, 0, 0};
            while (next_g < lay.ng && e_g0 + next_g <= e + PFG) ops[n++] = RhOp{kRhLoadWG, REGP + LDSP + next_g++, 0, 0, 0, 0};
            ops[n++] = RhOp{kRhMfma, lay.storage(e, REGP, LDSP), el.rb[WAVE][e], el.q[WAVE][e], el.d[WAVE][e], bslot[e], e};
            const bool last_of_rb = el.q[WAVE][e] == 2 * (el.rb[WAVE][e] + 1) - 1;
            if (last_of_rb) {
                pend_rb[npend] = el.rb[WAVE][e];
                pend_d[npend] = el.d[WAVE][e];
                pend_at[npend] = e + kDelay;
                ++npend;
            }
            while (done < npend && pend_at[done] <= e) {
                ops[n++] = RhOp{kRhRowEpi, 0, pend_rb[done], 0, pend_d[done], 0};
                ++done;
            }
        }
        while (done < npend) {
            ops[n++] = RhOp{kRhRowEpi, 0, pend_rb[done], 0, pend_d[done], 0};
            ++done;
        }
        const RhPlan<NS, NRB> plan{};
        for (int d = 0; d < NS; ++d)
            if (plan.has(WAVE, d)) ops[n++] = RhOp{kRhFinalEpi, 0, 0, 0, d, 0};
    }
};

template <int NS, int D, int NRB, int WAVE, int REGP, int LDSP>
__device__ __forceinline__ void rh_mfma_phase(const GpConst<NS, D>& gc, GpTileLds<NS, D>& lds, const v2d* wlds_wave, int lane,
                                              const v2d (&wreg)[REGP]) {
    constexpr int PF = SX_RW_PF, PFL = SX_RH_PFL, PFG = SX_RH_PFG;
    constexpr int MAXP = RhPlan<NS, NRB>{}.max_pairs();
    static constexpr RhProgram<NS, NRB, WAVE, REGP, LDSP, PF, PFL, PFG> prog{};
    static constexpr RhStream<NS, NRB, MAXP, REGP, LDSP> stream{};
    const v2d* kbase = reinterpret_cast<const v2d*>(lds.kfrag) + lane;
    // (the streamed pairs are the same every step: hidden behind an opaque copy of the pointer, or the compiler hoists
    // their loads out of the step loop and spills what it hoisted)
    // (explicit address spaces: behind the opaque copy the compiler no longer knows that this is global memory, and a
    // FLAT load counts on both wait counters -- every wait for a Kstar fragment would wait for the L2 as well)
    typedef const __attribute__((address_space(1))) v2d* gv2d;
    typedef const __attribute__((address_space(3))) v2d* lv2d;
    const v2d* ap_opaque = reinterpret_cast<const v2d*>(gc.a_pack) + lane;
    asm volatile("" : "+v"(ap_opaque));
    const gv2d ap = (gv2d)ap_opaque;
    const lv2d wl_base = (lv2d)wlds_wave + lane;
    v4d acc[NRB];
    v2d b[PF + 1];
    v2d wl[PFL + 1];
    v2d wg[PFG + 1];
    double ssq[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) ssq[d] = 0.0;
    static_for<0, prog.n>([&](auto ktag) {
        constexpr RhOp op = prog.ops[decltype(ktag)::value];
        if constexpr (op.kind == kRhLoadB) {
            b[op.slot] = kbase[(op.q * NS + op.d) * 64];
        } else if constexpr (op.kind == kRhLoadWL) {
            wl[op.i % (PFL + 1)] = wl_base[(op.i - REGP) * 64];
        } else if constexpr (op.kind == kRhLoadWG) {
            wg[op.i % (PFG + 1)] = ap[(size_t)stream.pair[WAVE][op.i] * 64];
        } else if constexpr (op.kind == kRhMfma) {
#if SX_RH_PRIO == 2
            // progress-based priority: a wave that is behind its SIMD partner (in quarters of its own program) wins the pipe
            if constexpr ((4 * op.e) / prog.total != (4 * (op.e - 1)) / prog.total || op.e == 0)
                __builtin_amdgcn_s_setprio(3 - (4 * op.e) / prog.total);
#endif
            SX_PIN();
            const v2d bq = b[op.slot];
            v2d a;
            if constexpr (op.i < REGP)
                a = wreg[op.i];
            else if constexpr (op.i < REGP + LDSP)
                a = wl[op.i % (PFL + 1)];
            else
                a = wg[op.i % (PFG + 1)];
            if constexpr (op.q == 0)
                acc[op.rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bq.x, v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
            else
                acc[op.rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bq.x, acc[op.rb], 0, 0, 0);
            acc[op.rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bq.y, acc[op.rb], 0, 0, 0);
            SX_PIN();
        } else if constexpr (op.kind == kRhRowEpi) {
            constexpr int rb = op.rb, d = op.d;
            double s;
            if constexpr (rb >= NRB - 2) {
                // rows >= N live in the last row-block(s): N .. N + D are the mean / Jacobian rows, above is padding
                const int row0 = rb * 16 + (lane >> 4);
                s = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 4 * r;
                    if (row < gc.n_train)
                        s = fma(acc[rb][r], acc[rb][r], s);
                    else if (row - gc.n_train <= D)
                        lds.mj[d * 256 + (row - gc.n_train) * 16 + (lane & 15)] = acc[rb][r];
                }
            } else {
                s = fma(acc[rb][3], acc[rb][3], fma(acc[rb][2], acc[rb][2], fma(acc[rb][1], acc[rb][1], acc[rb][0] * acc[rb][0])));
            }
            ssq[d] += s;
        } else {
            constexpr int d = op.d;
            // the four lanes l, l ^ 16, l ^ 32, l ^ 48 hold the partial sums of one query point: ones(16 x 4) . B totals them
            const v4d tot = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, ssq[d], v4d{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
            if (lane < 16) lds.part[(WAVE * NS + d) * 16 + lane] = tot[0];
        }
    });
}

// gp_collect over the waves that own row-blocks of output d
template <int NS, int D, int NRB, bool WITH_JAC, typename G>
__device__ __forceinline__ void rh_collect(const G& gc, const GpTileLds<NS, D>& lds, int c, const double (&z)[D],
                                           double (&mean)[NS], double (&var)[NS], double (&jac)[NS][D]) {
    constexpr RhPlan<NS, NRB> plan{};
#pragma unroll
    for (int d = 0; d < NS; ++d) {
        double q = 0.0;
#pragma unroll
        for (int w = 0; w < kRhWaves; ++w)
            if (plan.has(w, d)) q += lds.part[(w * NS + d) * 16 + c];
        var[d] = (gc.outputscale[d] - q) + gc.noise[d];
        const double m = lds.mj[d * 256 + c];
        mean[d] = m;
        if constexpr (WITH_JAC) {
#pragma unroll
            for (int j = 0; j < D; ++j) jac[d][j] = lds.mj[d * 256 + (1 + j) * 16 + c] - z[j] * gc.inv_ls2[d * D + j] * m;
        }
    }
}

template <int NS, int NRB>
constexpr int rh_regp() {
    constexpr int m = RhPlan<NS, NRB>{}.max_pairs();
    return m < SX_RH_REGP ? m : SX_RH_REGP;
}
template <int NS, int NRB>
constexpr int rh_ldsp() {
    constexpr int rest = RhPlan<NS, NRB>{}.max_pairs() - rh_regp<NS, NRB>();
    return rest < SX_RH_LDSP ? rest : SX_RH_LDSP;
}
// dynamic LDS of the kernel, in doubles (+ sizeof(RwConst) bytes)
template <int NS, int NU, int NRB>
inline size_t rh_lds_doubles(int n_train, int n_pad, int H) {
    constexpr int D = NS + NU, S = NS + NS * NS;
    return (size_t)NS * n_pad * 16 + NS * 256 + (size_t)kRhWaves * NS * 16 + 32 * D + (((size_t)SX_TILE * H * NU + 1) & ~(size_t)1) +
           (((size_t)SX_TILE * (2 * (S + NS) + 3) + 1) & ~(size_t)1) + RwKstarLds<NS, D>::doubles(n_pad) +
           (size_t)kRhWaves * rh_ldsp<NS, NRB>() * 128;
}

template <int NS, int NU, int NRB>
__global__ __launch_bounds__(kRhThreads) void cem_rollout_rh_kernel(GpConst<NS, NS + NU> gc, ReachConst<NS, NU> rc,
                                                                    CostConst<SX_MAX_M, NS, NU> cc, RolloutPtrs rp) {
    constexpr int D = NS + NU;
    constexpr int S = NS + NS * NS;
    constexpr int PQS = S + NS;   // a particle's state row in LDS: p, Q (row-major), var of the step that produced them
    constexpr RhPlan<NS, NRB> plan{};
    constexpr int MAXP = plan.max_pairs();
    constexpr int REGP = rh_regp<NS, NRB>();
    constexpr int LDSP = rh_ldsp<NS, NRB>();
    extern __shared__ __attribute__((aligned(16))) double smem[];
    GpTileLds<NS, D> lds;
    constexpr int nw = kRhWaves;
    // (GpTileLds without its X rows and its 256-entry exp table: the Kstar phase here has its own, RwKstarLds)
    lds.xs = nullptr;
    lds.etab = nullptr;
    lds.kfrag = smem;
    lds.mj = lds.kfrag + (size_t)NS * gc.n_pad * 16;
    lds.part = lds.mj + NS * 256;
    lds.zs = lds.part + (size_t)nw * NS * 16;
    double* acts = lds.zs + 32 * D;  // [16][H][NU]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int H = rp.H;
    const int tiles_per_problem = (rp.P + SX_TILE - 1) / SX_TILE;
    const int total_tiles = rp.E * tiles_per_problem;

    // LDS behind the action table: finish()'s constants, the per-particle state, the Kstar phase's rows and table, W overflow
    RwConst<NS, NU>* const cst = reinterpret_cast<RwConst<NS, NU>*>(acts + (((size_t)SX_TILE * rp.H * NU + 1) & ~(size_t)1));
    double* const pq = reinterpret_cast<double*>(cst + 1);
    RwKstarLds<NS, D> kl;
    kl.carve(pq + (((size_t)SX_TILE * (2 * PQS + 3) + 1) & ~(size_t)1), gc.n_pad);   // 2 state buffers + the cost rows
    // W overflow, [8 waves][LDSP][64 lanes] of 16 bytes (every piece before it is a multiple of 16 bytes long)
    v2d* const wlds = reinterpret_cast<v2d*>(kl.xbar + ((D + 1) & ~1));

    // This wave's resident share of W (stream positions [0, REGP) to registers, [REGP, REGP + LDSP) to LDS) is requested
    // BEHIND the first tile's prologue (below, `first_tile`): loads return in order, so with the 272 KB of W per workgroup
    // requested first -- as the first version did -- everything the prologue loads (table, X rows, elite rows, noise) waited
    // behind it, ~2.7 us of a launch's ~10 us of fixed cost (tools/horizon_sweep.sh).  Requested last, W travels under the
    // first Kstar phase; the LDS part goes by DMA (buffer_load ... lds: no staging registers), awaited before the first
    // matrix phase's barrier.
    static constexpr RhStream<NS, NRB, MAXP, REGP, LDSP> stream{};
    v2d wreg[REGP];
#if SX_RH_WFIRST   // (A/B switch: the first version's order)
    {
        const v2d* __restrict__ const ap0 = reinterpret_cast<const v2d*>(gc.a_pack) + lane;
#pragma unroll
        for (int i = 0; i < REGP; ++i) wreg[i] = ap0[(size_t)stream.pair[wave][i] * 64];
        if constexpr (LDSP > 0) {
#pragma unroll
            for (int i = 0; i < LDSP; ++i) wlds[(wave * LDSP + i) * 64 + lane] = ap0[(size_t)stream.pair[wave][REGP + i] * 64];
        }
    }
#endif
    {
        static_assert(sizeof(RwConst<NS, NU>) % 8 == 0 && sizeof(ReachConst<NS, NU>) % 8 == 0 &&
                      sizeof(CostConst<SX_MAX_M, NS, NU>) % 8 == 0, "copied in 8-byte words");
        constexpr int nrc = sizeof(ReachConst<NS, NU>) / 8, ncc = sizeof(CostConst<SX_MAX_M, NS, NU>) / 8;
        double* dst = reinterpret_cast<double*>(cst);
        const double* src_rc = reinterpret_cast<const double*>(&rc);
        const double* src_cc = reinterpret_cast<const double*>(&cc);
        for (int i = tid; i < nrc; i += kRhThreads) dst[i] = src_rc[i];
        for (int i = tid; i < ncc; i += kRhThreads) dst[nrc + i] = src_cc[i];
        if (tid < NS * D) cst->inv_ls2[tid] = gc.inv_ls2[tid];
        if (tid < NS) {
            cst->outputscale[tid] = gc.outputscale[tid];
            cst->noise[tid] = gc.noise[tid];
        }
    }
    const RwConst<NS, NU>& fc = *cst;
    const ReachConst<NS, NU>& frc = cst->rc;
    const CostConst<SX_MAX_M, NS, NU>& fcc = cst->cc;
    rw_kstar_setup(gc, kl, lds.kfrag);
    const int kstar_pairs = (gc.n_train + 7) >> 3;   // pairs of 8 rows with at least one training row

    const bool owner = tid < SX_TILE;
    // Kstar shares (pairs of fragments).  Step 0: all waves alike.  From step 1 on wave 0 runs finish(); waves w and w + 4
    // share a SIMD, so wave 4 competes with finish() for its pipe and gets a smaller share.
    int q0_begin, q0_end, q_begin = 0, q_end = 0;
    kstar_pair_range(kstar_pairs, wave, 1, nw, q0_begin, q0_end);
    if (wave > 0) {
        // shares of waves 1 .. 7 (wave 0 runs finish_state()).  Waves w and w + 4 share SIMD w and the older wave wins the
        // arbitration, so a SIMD is done when the SUM of its two waves' work is: SIMD 0 has finish_state() (latency-bound,
        // ~3.5k cycles) beside wave 4 (finish_costs() + its share), SIMDs 1 .. 3 two Kstar waves each.
        constexpr int kShare[8] = {0, SX_RH_SHARES};
        int before = 0, total = 0;
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            if (w < wave) before += kShare[w];
            total += kShare[w];
        }
        kstar_pair_range(kstar_pairs, before, kShare[wave], total, q_begin, q_end);
    }
    double* const zs_base = lds.zs;

    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
#ifdef SX_STAMPS
        const unsigned long long tile_t0 = stamp();
        unsigned long long first_k = 0, first_kb = 0;
#endif
        const int e = tile / tiles_per_problem;
        const int c0 = (tile - e * tiles_per_problem) * SX_TILE;  // first particle of the tile within problem e
        const bool valid = owner && (c0 + tid < rp.P);

        // ---- the sampling distribution and this tile's action sequences (as cem_rollout_kernel's prologue) ----
        const double* dist_mean = rp.mean + (int64_t)e * H * NU;
        const double* dist_std = rp.std + (int64_t)e * H * NU;
        // (requesting the first noise draw or the start state here, ahead of the refit, as cem_rollout_kernel does: 12 to 23
        // spilled dwords in the step loop and 1.2 % slower; not done)
        bool have_q = rp.q0 != nullptr;
        if (rp.elite_rows) {
            const int L = H * NU, W = 2 + L;
            double* const ms = lds.kfrag;   // [2][L]: the Kstar buffer is free until the first step
            const double* rows = rp.elite_rows + (int64_t)e * rp.elite_k * W + 2;
            const bool publish = c0 == 0 && rp.mean_out;
            int cshift = 0;
            while ((nw << cshift) < L && cshift < 6) ++cshift;
            const int ccol = lane & ((1 << cshift) - 1);
            for (int cb = wave << cshift; cb < L; cb += nw << cshift) {
                const int col = cb + ccol;
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
        for (int i = tid; i < SX_TILE * H * NU; i += kRhThreads) {
            const int c = i / (H * NU);
            const int r = i - c * (H * NU);
            double a = 0.0;
            if (c0 + c < rp.P) {
                const int64_t gi = ((int64_t)e * rp.P + c0 + c) * (H * NU) + r;
                if (rp.noise) {
                    a = dist_mean[r] + dist_std[r] * rp.noise[gi];
                    rp.actions[gi] = a;
                } else {
                    a = rp.actions[gi];
                }
            }
            acts[i] = a;
        }
        // Per-particle state in LDS.  st[b][c] = (p, Q, var) -- two buffers: the state after step s (p_{s+1}, Q_{s+1}, with the
        // predictive variance of step s) lives in buffer (s + 1) & 1, so the wave that advances the state and the wave that
        // prices the previous step never touch the same row in one phase.  ac[c] = (objective cost, constraint cost, status).
        const bool lane_owner = lane < SX_TILE;                 // (waves 0 and 4 both run per-particle code on lanes 0..15)
        const bool lane_valid = lane_owner && (c0 + lane < rp.P);
        auto st_row = [&](int buf, int c) { return pq + (buf * SX_TILE + c) * PQS; };
        double* const ac_row = pq + 2 * SX_TILE * PQS + (lane_owner ? lane : 0) * 3;
        if (owner) {
            double* my = st_row(0, tid);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                my[i] = rp.x0[(int64_t)e * NS + i];
#pragma unroll
                for (int j = 0; j < NS; ++j) my[NS + i * NS + j] = have_q ? rp.q0[((int64_t)e * NS + i) * NS + j] : 0.0;
            }
            ac_row[0] = 0.0;   // objective cost
            ac_row[1] = 0.0;   // constraint cost
            reinterpret_cast<int*>(ac_row + 2)[0] = 0;   // status bits
        }
        __syncthreads();
        if (owner) {
            const double* my = st_row(0, tid);
#pragma unroll
            for (int i = 0; i < NS; ++i) lds.zs[tid * D + i] = my[i];
#pragma unroll
            for (int cidx = 0; cidx < NU; ++cidx) lds.zs[tid * D + NS + cidx] = acts[(tid * H + 0) * NU + cidx];
        }
        __syncthreads();

        // The first tile of this workgroup requests the resident W (272 KB per workgroup through a vector memory path that
        // takes 64 bytes per cycle: 4.3k cycles).  Letting waves 4 .. 7 run their share of the first Kstar phase first and
        // request theirs behind it changed nothing measurable (121.3-121.7 against 121.5-121.7 us, one box).
        if (!SX_RH_WFIRST && tile == (int)blockIdx.x) {
            const v2d* __restrict__ const ap = reinterpret_cast<const v2d*>(gc.a_pack) + lane;
            if constexpr (LDSP > 0) {
                const __amdgpu_buffer_rsrc_t arsrc =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(gc.a_pack), 0, (int)0xffffffffu, 0x00020000);
#pragma unroll
                for (int i = 0; i < LDSP; ++i) {
                    const int pair = __builtin_amdgcn_readfirstlane(stream.pair[wave][REGP + i]);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        arsrc, (__attribute__((address_space(3))) void*)(wlds + (wave * LDSP + i) * 64), 16, lane * 16, pair << 10, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < REGP; ++i) wreg[i] = ap[(size_t)stream.pair[wave][i] * 64];
        }

        // centre of particle c at step t >= 1 from z_{t-1} and the means of step t - 1 (cem_rollout_kernel's chain, bit for bit)
        auto next_centre = [&](const ReachConst<NS, NU>& R, int c, const double* z_prev, double (&out)[NS]) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                double s = lds.mj[i * 256 + c];
#pragma unroll
                for (int j = 0; j < NS; ++j) s = fma(R.a[i * NS + j], z_prev[j], s);
#pragma unroll
                for (int cidx = 0; cidx < NU; ++cidx) s = fma(R.b[i * NU + cidx], z_prev[NS + cidx], s);
                out[i] = s;
            }
        };
        // finish(), first half -- the part the NEXT step depends on (lanes 0..15 of wave 0, during the Kstar phase of step
        // t + 1): posterior assembly, one-step reachability, the next query point; (p_{t+1}, Q_{t+1}, var_t) to LDS.
        // Everything that only prices the step -- state constraint, objective, action box, trajectory / sigma output --
        // is finish_costs(t), run ONE STEP LATER by wave 4 beside finish_state(t + 1): it is ~a quarter of finish()'s
        // dependent chain (the polytope rows end in a sqrt each), and finish() on wave 0 is what the phase waits for.
        auto finish_state = [&](int t) {
            double z[D], u[NU], mean[NS], var[NS], jac[NS][D], p1[NS], Q1[NS][NS];
            double p[NS], Q[NS][NS];
            // Everything this call reads from LDS is REQUESTED FIRST, in one batch: left to itself the compiler strings the
            // loads along the arithmetic (read, wait, use, read, ...), a dozen exposed LDS round trips of ~130 cycles at the
            // head of the one dependent chain the whole phase waits for.
            constexpr RhPlan<NS, NRB> plan{};
            const double* my = st_row(t & 1, lane);
            double partv[NS][kRhWaves], mjv[NS][1 + D], ils2[NS][D], osc[NS], nzv[NS];
#pragma unroll
            for (int d = 0; d < NS; ++d) {
#pragma unroll
                for (int w = 0; w < kRhWaves; ++w) partv[d][w] = plan.has(w, d) ? lds.part[(w * NS + d) * 16 + lane] : 0.0;
#pragma unroll
                for (int r = 0; r < 1 + D; ++r) mjv[d][r] = lds.mj[d * 256 + r * 16 + lane];
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                p[i] = my[i];
#pragma unroll
                for (int j = 0; j < NS; ++j) Q[i][j] = my[NS + i * NS + j];
            }
#pragma unroll
            for (int cidx = 0; cidx < NU; ++cidx) u[cidx] = acts[(lane * H + t) * NU + cidx];
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                osc[d] = fc.outputscale[d];
                nzv[d] = fc.noise[d];
#pragma unroll
                for (int j = 0; j < D; ++j) ils2[d][j] = fc.inv_ls2[d * D + j];
            }
            SX_PIN();
            const ReachConst<NS, NU>& rcv = frc;   // (read where used: a second batch of 17 doubles does not fit the registers)
#pragma unroll
            for (int j = 0; j < NS; ++j) z[j] = p[j];
#pragma unroll
            for (int cidx = 0; cidx < NU; ++cidx) z[NS + cidx] = u[cidx];
            // posterior of step t (gp_collect): variance with the likelihood noise, mean, mean Jacobian
#pragma unroll
            for (int d = 0; d < NS; ++d) {
                double q = 0.0;
#pragma unroll
                for (int w = 0; w < kRhWaves; ++w)
                    if (plan.has(w, d)) q += partv[d][w];
                var[d] = (osc[d] - q) + nzv[d];
                mean[d] = mjv[d][0];
#pragma unroll
                for (int j = 0; j < D; ++j) jac[d][j] = mjv[d][1 + j] - z[j] * ils2[d][j] * mean[d];
            }
            int st_step = 0;
            if (have_q) {
                reach_ellipsoid<NS, NU>(rcv, p, Q, u, mean, var, jac, p1, Q1, st_step);
            } else {
                reach_point<NS, NU>(rcv, p, u, mean, var, p1, Q1, st_step);
            }
            have_q = true;
            {
                double zt[D];
#pragma unroll
                for (int j = 0; j < NS; ++j) zt[j] = p[j];
#pragma unroll
                for (int cidx = 0; cidx < NU; ++cidx) zt[NS + cidx] = u[cidx];
                // (next_centre's chain on the means loaded above: the same values, the same operations)
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    double sc = mean[i];
#pragma unroll
                    for (int j = 0; j < NS; ++j) sc = fma(rcv.a[i * NS + j], zt[j], sc);
#pragma unroll
                    for (int cidx = 0; cidx < NU; ++cidx) sc = fma(rcv.b[i * NU + cidx], zt[NS + cidx], sc);
                    p1[i] = sc;
                }
                if (t + 1 < H) {
                    double* zn = zs_base + ((t + 1) & 1) * 16 * D + lane * D;
#pragma unroll
                    for (int i = 0; i < NS; ++i) zn[i] = p1[i];
#pragma unroll
                    for (int cidx = 0; cidx < NU; ++cidx) zn[NS + cidx] = acts[(lane * H + t + 1) * NU + cidx];
                }
            }
            if (lane_valid && st_step) reinterpret_cast<int*>(ac_row + 2)[0] |= st_step;
            double* nx = st_row((t + 1) & 1, lane);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                nx[i] = p1[i];
#pragma unroll
                for (int j = 0; j < NS; ++j) nx[NS + i * NS + j] = Q1[i][j];
                nx[S + i] = var[i];
            }
        };
        // finish(), second half (all 64 lanes of the calling wave; costs on lanes 0..15): the price of step t from the state
        // row finish_state(t) left in buffer (t + 1) & 1.  State constraint (safempc_cem.py:102-132,
        // gp_reachability_pytorch.py:184-231): lane l checks polytope rows (l >> 4) + 4 i of particle l & 15 -- d = h.p +
        // sqrt(h.Q h) - b >= 0 violates (a NaN distance does not, as in the reference); the verdicts of a particle's four
        // lanes meet in a ballot.
        auto finish_costs = [&](int t) {
            const int c = lane & 15;
            const double* o = st_row((t + 1) & 1, c);
            // (all LDS operands requested in one batch, as in finish_state: the state row, this lane's first polytope row,
            // the action, the cost sums and the objective's constants)
            double pp[NS], QQ[NS][NS], var[NS], hrow[NS], hv, uu[NU], umin[NU], umax[NU], wab[NS], tgt[NS], wli[NS];
            const int m = fcc.m, obj_mode = fcc.obj_mode, con_mode = fcc.con_mode;
            const int r0 = lane >> 4;
            const int r0c = r0 < m ? r0 : 0;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                pp[i] = o[i];
#pragma unroll
                for (int j = 0; j < NS; ++j) QQ[i][j] = o[NS + i * NS + j];
                var[i] = o[S + i];
                hrow[i] = fcc.h_mat[r0c * NS + i];
                wab[i] = fcc.w_abs[i];
                tgt[i] = fcc.target[i];
                wli[i] = fcc.w_lin[i];
            }
            hv = fcc.h_vec[r0c];
#pragma unroll
            for (int cidx = 0; cidx < NU; ++cidx) {
                uu[cidx] = acts[(c * H + t) * NU + cidx];
                umin[cidx] = fcc.u_min[cidx];
                umax[cidx] = fcc.u_max[cidx];
            }
            double obj = ac_row[0], con = ac_row[1];
            SX_PIN();
            bool viol = false;
            if (con_mode == SX_CON_ALL_STATES || t == H - 1) {
                for (int r = r0; r < m; r += 4) {
                    if (r != r0) {     // (polytopes of more than 4 rows: the further rows of this lane)
#pragma unroll
                        for (int i = 0; i < NS; ++i) hrow[i] = fcc.h_mat[r * NS + i];
                        hv = fcc.h_vec[r];
                    }
                    double hc = 0.0, hq = 0.0;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        hc += hrow[i] * pp[i];
                        double sacc = 0.0;
#pragma unroll
                        for (int j = 0; j < NS; ++j) sacc += QQ[i][j] * hrow[j];
                        hq += hrow[i] * sacc;
                    }
                    const double dist = hc + sqrt(hq) - hv;
                    viol = viol || (dist >= 0.0);
                }
            }
            unsigned long long bits = __ballot(viol);
            bits |= bits >> 32;
            bits |= bits >> 16;
            if (lane_owner) {
                if ((bits >> lane) & 1ull) con += SX_STATE_VIOLATION_COST;
                bool uviol = false;
#pragma unroll
                for (int cidx = 0; cidx < NU; ++cidx) uviol = uviol || (uu[cidx] < umin[cidx]) || (uu[cidx] > umax[cidx]);
                if (uviol) con += SX_ACTION_VIOLATION_COST;
                // objective (safempc_cem.py:304-312): -sum of the predicted variances, or the affine-abs form of the hook
                // (objective_cost's operation order: the step's cost is summed from 0, then added)
                double oadd = 0.0;
                if (obj_mode == SX_OBJ_NEG_VARIANCE) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) oadd -= var[i];
                } else {
#pragma unroll
                    for (int i = 0; i < NS; ++i) oadd += wab[i] * fabs(tgt[i] - pp[i]) + wli[i] * pp[i];
                }
                obj += oadd;
                ac_row[0] = obj;
                ac_row[1] = con;
                const int64_t g = (int64_t)e * rp.P + c0 + lane;
                if (lane_valid && rp.traj) {
                    double* tr = rp.traj + (g * H + t) * S;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        tr[i] = pp[i];
#pragma unroll
                        for (int j = 0; j < NS; ++j) tr[NS + i * NS + j] = QQ[i][j];
                    }
                }
                if (lane_valid && rp.sigma) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) rp.sigma[(g * H + t) * NS + i] = var[i];
                }
            }
        };

#ifdef SX_STAMPS
        unsigned long long c_k = 0, c_kb = 0, c_m = 0, c_mb = 0;
        unsigned long long rt0;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
        const unsigned long long ct0 = stamp();
#endif
        for (int t = 0; t < H; ++t) {
#ifdef SX_STAMPS
            const unsigned long long t0 = stamp();
#endif
            const bool kstar_wave = !(t > 0 && wave == 0);
            if (wave == 4 && t > 1) finish_costs(t - 2);
            if (kstar_wave) {
                double zq[D];
                const int c = lane & 15;
                if (t == 0) {
#pragma unroll
                    for (int j = 0; j < D; ++j) zq[j] = zs_base[c * D + j];
                    rw_kstar_phase(gc, kl, lds.kfrag, q0_begin, q0_end, zq);
                } else {
                    // (requesting the first trip's rows here, ahead of next_centre's round trip -- rw_kstar_prefetch -- costs
                    // 20 more live registers than this kernel has: 39 spilled dwords; not done)
                    double pc[NS];
                    next_centre(rc, c, zs_base + ((t - 1) & 1) * 16 * D + c * D, pc);
#pragma unroll
                    for (int i = 0; i < NS; ++i) zq[i] = pc[i];
#pragma unroll
                    for (int cidx = 0; cidx < NU; ++cidx) zq[NS + cidx] = acts[(c * H + t) * NU + cidx];
                    rw_kstar_phase(gc, kl, lds.kfrag, q_begin, q_end, zq);
                }
            } else if (owner) {
                finish_state(t - 1);
            }
#ifdef SX_STAMPS
            const unsigned long long t1 = stamp();
#endif
            if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the W pairs on their way to LDS by DMA)
            __syncthreads();
#ifdef SX_STAMPS
            const unsigned long long t2 = stamp();
#endif
            const v2d* wl = wlds + wave * LDSP * 64;
#if SX_RH_PRIO == 1
            // Two waves of a SIMD that take turns on the matrix pipe switch its accumulator with every instruction (67 instead
            // of 64.5 cycles per MFMA, tools/mfma_probe3.hip): the first four waves get the pipe whenever they are ready, their
            // partners fill the gaps and run alone afterwards.
            if (wave < 4) __builtin_amdgcn_s_setprio(3);
#endif
            switch (wave) {
                case 0: rh_mfma_phase<NS, D, NRB, 0, REGP, LDSP>(gc, lds, wl, lane, wreg); break;
                case 1: rh_mfma_phase<NS, D, NRB, 1, REGP, LDSP>(gc, lds, wl, lane, wreg); break;
                case 2: rh_mfma_phase<NS, D, NRB, 2, REGP, LDSP>(gc, lds, wl, lane, wreg); break;
                case 3: rh_mfma_phase<NS, D, NRB, 3, REGP, LDSP>(gc, lds, wl, lane, wreg); break;
                case 4: rh_mfma_phase<NS, D, NRB, 4, REGP, LDSP>(gc, lds, wl, lane, wreg); break;
                case 5: rh_mfma_phase<NS, D, NRB, 5, REGP, LDSP>(gc, lds, wl, lane, wreg); break;
                case 6: rh_mfma_phase<NS, D, NRB, 6, REGP, LDSP>(gc, lds, wl, lane, wreg); break;
                default: rh_mfma_phase<NS, D, NRB, 7, REGP, LDSP>(gc, lds, wl, lane, wreg); break;
            }
#if SX_RH_PRIO == 1
            if (wave < 4) __builtin_amdgcn_s_setprio(0);
#elif SX_RH_PRIO == 2
            __builtin_amdgcn_s_setprio(0);
#endif
#ifdef SX_STAMPS
            const unsigned long long t3 = stamp();
#endif
            __syncthreads();
#ifdef SX_STAMPS
            const unsigned long long t4 = stamp();
            c_k += t1 - t0; c_kb += t2 - t1; c_m += t3 - t2; c_mb += t4 - t3;
            if (t == 0) { first_k = t1 - t0; first_kb = t2 - t1; }
#endif
        }
#ifdef SX_STAMPS
        const unsigned long long loop_t1 = stamp();
        if (rp.stamps && lane == 0 && tile == (int)blockIdx.x) {
            unsigned long long* o = rp.stamps + ((size_t)blockIdx.x * nw + wave) * 8;
            o[0] = c_k; o[1] = c_kb; o[2] = c_m; o[3] = c_mb;
            o[4] = (ct0 - tile_t0) | (first_k << 32);   // prologue | the first step's Kstar phase
            unsigned long long rt1;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
            o[6] = stamp() - ct0;   // shader cycles of the step loop
            o[7] = rt1 - rt0;       // the same span in 10 ns ticks
        }
#endif
        // the last step's state on wave 0 while wave 4 closes the costs of step H - 2 beside it (they read the other state
        // buffer), then the last step's costs (the cost sums are read-modify-write: after wave 4's, hence the barrier)
        if (wave == 0 && owner) finish_state(H - 1);
        if (wave == 4 && H > 1) finish_costs(H - 2);
        __syncthreads();
        if (wave == 0) finish_costs(H - 1);
        if (valid) {
            const int64_t g = (int64_t)e * rp.P + c0 + tid;
            rp.obj_cost[g] = ac_row[0];
            rp.con_cost[g] = ac_row[1];
            const int st = reinterpret_cast<const int*>(ac_row + 2)[0];
            if (st) atomicOr(rp.status, st);
        }
        __syncthreads();   // the next tile's prologue reuses the Kstar buffer and the action table
#ifdef SX_STAMPS
        if (rp.stamps && lane == 0 && tile == (int)blockIdx.x)   // epilogue | the first step's wait at the barrier (W arriving)
            rp.stamps[((size_t)blockIdx.x * nw + wave) * 8 + 5] = (stamp() - loop_t1) | (first_kb << 32);
#endif
    }
}

}  // namespace sx
