// Elite refit helper shared by the rollout kernel's prologue (sx_cem_rollout_elites) and the ranking kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace sx {

// Mean and unbiased standard deviation of columns of the elite rows -- v_r = col[r * stride], r < k -- by ONE wave and
// without a workgroup barrier.  The wave takes 2^cshift adjacent columns at once: lane l works on column l & (2^cshift - 1)
// and adds the rows g, g + RG, g + 2 RG, ... (g = l >> cshift, RG = 64 >> cshift row groups) in that order, a butterfly over
// the row groups adds the partial sums (the same total in every lane of a column, bit for bit: each level adds the same
// two numbers on both sides); the second pass runs on the values the first one kept in registers (8 per lane; rows beyond
// are read again: L2 hits).  The loads of a batch -- 8 per lane, none beyond row k -- are in flight together: at config 2
// (k = 409, 15 columns, 8 waves x 2 columns: 13 loads per lane) two round trips per pass.  Std is 0 for k == 1.
// `col` points at this lane's column; rows past k read row k - 1.
__device__ __forceinline__ double wave_total(double v, int cshift) {
    for (int off = 1 << cshift; off < 64; off <<= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ void wave_refit_columns(const double* __restrict__ col, int k, int stride, int lane, int cshift,
                                                   double& mean, double& sd) {
    constexpr int kKeep = 8;   // loads in flight per lane (more of them cost the rollout kernel's step loop registers: sx_rollout.hpp)
    const int g = lane >> cshift, rg = 64 >> cshift;
    double keep[kKeep];
    auto fetch = [&](int first, double (&v)[kKeep]) {
#pragma unroll
        for (int j = 0; j < kKeep; ++j) {
            v[j] = 0.0;
            if (first + rg * j < k) {   // (uniform: no load is issued for row groups that lie wholly past the rows)
                const int r = first + rg * j + g;
                v[j] = col[(long long)(r < k ? r : k - 1) * stride];
            }
        }
#pragma unroll
        for (int j = 0; j < kKeep; ++j)
            if (first + rg * j + g >= k) v[j] = 0.0;
    };
    fetch(0, keep);
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < kKeep; ++j) s += keep[j];
    for (int first = rg * kKeep; first < k; first += rg * kKeep) {
        double v[kKeep];
        fetch(first, v);
#pragma unroll
        for (int j = 0; j < kKeep; ++j) s += v[j];
    }
    mean = wave_total(s, cshift) / k;
    double ss = 0.0;
#pragma unroll
    for (int j = 0; j < kKeep; ++j) {
        const double dv = keep[j] - mean;
        if (rg * j + g < k) ss += dv * dv;
    }
    for (int first = rg * kKeep; first < k; first += rg * kKeep) {
        double v[kKeep];
        fetch(first, v);
#pragma unroll
        for (int j = 0; j < kKeep; ++j) {
            const double dv = v[j] - mean;
            if (first + rg * j + g < k) ss += dv * dv;
        }
    }
    sd = (k > 1) ? sqrt(wave_total(ss, cshift) / (k - 1)) : 0.0;
}

}  // namespace sx
