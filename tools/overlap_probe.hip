// Probe: do f64 MFMAs of one wave overlap with VALU work of ANOTHER wave on the same SIMD (gfx950)?
// 512 threads = 8 waves, waves w and w + 4 share a SIMD.  Waves 0-3 run role RA, waves 4-7 role RB:
//   0 idle, 1 v_mfma_f64_16x16x4 chain, 2 f64 FMA (4 independent chains), 3 i32 add (4 independent chains),
//   4 f32 FMA (4 independent chains)
// One MFMA = 64 matrix cycles; 16 independent f64 FMAs = 64 VALU cycles, so every role is sized to the same
// stand-alone time.  Overlap: t(1,2) ~ max(t(1,0), t(0,2)); shared pipe: t(1,2) ~ sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int ROLE>
__device__ __forceinline__ double role(int iters, double seed) {
    double a = seed, b = seed * 2, c = seed * 3, d = seed * 4;
    const double m = 0.999999, k = 1e-9;
    float fa = (float)seed, fb = fa * 2, fc = fa * 3, fd = fa * 4;
    const float fm = 0.9999f, fk = 1e-6f;
    int x = (int)seed, y = 3, z = 5, w = 7;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
        if (ROLE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        if (ROLE == 2) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(b) : "v"(m), "v"(k));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(c) : "v"(m), "v"(k));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(m), "v"(k));
            }
        }
        if (ROLE == 3) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(y) : "v"(z));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(z) : "v"(w));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(w) : "v"(x));
            }
        }
        if (ROLE == 4) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fm), "v"(fk));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fb) : "v"(fm), "v"(fk));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fc) : "v"(fm), "v"(fk));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fd) : "v"(fm), "v"(fk));
            }
        }
    }
    return a + b + c + d + acc[0] + acc[1] + acc[2] + acc[3] + x + y + z + w + fa + fb + fc + fd;
}

template <int RA, int RB>
__global__ void __launch_bounds__(512) probe(double* out, int iters, double seed) {
    const int wave = threadIdx.x >> 6;
    double r;
    if (wave < 4)
        r = role<RA>(iters, seed + threadIdx.x);
    else
        r = role<RB>(iters, seed + threadIdx.x);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int RA, int RB>
float run(const char* name) {
    const int iters = 2000, blocks = 256;
    double* out;
    hipMalloc(&out, sizeof(double) * 512 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<RA, RB><<<blocks, 512>>>(out, iters, 1.0);
    hipEventRecord(e0);
    probe<RA, RB><<<blocks, 512>>>(out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f ms  (%.1f ns per 512-cycle unit)\n", name, ms, ms * 1e6 / iters);
    hipFree(out);
    return ms;
}

int main() {
    run<1, 0>("mfma | idle");
    run<0, 2>("idle | f64 fma");
    run<0, 3>("idle | i32 add");
    run<0, 4>("idle | f32 fma");
    run<1, 1>("mfma | mfma");
    run<2, 2>("f64 fma | f64 fma");
    run<1, 2>("mfma | f64 fma");
    run<1, 3>("mfma | i32 add");
    run<1, 4>("mfma | f32 fma");
    return 0;
}
