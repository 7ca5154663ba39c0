// Probe: f64 VALU issue rate of ONE wave per SIMD against two, for 1 / 2 / 4 / 8 independent chains (gfx950).
// The register-resident rollout (sx_rollout_rw.hpp) runs its Kstar phase with one wave per SIMD.
//     hipcc --offload-arch=gfx950 -O3 tools/valu_probe3.hip -o tools/valu_probe3.bin && tools/valu_probe3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int KIND>
__global__ void probe(double* out, int iters, double a0) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = a0 + threadIdx.x * 1e-3 + i;
    const double m = 0.999999, k = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / CH; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(k));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&a[c]) : "v"((float)m), "v"((float)k));
                if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(*(int*)&a[c]) : "v"(3));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH, int KIND>
void run(const char* name, int threads) {
    const int iters = 500, blocks = 256;
    double* out;
    (void)hipMalloc(&out, sizeof(double) * threads * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    probe<CH, KIND><<<blocks, threads>>>(out, iters, 1.0);
    (void)hipEventRecord(e0);
    probe<CH, KIND><<<blocks, threads>>>(out, iters, 1.0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int wps = threads / 256;
    printf("%-8s chains=%d waves/SIMD=%d: %.3f ms -> %.2f ns per wave-instruction per SIMD\n", name, CH, wps, ms,
           ms * 1e6 / (iters * 64.0 * wps));
    (void)hipFree(out);
}
int main() {
    for (int threads : {256, 512}) {
        run<1, 0>("fma_f64", threads); run<2, 0>("fma_f64", threads); run<4, 0>("fma_f64", threads); run<8, 0>("fma_f64", threads);
        run<1, 1>("fma_f32", threads); run<4, 1>("fma_f32", threads); run<8, 1>("fma_f32", threads);
        run<1, 2>("add_u32", threads); run<4, 2>("add_u32", threads);
    }
}
