// Probe: issue cost of the non-FMA f64 instructions exp() needs (gfx950), 4 independent chains, 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(512) probe(double* out, int iters, double a0) {
    double a = a0 + threadIdx.x * 1e-3, b = a0 * 2 + threadIdx.x * 1e-3, c = a0 * 3, d = a0 * 4;
    const double m = 0.999999, k = 1e-9;
    int x = threadIdx.x & 3, y = 1, z = 2, w = 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(b) : "v"(m), "v"(k));
                             asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(c) : "v"(m), "v"(k)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(m), "v"(k)); }
            if (KIND == 1) { asm volatile("v_rndne_f64 %0, %0" : "+v"(a)); asm volatile("v_rndne_f64 %0, %0" : "+v"(b));
                             asm volatile("v_rndne_f64 %0, %0" : "+v"(c)); asm volatile("v_rndne_f64 %0, %0" : "+v"(d)); }
            if (KIND == 2) { asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(x) : "v"(a)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(y) : "v"(b));
                             asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(z) : "v"(c)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(w) : "v"(d)); }
            if (KIND == 3) { asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(x)); asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(b) : "v"(x));
                             asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(c) : "v"(x)); asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d) : "v"(x)); }
            if (KIND == 4) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(m)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(b) : "v"(m));
                             asm volatile("v_mul_f64 %0, %0, %1" : "+v"(c) : "v"(m)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(m)); }
            if (KIND == 5) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(k)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(b) : "v"(k));
                             asm volatile("v_add_f64 %0, %0, %1" : "+v"(c) : "v"(k)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(k)); }
            if (KIND == 6) { asm volatile("v_lshl_add_u32 %0, %1, 20, %0" : "+v"(x) : "v"(y)); asm volatile("v_lshl_add_u32 %0, %1, 20, %0" : "+v"(y) : "v"(z));
                             asm volatile("v_lshl_add_u32 %0, %1, 20, %0" : "+v"(z) : "v"(w)); asm volatile("v_lshl_add_u32 %0, %1, 20, %0" : "+v"(w) : "v"(x)); }
            if (KIND == 7) { asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(k)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(b) : "v"(k));
                             asm volatile("v_max_f64 %0, %0, %1" : "+v"(c) : "v"(k)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(d) : "v"(k)); }
            if (KIND == 8) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(y) : "v"(z));
                             asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(z) : "v"(w)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w) : "v"(x)); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + x + y + z + w;
}
template <int KIND>
void run(const char* name) {
    const int iters = 500, blocks = 256, threads = 512;
    double* out;
    (void)hipMalloc(&out, sizeof(double) * threads * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    probe<KIND><<<blocks, threads>>>(out, iters, 1.0);
    (void)hipEventRecord(e0);
    probe<KIND><<<blocks, threads>>>(out, iters, 1.0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // 2 waves per SIMD, each iters * 64 instructions
    printf("%-18s %.3f ms -> %.2f ns per wave-instruction per SIMD (fma = 4 cycles)\n", name, ms, ms * 1e6 / (iters * 64.0 * 2));
    (void)hipFree(out);
}
int main() {
    run<0>("v_fma_f64"); run<4>("v_mul_f64"); run<5>("v_add_f64"); run<7>("v_max_f64"); run<1>("v_rndne_f64");
    run<2>("v_cvt_i32_f64"); run<3>("v_ldexp_f64"); run<6>("v_lshl_add_u32"); run<8>("v_cndmask_b32");
}
