// Probe: issue cost of dependent vs independent f64 / i32 VALU streams on gfx950, one and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ void probe(double* out, unsigned long long* cyc, int iters, double a0) {
    double a = a0 + threadIdx.x, b = a0 * 2 + threadIdx.x, c = a0 * 3, d = a0 * 4;
    const double m = 0.999999, k = 1e-9;
    int x = threadIdx.x, y = 3, z = 5, w = 7;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));
                             asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k)); }
            if (KIND == 1) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(b) : "v"(m), "v"(k));
                             asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(c) : "v"(m), "v"(k)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(m), "v"(k)); }
            if (KIND == 2) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                             asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y)); }
            if (KIND == 3) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(y) : "v"(z));
                             asm volatile("v_add_u32 %0, %0, %1" : "+v"(z) : "v"(w)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(w) : "v"(x)); }
            if (KIND == 4) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(b) : "v"(m), "v"(k));
                             asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(b) : "v"(m), "v"(k)); }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + x + y + z + w;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int KIND>
void run(const char* name, int threads) {
    const int iters = 500, blocks = 256;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * threads * blocks); hipMalloc(&cyc, 8 * blocks * (threads / 64));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<blocks, threads>>>(out, cyc, iters, 1.0);
    hipEventRecord(e0);
    probe<KIND><<<blocks, threads>>>(out, cyc, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= h.size();
    const double ninst = iters * 32.0;
    printf("%-28s threads=%4d: %.2f cycles/inst per wave; wall %.3f ms -> %.2f ns per inst per SIMD-wave-slot\n", name, threads, mean / ninst, ms,
           ms * 1e6 / (ninst * (threads / 256.0)));
}
int main() {
    for (int th : {256, 512, 1024}) {
        run<0>("f64 fma dependent", th); run<4>("f64 fma 2 chains", th); run<1>("f64 fma 4 independent", th);
        run<2>("i32 add dependent", th); run<3>("i32 add 4 independent", th);
    }
}
