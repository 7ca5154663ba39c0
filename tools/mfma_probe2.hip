// Probe: how much do VALU / SALU / LDS instructions placed between f64 MFMAs cost a single wave per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <int KIND, int N>
__global__ void probe(double* out, unsigned long long* cyc, int iters, double a0, double b0, int* tab) {
    __shared__ v2d lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = v2d{1.0 + i, 2.0};
    __syncthreads();
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    int x = threadIdx.x, y = __builtin_amdgcn_readfirstlane(tab[0]);
    long long p = (long long)tab[threadIdx.x & 63];
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (KIND == 0) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y)); }
                if (KIND == 1) { int s; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(x)); asm volatile("s_add_u32 %0, %0, %1" : "+s"(y) : "s"(s)); }
                if (KIND == 2) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(p) : "v"(p)); }
                if (KIND == 3) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(y)); }
                if (KIND == 4) { v2d v = lds[(x + k * 64) & 1023]; asm volatile("" :: "v"(v)); }
                if (KIND == 5) { asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a0)); }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + x + y + (double)p + a0;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND, int N>
void run(const char* name, int threads) {
    const int iters = 500, blocks = 256;
    double* out; unsigned long long* cyc; int* tab;
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipMalloc(&cyc, 8 * blocks * (threads / 64));
    hipMalloc(&tab, 4 * 64); hipMemset(tab, 0, 4 * 64);
    probe<KIND, N><<<blocks, threads>>>(out, cyc, iters, 1.0, 2.0, tab);
    probe<KIND, N><<<blocks, threads>>>(out, cyc, iters, 1.0, 2.0, tab);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= h.size();
    printf("%-14s N=%2d per MFMA-pair, threads=%d: %.1f cycles per MFMA\n", name, N, threads, mean / (iters * 8.0));
}

int main() {
    run<0, 0>("none", 256);
    run<0, 4>("v_add_u32", 256); run<0, 8>("v_add_u32", 256); run<0, 16>("v_add_u32", 256); run<0, 24>("v_add_u32", 256);
    run<1, 2>("readfirstlane", 256); run<1, 4>("readfirstlane", 256); run<1, 8>("readfirstlane", 256);
    run<2, 2>("lshl_add_u64", 256); run<2, 4>("lshl_add_u64", 256); run<2, 8>("lshl_add_u64", 256);
    run<3, 8>("s_add", 256); run<3, 16>("s_add", 256); run<3, 32>("s_add", 256);
    run<4, 1>("ds_read_b128", 256); run<4, 2>("ds_read_b128", 256); run<4, 4>("ds_read_b128", 256);
    run<5, 2>("v_fma_f64", 256); run<5, 4>("v_fma_f64", 256); run<5, 8>("v_fma_f64", 256);
    run<0, 16>("v_add_u32", 512); run<5, 8>("v_fma_f64", 512);
    return 0;
}
