// Probe: cycles per v_mfma_f64_16x16x4_f64 on gfx950 (register operands, 2 or 4 independent accumulators).
// hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void probe(double* out, unsigned long long* cyc, int iters, double a0, double b0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u % NACC], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
void run(int threads, int blocks) {
    const int iters = 2000;
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * (threads / 64));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<NACC><<<blocks, threads>>>(out, cyc, iters, 1.0, 2.0);
    hipEventRecord(e0);
    probe<NACC><<<blocks, threads>>>(out, cyc, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += v;
    mean /= h.size();
    const double mfma_per_wave = iters * 8.0;
    const double flops = mfma_per_wave * 2048.0 * h.size();
    printf("NACC=%d threads=%d blocks=%d: %.1f cycles/MFMA per wave, %.3f ms, %.1f TFLOP/s\n", NACC, threads, blocks,
           mean / mfma_per_wave, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    run<2>(256, 256);
    run<4>(256, 256);
    run<2>(512, 256);
    run<4>(512, 256);
    run<1>(256, 256);
    run<2>(256, 512);
    return 0;
}
