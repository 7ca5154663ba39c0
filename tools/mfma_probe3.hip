// Probe: what does the f64 MFMA stream of the register-resident rollout cost per instruction, and why is it not 64 cycles?
// One wave per SIMD (256 threads, launch bound 1 wave/EU), NP = 91 resident A pairs, patterns:
//   chain   one accumulator, A from the resident pairs, B a fixed register
//   pairs7  seven accumulators, two dependent MFMAs per (pair, row-block) as rw_mfma_phase issues them, B fixed
//   lds7    the same with B from LDS (ds_read_b128 two pairs ahead)
//   alt7    seven accumulators, ONE MFMA per accumulator in turn (no dependent neighbours)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe3.hip -o tools/mfma_probe3.bin && tools/mfma_probe3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int NP = 91;
#define PIN() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void probe(const v2d* __restrict__ w, double* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    v2d a[NP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NP; ++i) a[i] = w[(wave * NP + i) * 64 + lane];
    for (int i = threadIdx.x; i < 26 * 128; i += 256) smem[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const v2d* kb = reinterpret_cast<const v2d*>(smem) + lane;
    v2d bfix = {1.0 + lane * 1e-3, 2.0 - lane * 1e-3};
    double tot = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int t = 0; t < iters; ++t) {
        asm volatile("" : "+v"(bfix));
        v4d acc[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) acc[r] = v4d{0, 0, 0, 0};
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, bfix.x, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, bfix.y, acc[0], 0, 0, 0);
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                acc[(2 * i) % 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, bfix.x, acc[(2 * i) % 7], 0, 0, 0);
                PIN();
                acc[(2 * i + 1) % 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, bfix.y, acc[(2 * i + 1) % 7], 0, 0, 0);
                PIN();
            }
        } else {
            constexpr int rbs[7] = {12, 9, 8, 5, 4, 1, 0};
            int idx = 0;
            v2d b[3];
            if (MODE == 2) {
                b[0] = kb[0];
                b[1] = kb[128];
            }
#pragma unroll
            for (int q = 0; q < 26; ++q) {
                if (MODE == 2 && q + 2 < 26) b[(q + 2) % 3] = kb[(q + 2) * 128];
                PIN();
                const v2d bq = MODE == 2 ? b[q % 3] : bfix;
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    if (q < 2 * (rbs[r] + 1)) {
                        acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[idx].x, bq.x, acc[r], 0, 0, 0);
                        acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[idx].y, bq.y, acc[r], 0, 0, 0);
                        ++idx;
                    }
                }
                PIN();
            }
        }
#pragma unroll
        for (int r = 0; r < 7; ++r) tot += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * 256 + threadIdx.x] = tot;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, const v2d* w) {
    const int iters = 200, blocks = 256;
    double* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * 256 * blocks);
    (void)hipMalloc(&cyc, 8 * blocks * 4);
    probe<MODE><<<blocks, 256, 26 * 1024>>>(w, out, cyc, iters);
    probe<MODE><<<blocks, 256, 26 * 1024>>>(w, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += v;
    mean /= h.size();
    printf("%-8s %.2f cycles per MFMA (incl. %d-instruction epilogue per %d MFMAs)\n", name, mean / (iters * 2.0 * NP), 28, 2 * NP);
    (void)hipFree(out);
    (void)hipFree(cyc);
}
int main() {
    v2d* w;
    (void)hipMalloc(&w, sizeof(v2d) * 4 * NP * 64);
    (void)hipMemset(w, 0, sizeof(v2d) * 4 * NP * 64);
    run<0>("chain", w);
    run<1>("pairs7", w);
    run<2>("lds7", w);
    run<3>("alt7", w);
}
