// Micro-benchmark: one dependent fp32 32x32x2 MFMA chain per wave (one wave per SIMD) whose B operand comes from an LDS read issued
// DB steps earlier and whose A operand changes every step — the inner loop of the split-K latency kernel without its global loads.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_chain_lds mfma_chain_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f16v __attribute__((ext_vector_type(16)));

template <int DB, int STEPS, bool BARRIER>
__global__ __launch_bounds__(256) void chain(float* out, unsigned long long* clk, int iters, float a) {
    __shared__ float xs[2][32 * 48];
    f16v acc;
    for (int j = 0; j < 16; ++j) acc[j] = a + j;
    for (int i = threadIdx.x; i < 2 * 32 * 48; i += 256) (&xs[0][0])[i] = a * i;
    float av[STEPS];
    for (int j = 0; j < STEPS; ++j) av[j] = a + j + threadIdx.x;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b_lane = (8 * wave + (lane >> 5)) * 42 + (lane & 31);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const float* xsb = xs[it & 1];
        if (BARRIER) {
            xs[it & 1][threadIdx.x] = av[0];
            __syncthreads();
        }
        float bq[DB + 1];
#pragma unroll
        for (int d = 0; d < DB; ++d) bq[d] = xsb[b_lane + 2 * (d & 3) * 42 + (d >> 2)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sj = 0; sj < STEPS; ++sj) {
            if (sj + DB < STEPS) bq[(sj + DB) % (DB + 1)] = xsb[b_lane + 2 * ((sj + DB) & 3) * 42 + ((sj + DB) >> 2)];
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sj], bq[sj % (DB + 1)], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f;
    for (int j = 0; j < 16; ++j) r += acc[j];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int DB, int STEPS, bool BARRIER>
static void run(int grid, float* d_out, unsigned long long* d_clk) {
    const int iters = 400;
    hipLaunchKernelGGL((chain<DB, STEPS, BARRIER>), dim3(grid), dim3(256), 0, 0, d_out, d_clk, 10, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<DB, STEPS, BARRIER>), dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h = 0;
    hipMemcpy(&h, d_clk, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)iters * STEPS;
    printf("DB=%d steps/chunk=%2d barrier=%d grid=%d  %6.1f ns / MFMA  %6.1f ticks / MFMA\n", DB, STEPS, (int)BARRIER, grid, ms * 1e6 / n, (double)h / n);
}

int main() {
    float* d_out;
    unsigned long long* d_clk;
    hipMalloc(&d_out, 1024 * 256 * sizeof(float));
    hipMalloc(&d_clk, 16);
    run<1, 44, false>(176, d_out, d_clk);
    run<3, 44, false>(176, d_out, d_clk);
    run<5, 44, false>(176, d_out, d_clk);
    run<1, 44, true>(176, d_out, d_clk);
    run<3, 44, true>(176, d_out, d_clk);
    run<3, 12, true>(176, d_out, d_clk);
    run<3, 44, true>(512, d_out, d_clk);
    return 0;
}
