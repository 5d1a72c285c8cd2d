// Micro-benchmark: issue interval of DEPENDENT fp32 32x32x2 MFMAs (one accumulator chain per wave, as the split-K latency kernel
// has) against 2 / 4 independent chains, at one wave per SIMD, on a sparse (176 workgroups) and a full (256, 1024) grid.
// Prints ns and shader-clock cycles (s_memtime) per MFMA and the clock the kernel ran at (s_memtime / wall_clock64).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_chain mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, unsigned long long* clk, int iters, float a, float b) {
    f16v acc[NACC];
    const float x = (float)threadIdx.x * 1e-3f + a;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = x + i + j;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int g = 0; g < NACC; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
    }
    float r = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) r += acc[i][j];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int NACC>
static void run(int grid, float* d_out, unsigned long long* d_clk) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((chain<NACC>), dim3(grid), dim3(256), 0, 0, d_out, d_clk, 10, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<NACC>), dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, d_clk, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * NACC;
    // wall_clock64 ticks at 100 MHz
    printf("chains=%d grid=%4d  %8.3f ms  %6.1f ns / MFMA  %6.1f s_memtime ticks / MFMA  (s_memtime %.0f MHz)  %6.1f TF\n", NACC, grid, ms,
           ms * 1e6 / n, (double)h[0] / n, (double)h[0] / ((double)h[1] / 100.0), (double)grid * 4 * n * 4096.0 / ms * 1e-9);
}

int main() {
    float* d_out;
    unsigned long long* d_clk;
    hipMalloc(&d_out, 1024 * 256 * sizeof(float));
    hipMalloc(&d_clk, 16);
    for (int grid : {176, 256, 1024}) {
        run<1>(grid, d_out, d_clk);
        run<2>(grid, d_out, d_clk);
        run<4>(grid, d_out, d_clk);
    }
    return 0;
}
