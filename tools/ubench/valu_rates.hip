// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (and a few others) on gfx950, many waves per SIMD, independent chains.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 32; i += 2)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&v[i]) : "v"(*(double*)&v[0 + 0 * i]), "v"(*(double*)&v[0]));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(a));
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
        }
    }
    float r = 0;
    for (int i = 0; i < 32; ++i) r += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, int per_iter, double flops_per_instr, float* d) {
    const int iters = 2000, grid = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)grid * 4 * iters * per_iter;   // wave-instructions
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal: time * 2.4e9 * (1024 SIMDs) / instr
    printf("%-14s %8.3f ms  %7.2f cycles/wave-instr/SIMD (at 2.4 GHz)  %7.1f TFLOP/s\n", name, ms,
           ms * 1e-3 * 2.4e9 * 1024.0 / instr, instr * 64 * flops_per_instr / ms * 1e-9);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", 32, 2, d);
    run<1>("v_pk_fma_f32", 16, 4, d);
    run<2>("v_mul_f32", 32, 1, d);
    run<3>("v_exp_f32", 32, 1, d);
    run<4>("v_mov_b32", 32, 0, d);
    run<5>("v_add_u32", 32, 0, d);
    return 0;
}
