// Micro-benchmark: can MFMA (fp32 32x32x2 or, with -DUSE_F16, f16 32x32x16) and fp32 VALU FMAs execute concurrently on gfx950?
// Arms: MFMA only, VALU only, and interleaved (1 MFMA + R v_fma per group), at 1..4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NMFMA, int NVALU>
__global__ __launch_bounds__(256) void arm(float* out, int iters, float a, float b) {
    f16v acc[4];
    float v[32];
    const float x = (float)threadIdx.x * 1e-3f + a;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = x + i + j;
    for (int i = 0; i < 32; ++i) v[i] = x * (i + 1);
    float s0 = a, s1 = b;
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(x + i); hb[i] = (_Float16)(x - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#ifdef USE_F16
            if (NMFMA) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[g], 0, 0, 0);
#else
            if (NMFMA) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, s1, acc[g], 0, 0, 0);
#endif
#pragma unroll
            for (int r = 0; r < NVALU; ++r) {
                const int idx = (g * NVALU + r) & 31;
                v[idx] = __builtin_fmaf(v[idx], s0, s1);
            }
            if (NMFMA && NVALU) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) r += acc[i][j];
    for (int i = 0; i < 32; ++i) r += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NM, int NV>
static void run(const char* name, int blocks_per_cu, float* d_out) {
    const int iters = 4000;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((arm<NM, NV>), dim3(grid), dim3(256), 0, 0, d_out, 10, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((arm<NM, NV>), dim3(grid), dim3(256), 0, 0, d_out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * 4;
    #ifdef USE_F16
    const double mfma_flops = NM ? waves * iters * 4.0 * (32.0 * 32 * 16 * 2) : 0.0;
#else
    const double mfma_flops = NM ? waves * iters * 4.0 * (32.0 * 32 * 2 * 2) : 0.0;
#endif
    const double valu_flops = waves * iters * 4.0 * NV * 64.0 * 2;
    printf("%-22s waves/SIMD=%d  %8.3f ms  MFMA %7.1f TF  VALU %7.1f TF  sum %7.1f TF\n", name, blocks_per_cu, ms,
           mfma_flops / ms * 1e-9, valu_flops / ms * 1e-9, (mfma_flops + valu_flops) / ms * 1e-9);
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    for (int w = 1; w <= 4; ++w) {
        run<1, 0>("mfma only", w, d_out);
        run<0, 8>("valu only (8)", w, d_out);
        run<0, 32>("valu only (32)", w, d_out);
        run<1, 1>("mfma + 1 fma", w, d_out);
        run<1, 2>("mfma + 2 fma", w, d_out);
        run<1, 4>("mfma + 4 fma", w, d_out);
        run<1, 8>("mfma + 8 fma", w, d_out);
        run<1, 6>("mfma + 6 fma", w, d_out);
        run<1, 16>("mfma + 16 fma", w, d_out);
    }
    return 0;
}
