// Micro-benchmark (round 3): what does one non-MFMA instruction cost beside exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32) on gfx950?
// The fp32 MFMA runs at the fp32 VALU rate, so "does this op steal matrix time" decides how activations are best evaluated
// (SiLU = v_exp + v_rcp; snake = sin^2 by polynomial or by the hardware v_cos).
//   arm A: a wave issues 1 MFMA + R ops of kind K per group (same wave)
//   arm B: 512-thread workgroups, waves 0-3 only MFMAs, waves 4-7 only ops of kind K (two waves per SIMD, separate streams)
// Also prints the accuracy of sin^2 via v_cos / v_sin against double on a sweep.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_mix mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum Kind { K_FMA = 0, K_EXP, K_RCP, K_SIN, K_COS, K_PKFMA, K_RNDNE, K_LDS, K_COUNT };
static const char* kNames[K_COUNT] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_sin_f32", "v_cos_f32", "v_pk_fma_f32", "v_rndne_f32", "ds_read_b32"};

template <int K>
__device__ __forceinline__ float op(float v, float s0, float s1, const float* lds, int i) {
    if (K == K_FMA) return __builtin_fmaf(v, s0, s1);
    if (K == K_EXP) return __builtin_amdgcn_exp2f(v);
    if (K == K_RCP) return __builtin_amdgcn_rcpf(v);
    if (K == K_SIN) return __builtin_amdgcn_sinf(v);
    if (K == K_COS) return __builtin_amdgcn_cosf(v);
    if (K == K_RNDNE) return __builtin_rintf(v);
    if (K == K_LDS) return lds[(threadIdx.x + i * 64) & 4095];
    return v;
}

template <int K, int NMFMA, int NOP, bool SPLIT, int PRIO = 0, int BURST = 1>
__global__ __launch_bounds__(SPLIT ? 512 : 256) void arm(float* out, int iters, float a, float b, long long* cyc = nullptr) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = a + i * 1e-6f;
    __syncthreads();
    f16v acc[4];
    float v[16];
    f2 pv[8];
    const float x = (float)(threadIdx.x & 63) * 1e-3f + a;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = x + i + j;
    for (int i = 0; i < 16; ++i) v[i] = x * (i + 1) * 0.01f + 0.5f;
    for (int i = 0; i < 8; ++i) pv[i] = f2{x + i, x - i};
    const float s0 = a, s1 = b;
    const bool do_mfma = !SPLIT || threadIdx.x < 256;
    const bool do_op = !SPLIT || threadIdx.x >= 256;
    if (SPLIT && PRIO && do_op) __builtin_amdgcn_s_setprio(PRIO);
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (SPLIT && BURST > 1 && do_op) {   // the op wave issues BURST iterations' worth of ops back to back, then sleeps
            if (it % BURST == 0) {
#pragma unroll 1
                for (int bb = 0; bb < BURST; ++bb) {
#pragma unroll
                    for (int r = 0; r < 4 * NOP; ++r) {
                        const int idx = r & 15;
                        if (K == K_PKFMA) pv[idx & 7] = __builtin_elementwise_fma(pv[idx & 7], f2{s0, s0}, f2{s1, s1});
                        else v[idx] = op<K>(v[idx], s0, s1, lds, idx);
                    }
                }
            } else {
                __builtin_amdgcn_s_sleep(3);
            }
            continue;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (NMFMA && do_mfma) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, s1, acc[g], 0, 0, 0);
            if (do_op) {
#pragma unroll
                for (int r = 0; r < NOP; ++r) {
                    const int idx = (g * NOP + r) & 15;
                    if (K == K_PKFMA) pv[idx & 7] = __builtin_elementwise_fma(pv[idx & 7], f2{s0, s0}, f2{s1, s1});
                    else v[idx] = op<K>(v[idx], s0, s1, lds, idx);
                }
            }
            if (!SPLIT && NMFMA && NOP) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(K == K_LDS ? 0x100 : 0x002, NOP, 0);
            }
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (cyc && (threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    float r = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) r += acc[i][j];
    for (int i = 0; i < 16; ++i) r += v[i];
    for (int i = 0; i < 8; ++i) r += pv[i].x + pv[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// two waves per SIMD, one only MFMAs, one only ops: per-wave s_memtime cycles (100 MHz ticks -> shown as a ratio to the MFMA-only run)
template <int K, int NOP, int PRIO, int BURST>
static void split_detail(float* d_out, const char* tag) {
    const int iters = 2000, grid = 256;
    long long* d_c;
    hipMalloc(&d_c, grid * 8 * sizeof(long long));
    hipMemset(d_c, 0, grid * 8 * sizeof(long long));
    hipLaunchKernelGGL((arm<K, 1, NOP, true, PRIO, BURST>), dim3(grid), dim3(512), 0, 0, d_out, iters, 1.0f, 0.5f, d_c);
    hipDeviceSynchronize();
    std::vector<long long> c(grid * 8);
    hipMemcpy(c.data(), d_c, c.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0, o = 0;
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? m : o) += (double)c[b * 8 + w];
    m /= grid * 4; o /= grid * 4;
    printf("  %-13s %-28s ops/MFMA=%2d: MFMA wave %7.1f ticks/1000 MFMAs, op wave %7.1f ticks/1000 groups\n", kNames[K], tag, NOP, m / (iters * 4) * 1000, o / (iters * 4) * 1000);
    hipFree(d_c);
}

template <int K, int NM, int NOP, bool SPLIT>
static double run(float* d_out, int blocks_per_cu = 1) {
    const int iters = 2000;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((arm<K, NM, NOP, SPLIT>), dim3(grid), dim3(SPLIT ? 512 : 256), 0, 0, d_out, 10, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((arm<K, NM, NOP, SPLIT>), dim3(grid), dim3(SPLIT ? 512 : 256), 0, 0, d_out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / (iters * 4.0);   // seconds per group (1 MFMA + NOP ops)
}

template <int K>
static void kind(float* d_out, double clk) {
    const double m = run<K, 1, 0, false>(d_out) * clk;
    const double o8 = run<K, 0, 8, false>(d_out) * clk / 8;
    const double a1 = run<K, 1, 1, false>(d_out) * clk, a2 = run<K, 1, 2, false>(d_out) * clk, a4 = run<K, 1, 4, false>(d_out) * clk,
                 a8 = run<K, 1, 8, false>(d_out) * clk;
    const double s4 = run<K, 1, 4, true>(d_out) * clk, s8 = run<K, 1, 8, true>(d_out) * clk, s16 = run<K, 1, 16, true>(d_out) * clk;
    printf("%-13s alone %5.1f cyc/op | same wave: MFMA %5.1f, +1 %5.1f, +2 %5.1f, +4 %5.1f, +8 %5.1f (per op %+5.1f) | partner wave: +4 %5.1f, +8 %5.1f, +16 %5.1f (per op %+5.1f)\n",
           kNames[K], o8, m, a1, a2, a4, a8, (a8 - m) / 8, s4, s8, s16, (s16 - m) / 16);
}

__global__ void acc_kernel(const float* z, float* o_cos, float* o_sin, float* o_poly, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = z[i];
    const float c = __builtin_amdgcn_cosf(v * 0.318309886183790672f);   // cos(2 pi * z / pi) = cos(2z)
    o_cos[i] = 0.5f - 0.5f * c;
    const float s = __builtin_amdgcn_sinf(v * 0.159154943091895336f);
    o_sin[i] = s * s;
    float k = rintf(v * 0.318309886183790672f);
    float r = fmaf(k, -3.140625f, v);
    r = fmaf(k, -9.67502593994140625e-4f, r);
    r = fmaf(k, -1.509957990978376432e-7f, r);
    const float r2 = r * r;
    float p = fmaf(r2, -2.50521083854417188e-8f, 2.75573192239858925e-6f);
    p = fmaf(r2, p, -1.98412698412698413e-4f);
    p = fmaf(r2, p, 8.33333333333333322e-3f);
    p = fmaf(r2, p, -1.66666666666666657e-1f);
    const float sn = fmaf(r * r2, p, r);
    o_poly[i] = sn * sn;
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 256 * 8 * 512 * sizeof(float));
    // clock estimate: MFMA-only arm is 64 cycles per MFMA
    const double t_m = run<K_FMA, 1, 0, false>(d_out);
    const double clk = 64.0 / t_m;
    printf("clock estimate from the MFMA-only arm: %.2f GHz (64 cycles per 32x32x2 fp32 MFMA)\n", clk * 1e-9);
    kind<K_FMA>(d_out, clk);
    kind<K_PKFMA>(d_out, clk);
    kind<K_EXP>(d_out, clk);
    kind<K_RCP>(d_out, clk);
    kind<K_SIN>(d_out, clk);
    kind<K_COS>(d_out, clk);
    kind<K_RNDNE>(d_out, clk);
    kind<K_LDS>(d_out, clk);

    printf("two waves per SIMD (s_memtime ticks of each wave; MFMA-only reference first):\n");
    split_detail<K_FMA, 0, 0, 1>(d_out, "no ops");
    split_detail<K_FMA, 4, 0, 1>(d_out, "prio 0");
    split_detail<K_FMA, 4, 3, 1>(d_out, "op wave s_setprio 3");
    split_detail<K_FMA, 8, 0, 1>(d_out, "prio 0");
    split_detail<K_FMA, 8, 3, 1>(d_out, "op wave s_setprio 3");
    split_detail<K_FMA, 8, 0, 16>(d_out, "bursts of 16 groups, prio 0");
    split_detail<K_FMA, 8, 3, 16>(d_out, "bursts of 16 groups, prio 3");
    split_detail<K_EXP, 4, 0, 1>(d_out, "prio 0");
    split_detail<K_EXP, 4, 3, 1>(d_out, "op wave s_setprio 3");
    split_detail<K_PKFMA, 4, 0, 1>(d_out, "prio 0");
    split_detail<K_PKFMA, 4, 3, 1>(d_out, "op wave s_setprio 3");
    split_detail<K_LDS, 8, 0, 1>(d_out, "prio 0");

    const int n = 1 << 20;
    std::vector<float> z(n), a(n), b(n), c(n);
    for (int i = 0; i < n; ++i) z[i] = -40.0f + 80.0f * (float)i / n;
    float *dz, *da, *db, *dc;
    hipMalloc(&dz, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dz, z.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(acc_kernel, dim3(n / 256), dim3(256), 0, 0, dz, da, db, dc, n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    for (double lim : {1.0, 5.0, 40.0}) {
        double ea = 0, eb = 0, ec = 0;
        for (int i = 0; i < n; ++i) {
            if (std::fabs(z[i]) > lim) continue;
            const double ref = std::sin((double)z[i]) * std::sin((double)z[i]);
            ea = std::fmax(ea, std::fabs(a[i] - ref));
            eb = std::fmax(eb, std::fabs(b[i] - ref));
            ec = std::fmax(ec, std::fabs(c[i] - ref));
        }
        printf("sin^2(z), |z| <= %4.0f: max abs error  0.5-0.5*v_cos %.3e   v_sin^2 %.3e   polynomial %.3e\n", lim, ea, eb, ec);
    }
    return 0;
}
