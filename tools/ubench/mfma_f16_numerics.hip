// Numerics probe for v_mfma_f32_32x32x16_f16 / _bf16 on gfx950: subnormal inputs, exactness of the fp32 accumulation.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_f16_numerics mfma_f16_numerics.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// A: 32 x 16 (row m, k), B: 16 x 32 (k, col n), C in, D out (32 x 32), all row-major fp32 in memory; converted to f16 here.
__global__ void probe_f16(const float* A, const float* B, const float* C, float* D) {
    const int lane = threadIdx.x;
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        const int k = 8 * (lane >> 5) + i;
        a[i] = (_Float16)A[(lane & 31) * 16 + k];
        b[i] = (_Float16)B[k * 32 + (lane & 31)];
    }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}
__global__ void probe_bf16(const float* A, const float* B, const float* C, float* D) {
    const int lane = threadIdx.x;
    b8 a, b;
    for (int i = 0; i < 8; ++i) {
        const int k = 8 * (lane >> 5) + i;
        a[i] = (__bf16)A[(lane & 31) * 16 + k];
        b[i] = (__bf16)B[k * 32 + (lane & 31)];
    }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}

static float to_h(float x) { return (float)(_Float16)x; }
static float to_b(float x) { return (float)(__bf16)x; }

int main() {
    std::vector<float> A(32 * 16), B(16 * 32), C(32 * 32), D(32 * 32);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    auto run = [&](bool bf) {
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        if (bf) hipLaunchKernelGGL(probe_bf16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        else hipLaunchKernelGGL(probe_f16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    };
    // 1. subnormal f16 inputs: A = 2^-20 (subnormal in f16), B = 2^10 -> each product 2^-10, 16 of them -> 2^-6
    for (auto& v : A) v = ldexpf(1.f, -20);
    for (auto& v : B) v = ldexpf(1.f, 10);
    for (auto& v : C) v = 0.f;
    run(false);
    printf("f16 subnormal A (2^-20) x 2^10, K=16: got %.9g expected %.9g (0 => inputs flushed)\n", D[0], ldexpf(1.f, -6));
    for (auto& v : A) v = ldexpf(1.f, 10);
    for (auto& v : B) v = ldexpf(1.f, -20);
    run(false);
    printf("f16 subnormal B (2^-20) x 2^10, K=16: got %.9g expected %.9g\n", D[0], ldexpf(1.f, -6));
    // 2. random data with wide dynamic range: compare against the exact (double) dot product rounded once to fp32 and
    //    against a k-ordered fp32 fma chain
    for (int bf = 0; bf < 2; ++bf) {
        srand(7);
        double max_rel_exact = 0, max_rel_chain = 0, sum_rel = 0;
        int n = 0, bit_equal_chain = 0, bit_equal_exact = 0;
        for (int trial = 0; trial < 50; ++trial) {
            for (auto& v : A) { float x = ((rand() % 20001) - 10000) / 10000.f * ldexpf(1.f, (rand() % 17) - 8); v = bf ? to_b(x) : to_h(x); }
            for (auto& v : B) { float x = ((rand() % 20001) - 10000) / 10000.f * ldexpf(1.f, (rand() % 17) - 8); v = bf ? to_b(x) : to_h(x); }
            for (auto& v : C) v = ((rand() % 20001) - 10000) / 10000.f * 16.f;
            run(bf);
            for (int m = 0; m < 32; ++m)
                for (int nn = 0; nn < 32; ++nn) {
                    double ex = C[m * 32 + nn], mag = fabs(C[m * 32 + nn]);
                    float ch = C[m * 32 + nn];
                    for (int k = 0; k < 16; ++k) {
                        ex += (double)A[m * 16 + k] * B[k * 32 + nn];
                        mag += fabs((double)A[m * 16 + k] * B[k * 32 + nn]);
                        ch = fmaf(A[m * 16 + k], B[k * 32 + nn], ch);
                    }
                    const float got = D[m * 32 + nn];
                    max_rel_exact = fmax(max_rel_exact, fabs(got - ex) / mag);
                    max_rel_chain = fmax(max_rel_chain, fabs((double)got - ch) / mag);
                    sum_rel += fabs(got - ex) / mag;
                    bit_equal_chain += got == ch;
                    bit_equal_exact += got == (float)ex;
                    ++n;
                }
        }
        printf("%s K=16 with C: max |got-exact|/sum|ab| = %.3g (mean %.3g), vs fma chain %.3g; bit-equal to exact-rounded %d/%d, to chain %d/%d\n",
               bf ? "bf16" : "f16 ", max_rel_exact, sum_rel / n, max_rel_chain, bit_equal_exact, n, bit_equal_chain, n);
    }
    return 0;
}
