// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this engine uses
// (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of the bytes of a 16 B/lane stream; other widths uncalibrated).
// Each kernel streams a known number of bytes:  hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- /tmp/pmc_calib ; rocprofv3 --kernel-trace --pmc WRITE_SIZE -- /tmp/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy_b32(const float* __restrict__ a, float* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i] + 1.0f;
}
__global__ void copy_b128(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i];
        v.x += 1.0f;
        b[i] = v;
    }
}
__global__ void read_b32(const float* __restrict__ a, float* __restrict__ out, size_t n) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 12345.678f) out[0] = s;
}
int main() {
    const size_t n = (size_t)512 << 20;  // 512 Mi floats = 2 GiB per array (past the 256 MiB Infinity Cache)
    float *a, *b;
    hipMalloc(&a, n * 4);
    hipMalloc(&b, n * 4);
    hipMemset(a, 0, n * 4);
    hipMemset(b, 0, n * 4);
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(copy_b32, dim3(4096), dim3(256), 0, 0, a, b, n);
        hipLaunchKernelGGL(copy_b128, dim3(4096), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4);
        hipLaunchKernelGGL(read_b32, dim3(4096), dim3(256), 0, 0, a, b, n);
    }
    hipDeviceSynchronize();
    printf("bytes per array: %zu\n", n * 4);
    return 0;
}
