// EXPERIMENT (round 3), NOT part of libfishvoc_hip.so.  Needs conv_mfma_kernel's body as a device function
// `conv_mfma_tile<KS, DIL, WM, WN, MT, NT, SUM3>(p, tile_id, lds)` (a mechanical, resource-neutral refactor: 147 VGPRs before
// and after) plus `conv_fused3_run` / `fv_debug_conv_fused3` host glue; harness: probe_fused3.py.  Result (B = 32, bit-identical
// outputs, medians of 7 interleaved rounds): the shared tile queue is 10 - 15 % SLOWER than three launches on three streams
// (C = 128: 1.19 vs 1.08 ms per round of the three convs; C = 256: 0.66 vs 0.57; C = 64: 0.68 vs 0.59), with three or with two
// resident workgroups per CU (the three bodies in one kernel spill 220 - 280 B / lane at the 168-register cap).  profiles/LOG.md R3.12.
// EXPERIMENT (round 3): the same-depth convs of a stage's three ResBlock branches (k = 3 / 7 / 11, equal dilation,
// equal C) as ONE persistent launch over a shared tile queue, longest tiles first (k = 11, then 7, then 3): no per-kernel tail
// (1376 tiles on 768 workgroup slots = 1.8 rounds per kernel today) and no chip-wide lock-step between three launches.
#include "conv_mfma_impl.h"
#ifndef FV_X_F3_WAVES
#define FV_X_F3_WAVES 3
#endif

namespace fv {

struct Fused3Params {
    ConvParams p[3];     // p[0]: the longest (k = 11) ... p[2]: the shortest (k = 3)
    int n[3];            // tiles of each
    int total;
    int* counter;        // zeroed by the host before the launch
};

template <int K0, int K1, int K2, int DIL, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256, FV_X_F3_WAVES) void conv_fused3_kernel(const Fused3Params f) {
    constexpr int L0 = conv_mfma_lds_floats<K0, DIL, WN, NT>(), L1 = conv_mfma_lds_floats<K1, DIL, WN, NT>(), L2 = conv_mfma_lds_floats<K2, DIL, WN, NT>();
    constexpr int LMAX = L0 > L1 ? (L0 > L2 ? L0 : L2) : (L1 > L2 ? L1 : L2);
    __shared__ float xs[LMAX];
    __shared__ int s_tile;
    for (;;) {
        __syncthreads();   // every wave is out of the previous tile (its LDS reads, its copy of s_tile)
        if (threadIdx.x == 0) s_tile = atomicAdd(f.counter, 1);
        __syncthreads();
        const int q = __builtin_amdgcn_readfirstlane(s_tile);
        if (q >= f.total) break;
        if (q < f.n[0]) conv_mfma_tile<K0, DIL, WM, WN, MT, NT, false>(f.p[0], q, xs);
        else if (q < f.n[0] + f.n[1]) conv_mfma_tile<K1, DIL, WM, WN, MT, NT, false>(f.p[1], q - f.n[0], xs);
        else conv_mfma_tile<K2, DIL, WM, WN, MT, NT, false>(f.p[2], q - f.n[0] - f.n[1], xs);
    }
}

// three stride-1 'same' convs of equal (C_in, C_out, dilation) with k = 11 / 7 / 3; cfg: TILE_128x128 or TILE_128x64 or TILE_64x256
bool launch_conv_fused3(const ConvParams (&p)[3], int cfg, int dil, int batch, int* d_counter, hipStream_t s) {
    Fused3Params f;
    for (int i = 0; i < 3; ++i) {
        f.p[i] = p[i];
        f.n[i] = batch * p[i].m_blks * p[i].n_tiles;
    }
    f.total = f.n[0] + f.n[1] + f.n[2];
    f.counter = d_counter;
    const int grid = std::min(f.total, num_cus() * 3);
#define FV_F3(D, WM, WN, MT, NT) hipLaunchKernelGGL((conv_fused3_kernel<11, 7, 3, D, WM, WN, MT, NT>), dim3(grid), dim3(256), 0, s, f); return true;
#define FV_F3D(WM, WN, MT, NT) \
    if (dil == 1) { FV_F3(1, WM, WN, MT, NT) } if (dil == 3) { FV_F3(3, WM, WN, MT, NT) } if (dil == 5) { FV_F3(5, WM, WN, MT, NT) } return false;
    if (cfg == TILE_128x128) { FV_F3D(2, 2, 2, 2) }
    if (cfg == TILE_128x64) { FV_F3D(4, 1, 1, 2) }
    if (cfg == TILE_64x256) { FV_F3D(1, 4, 2, 2) }
    return false;
}

}  // namespace fv
