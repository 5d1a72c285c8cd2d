// Fused   pre-activation -> (dilated) Conv1d / polyphase ConvTranspose1d -> bias [-> scale] [-> +residual]
//         [-> post-activation] [-> MRF accumulate]
// as an implicit GEMM on the exact-fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces, per launch, what the reference runs as 3-5 separate torch kernels with full HBM round-trips:
// F.silu -> weight_norm -> aten::convolution -> add (fish_vocoder/modules/generators/hifigan.py:101-108).
//
// Mapping (one workgroup = 4 wavefronts of 64 lanes, output tile M_BLK x N_BLK):
//   GEMM M = output rows (C_out, or (C_out, phase) for the transposed conv), N = time, K = C_in * taps.
//   B operand (activations): an 8-channel x (N_BLK + (KS-1)*DIL) window of x is staged ONCE per chunk into LDS
//     (activation applied on the way in), and every tap j reads it at a shifted address -> KS-fold reuse out of LDS,
//     reads are lane-consecutive (bank-conflict free ds_read_b32 with immediate offsets).
//   A operand (weights): pre-packed on the host in MFMA-fragment order so that each lane fetches its four
//     k-steps of a tap with one coalesced global_load_dwordx4 straight from L2 (weights are L2-resident; no LDS).
//   K ordering inside a chunk: tap-major, then channel pair p; MFMA k-half h (lane>>5) = channel 2p+h.
//   Accumulators: MT x NT tiles of 32x32 fp32 (16 VGPRs each) per wave.
#pragma once

#include "fv_internal.h"

namespace fv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case FV_ACT_SILU: return v * __frcp_rn(1.0f + __expf(-v));
        case FV_ACT_LEAKY_RELU: return v >= 0.f ? v : v * slope;
        case FV_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case FV_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

template <int KS, int DIL, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvParams p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int M_BLK = WM * MT * 32;  (void)M_BLK;
    constexpr int N_BLK = WN * NT * 32;
    constexpr int SPAN = (KS - 1) * DIL;
    constexpr int W = N_BLK + SPAN;                    // staged columns per channel row
    constexpr int NE = (kChunk * W + 255) / 256;       // staged elements per thread
    __shared__ float xs[2][kChunk * W];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int bid = blockIdx.x;
    const int n_tile = bid % p.n_tiles;
    bid /= p.n_tiles;
    const int m_blk = bid % p.m_blks;
    const int b = bid / p.m_blks;
    const int n0 = n_tile * N_BLK;
    const float* __restrict__ xb = p.x + (long long)b * p.x_bstride;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float stage[NE];
    const int tbase = n0 - p.pad_l;
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256;
            const int r = e / W;
            const int col = e - r * W;
            const int ci = c * kChunk + r;
            const int t = tbase + col;
            const bool ok = (e < kChunk * W) && (ci < p.Cin) && (t >= 0) && (t < p.Tin);
            stage[i] = ok ? xb[(long long)ci * p.Tin + t] : 0.f;
        }
    };
    auto store_chunk = [&](float* dst) {
        if (p.pre_act == FV_ACT_SILU) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int e = tid + i * 256;
                const float v = stage[i];
                if (e < kChunk * W) dst[e] = v * __frcp_rn(1.0f + __expf(-v));
            }
        } else if (p.pre_act == FV_ACT_NONE) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int e = tid + i * 256;
                if (e < kChunk * W) dst[e] = stage[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int e = tid + i * 256;
                // act(0) == 0 for every supported activation, so zero padding commutes with it
                if (e < kChunk * W) dst[e] = act_apply(stage[i], p.pre_act, p.slope);
            }
        }
    };

    // A-operand base: packed as [m_tile][chunk][tap][lane] float4 (4 channel pairs)
    const int mt0 = (m_blk * WM + wm) * MT;
    const float4* __restrict__ wbase = p.wp + lane;
    const int b_lane = (lane >> 5) * W + wn * (NT * 32) + (lane & 31);

    load_chunk(0);
    for (int c = 0; c < p.nchunk; ++c) {
        float* xsb = xs[c & 1];
        store_chunk(xsb);
        __syncthreads();
        if (c + 1 < p.nchunk) load_chunk(c + 1);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            float4 a[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                a[i] = wbase[((long long)((mt0 + i) * p.nchunk + c) * KS + j) * 64];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                float bv[NT];
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) bv[jn] = xsb[b_lane + (2 * pp) * W + jn * 32 + j * DIL];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float av = pp == 0 ? a[i].x : pp == 1 ? a[i].y : pp == 2 ? a[i].z : a[i].w;
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[jn], acc[i][jn], 0, 0, 0);
                }
            }
        }
    }

    // Epilogue.  C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    const int ncol0 = n0 + wn * (NT * 32) + (lane & 31);
    float* __restrict__ yb = p.y + (long long)b * p.y_bstride;
    const float* __restrict__ rb = p.res ? p.res + (long long)b * p.y_bstride : nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (mt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= p.M) continue;
            const float bias = p.bias[m];
            const float gm = p.gamma ? p.gamma[m] : 1.0f;
            long long row_off;
            int cstride, cbase;
            if (p.convt) {
                const int co = m / p.u, ph = m - co * p.u;
                row_off = (long long)co * p.Tout;
                cstride = p.u;
                cbase = ph - p.pad_t;
            } else {
                row_off = (long long)m * p.N;
                cstride = 1;
                cbase = 0;
            }
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) {
                const int n = ncol0 + jn * 32;
                if (n >= p.N) continue;
                const int t = n * cstride + cbase;
                if (p.convt && (t < 0 || t >= p.Tout)) continue;
                const long long o = row_off + t;
                float v = (acc[i][jn][r] + bias) * gm;
                if (rb) v += rb[o];
                v = act_apply(v, p.post_act, p.slope);
                if (p.out_mode == OUT_ACCUM) v = (yb[o] + v) * p.out_scale;
                yb[o] = v;
            }
        }
    }
}

template <int KS, int DIL, int WM, int WN, int MT, int NT>
inline void launch_one(const ConvParams& p, int batch, hipStream_t s) {
    const int grid = batch * p.m_blks * p.n_tiles;
    hipLaunchKernelGGL((conv_mfma_kernel<KS, DIL, WM, WN, MT, NT>), dim3(grid), dim3(256), 0, s, p);
}

template <int KS, int DIL>
inline bool launch_cfg(const ConvParams& p, int cfg, int batch, hipStream_t s) {
    switch (cfg) {
        case TILE_128x128: launch_one<KS, DIL, 2, 2, 2, 2>(p, batch, s); return true;
        case TILE_64x256: launch_one<KS, DIL, 1, 4, 2, 2>(p, batch, s); return true;
        case TILE_32x512: launch_one<KS, DIL, 1, 4, 1, 4>(p, batch, s); return true;
        case TILE_128x64: launch_one<KS, DIL, 4, 1, 1, 2>(p, batch, s); return true;
        case TILE_32x128: launch_one<KS, DIL, 1, 4, 1, 1>(p, batch, s); return true;
        case TILE_64x128: launch_one<KS, DIL, 1, 4, 2, 1>(p, batch, s); return true;
        default: return false;
    }
}

}  // namespace fv
