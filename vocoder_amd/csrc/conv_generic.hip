// Fallback for (kernel size, dilation) pairs without a compile-time specialisation: same structure as
// conv_mfma_kernel (LDS-staged activation window, packed weights from L2, fp32 MFMA) but with runtime tap count /
// dilation, so tap offsets are computed instead of folded into immediates.  Tile 64 x 128 only.
#include "conv_mfma_impl.h"

namespace fv {

__global__ __launch_bounds__(256) void conv_mfma_generic_kernel(const ConvParams p) {
    constexpr int MT = 2;
    constexpr int N_BLK = 128;
    extern __shared__ __attribute__((aligned(16))) float xs_dyn[];
    const int KS = p.ks, DIL = p.dil;
    const int W = N_BLK + (KS - 1) * DIL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);

    int bid = blockIdx.x;
    const int n_tile = bid % p.n_tiles;
    bid /= p.n_tiles;
    const int m_blk = bid % p.m_blks;
    const int b = bid / p.m_blks;
    const int n0 = n_tile * N_BLK;
    const float* __restrict__ xb = p.x + (long long)b * p.x_bstride;
    const int tbase = n0 - p.pad_l;

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int mt0 = m_blk * MT;
    const float4* __restrict__ wbase = p.wp + lane;
    const int b_lane = (lane >> 5) * W + wn * 32 + (lane & 31);
    const int total = kChunk * W;

    for (int c = 0; c < p.nchunk_real; ++c) {
        float* xsb = xs_dyn + (c & 1) * total;
        for (int e = tid; e < total; e += 256) {
            const int r = e / W, col = e - r * W;
            const int ci = c * kChunk + r, t = tbase + col;
            float v = 0.f;
            if (ci < p.Cin && t >= 0 && t < p.Tin) v = act_apply(xb[(long long)ci * p.Tin + t], p.pre_act, p.slope);
            xsb[e] = v;
        }
        __syncthreads();
        for (int j = 0; j < KS; ++j) {
            float4 a[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = wbase[((long long)((mt0 + i) * p.nchunk + c) * KS + j) * 64];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const float bv = xsb[b_lane + (2 * pp) * W + j * DIL];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float av = pp == 0 ? a[i].x : pp == 1 ? a[i].y : pp == 2 ? a[i].z : a[i].w;
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                }
            }
        }
    }

    const int n = n0 + wn * 32 + (lane & 31);
    float* __restrict__ yb = p.y + (long long)b * p.y_bstride;
    const float* __restrict__ rb = p.res ? p.res + (long long)b * p.y_bstride : nullptr;
    if (n >= p.N) return;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (mt0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= p.M) continue;
            long long o;
            if (p.convt) {
                const int co = m / p.u, ph = m - co * p.u;
                const int t = n * p.u + ph - p.pad_t;
                if (t < 0 || t >= p.Tout) continue;
                o = (long long)co * p.Tout + t;
            } else {
                o = (long long)m * p.N + n;
            }
            float v = (acc[i][r] + p.bias[m]) * (p.gamma ? p.gamma[m] : 1.0f);
            if (rb) v += rb[o];
            v = act_apply(v, p.post_act, p.slope);
            if (p.out_mode == OUT_ACCUM) v = (yb[o] + v) * p.out_scale;
            yb[o] = v;
        }
    }
}

bool launch_conv_generic(const ConvParams& p, int cfg, int batch, hipStream_t s, size_t* lds_bytes) {
    (void)cfg;
    const int W = 128 + (p.ks - 1) * p.dil;
    const size_t lds = (size_t)2 * kChunk * W * sizeof(float);
    if (lds_bytes) *lds_bytes = lds;
    if (lds > 64 * 1024) return false;
    const int grid = batch * p.m_blks * p.n_tiles;
    hipLaunchKernelGGL(conv_mfma_generic_kernel, dim3(grid), dim3(256), lds, s, p);
    return true;
}

}  // namespace fv
