// Opt-in "f16x3" precision mode of the fused conv layer: the same implicit GEMM as conv_mfma_impl.h, but on the packed
// half-precision matrix cores of gfx950 (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate) with every fp32 operand split
// in two fp16 planes and three products per k-block, so that the result keeps fp32-class accuracy (~2^-22 relative per
// product, fp32 accumulation):
//
//     x = xh + xl * 2^-11        xh = fp16(x),  xl = fp16((x - xh) * 2^11)           (activation, split while staging)
//     w = wh + wl                wh = fp16(w * s_w),  wl = fp16(w * s_w - wh)         (weights, split on the host)
//     x * w * s_w  ~=  xh*wh + xh*wl + xl*(wh * 2^-11)                                (xl*wl ~ 2^-22 |x w| is dropped)
//
// s_w is a per-layer power of two that puts max|w| near 2^14, so wl and wh*2^-11 stay normal fp16 numbers for every weight
// within 2^-17 of the largest one; the activation residual is scaled by 2^11 instead (its magnitude is not known ahead
// of time) and the matching weight plane carries the 2^-11.  All three products therefore share one scale and one set
// of fp32 accumulators; the epilogue multiplies by 1/s_w.  fp16 subnormal inputs are honoured by the MFMA (measured,
// tools/ubench/mfma_f16_numerics.hip), so small activations degrade gracefully (absolute error <= 2^-35).
// Range contract: |activation| < 65504 after the pre-activation (larger values become inf and poison the output).
//
// Layout differences from the fp32 kernel:
//   * K runs over 16-channel chunks; lane l of a wave supplies k = 8 * (l >> 5) .. +7 (8 consecutive channels).
//   * LDS window: [plane][k-half][column] x 16 B (8 channels of one column), so a B fragment is one ds_read_b128 per
//     lane at lane-consecutive 16-byte slots (conflict-free) and taps are immediate-offset shifts, as before.
//   * Weights: host-packed [m_tile][chunk16][tap][plane (wh, wl, wh*2^-11)][lane] x 16 B, fetched from L2 with
//     SGPR-addressed raw buffer loads, prefetched DA k-blocks ahead.
//   * Wave tile 32 x (NT*32): waves are stacked along M so no two waves of a workgroup fetch the same weights.
#pragma once

#include "conv_mfma_impl.h"

namespace fv {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int kChunk16 = 16;
constexpr int kF16Prefetch = 2;   // weight prefetch distance in k-blocks (one block = 16 channels x 1 tap = 3*NT MFMAs)

template <int KS, int DIL, int WM, int WN, int NT>
__global__ __launch_bounds__(256, 2) void conv_f16x3_kernel(const ConvParams p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(NT % 2 == 0, "B fragments are loaded two n-tiles at a time");
    constexpr int N_BLK = WN * NT * 32;
    constexpr int SPAN = (KS - 1) * DIL;
    constexpr int W = N_BLK + SPAN;
    constexpr int ITEMS = 2 * W;                       // (k-half, column) staging items of 8 channels each
    constexpr int NE = (ITEMS + 255) / 256;
    constexpr int PLANE = 2 * W;                       // 16-byte slots per plane
    __shared__ h8 xs[2][2 * PLANE];                    // [buffer][plane][k-half][column]
    static_assert(sizeof(h8) == 16, "h8 is one 16-byte LDS slot");

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

    f32x16 acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    // ---- staging plan: item e = tid + i*256 -> (k-half h, column col); eight channel rows 8h .. 8h+7 of the chunk ----
    // st_off = byte offset of (row 8h, t) inside the chunk, or a marker >= 0xC0000000 when t is outside [0, Tin): adding
    // the row offsets (< 2^30) cannot wrap it, and the raw buffer load returns 0 beyond the descriptor's span.
    unsigned st_off[NE];
    const int tbase = n0 - p.pad_l;
    const unsigned row_b = (unsigned)p.Tin * 4u;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        int e = tid + i * 256;
        const bool in_tile = e < ITEMS;
        e = in_tile ? e : ITEMS - 1;
        const int h = e / W;
        const int col = e - h * W;
        const int t = tbase + col;
        const bool ok = in_tile && t >= 0 && t < p.Tin;
        st_off[i] = ok ? (unsigned)(8 * h * p.Tin + t) * 4u : 0xC0000000u;
    }
    float stage[NE][8];
    auto load_chunk = [&](int c) {
        const int cbase = c * kChunk16;
        const long long rows = (long long)(p.Cin - cbase) * p.Tin;   // rows of zero-padded channels read as 0
        const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(xb + (long long)cbase * p.Tin, (unsigned)(rows * 4));
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int r = 0; r < 8; ++r)
                stage[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, st_off[i] + (unsigned)r * row_b, 0, 0));
    };
    auto store_chunk = [&](h8* dst) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256;
            h8 hi, lo;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float v = stage[i][r];
                if (p.pre_act == FV_ACT_SILU) {
                    v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                } else if (p.pre_act != FV_ACT_NONE) {
                    v = act_apply(v, p.pre_act, p.slope);
                }
                const _Float16 vh = (_Float16)v;
                hi[r] = vh;
                lo[r] = (_Float16)((v - (float)vh) * 2048.0f);
            }
            if (e < ITEMS) {
                dst[e] = hi;
                dst[PLANE + e] = lo;
            }
        }
    };

    // ---- weights: k-block g = chunk * KS + tap of m-tile mt starts at byte ((mt * nch16 * KS) + g) * 3072 ----
    const int mt0 = m_blk * WM + wm;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wph, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    const int wbase = __builtin_amdgcn_readfirstlane(mt0 * p.nch16 * KS * 3072);
    auto load_a = [&](h8 (&dst)[3], int goff_b) {   // goff_b = g * 3072, wave-uniform
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wbase + goff_b + q * 1024, 0);
            dst[q] = __builtin_bit_cast(h8, v);
        }
    };
    // ---- activation fragments: two n-tiles x (xh, xl) per group ----
    const int b_lane = (lane >> 5) * W + wn * (NT * 32) + (lane & 31);
    auto load_bgrp = [&](h8 (&dst)[2][2], const h8* xsb, int j, int grp) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 2; ++q) dst[u][q] = xsb[q * PLANE + b_lane + (grp * 2 + u) * 32 + j * DIL];
    };

    constexpr int DA = kF16Prefetch;
    constexpr int NG = NT / 2;
    h8 aq[DA + 1][3];
    h8 bq[2][2][2];
    const int nch = p.nch16_real;
    load_chunk(0);
#pragma unroll
    for (int d = 0; d < DA; ++d) load_a(aq[d], d * 3072);
    for (int c = 0; c < nch; ++c) {
        h8* xsb = xs[c & 1];
        store_chunk(xsb);
        __syncthreads();
        if (c + 1 < nch) load_chunk(c + 1);
        const int gchunk_b = __builtin_amdgcn_readfirstlane((c * KS + DA) * 3072);
        load_bgrp(bq[0], xsb, 0, 0);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            load_a(aq[DA], gchunk_b + j * 3072);
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
                const int cur_idx = (j * NG + grp) & 1;   // compile-time after unrolling
                if (grp + 1 < NG) load_bgrp(bq[cur_idx ^ 1], xsb, j, grp + 1);
                else if (j + 1 < KS) load_bgrp(bq[cur_idx ^ 1], xsb, j + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int jn = grp * 2 + u;
                    acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[0][0], bq[cur_idx][u][0], acc[0][jn], 0, 0, 0);
                    acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[0][1], bq[cur_idx][u][0], acc[0][jn], 0, 0, 0);
                    acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[0][2], bq[cur_idx][u][1], acc[0][jn], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int d = 0; d < DA; ++d)
#pragma unroll
                for (int q = 0; q < 3; ++q) aq[d][q] = aq[d + 1][q];
        }
    }

    // epilogue in two column halves (bounds the live registers): acc * 1/s_w + bias ...
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        f32x16(&sub)[1][NT / 2] = reinterpret_cast<f32x16(&)[1][NT / 2]>(acc[0][hf * (NT / 2)]);
        conv_epilogue<1, NT / 2>(p, sub, b, mt0, n0 + wn * (NT * 32) + hf * (NT / 2) * 32 + (lane & 31), lane);
    }
}

template <int KS, int DIL>
inline bool launch_f16x3_cfg(const ConvParams& p, int cfg, int batch, hipStream_t s) {
    const int grid = batch * p.m_blks * p.n_tiles;
    switch (cfg) {
        case SPLIT_128x256: hipLaunchKernelGGL((conv_f16x3_kernel<KS, DIL, 4, 1, 8>), dim3(grid), dim3(256), 0, s, p); return true;
        case SPLIT_128x128: hipLaunchKernelGGL((conv_f16x3_kernel<KS, DIL, 4, 1, 4>), dim3(grid), dim3(256), 0, s, p); return true;
        case SPLIT_64x256: hipLaunchKernelGGL((conv_f16x3_kernel<KS, DIL, 2, 2, 4>), dim3(grid), dim3(256), 0, s, p); return true;
        default: return false;
    }
}

}  // namespace fv
