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
//   * Weights: host-packed [m_tile][chunk16][tap][plane (wh, wl)][lane] x 16 B (the third operand wh*2^-11 is derived in registers), fetched from L2 with
//     SGPR-addressed raw buffer loads, prefetched DA k-blocks ahead.
//   * Wave tile 32 x (NT*32): waves are stacked along M so no two waves of a workgroup fetch the same weights.
#pragma once

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "conv_mfma_impl.h"

namespace fv {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// Epilogue of the split kernel: same arithmetic as conv_epilogue (non-transposed layers), but the residual operand of the
// WHOLE register tile is requested before anything is stored — one HBM round trip per tile instead of one per 4-row group
// (res may alias y, so the compiler cannot hoist those loads across the stores itself).  The split kernel's MFMA phase is
// ~5x shorter than the fp32 kernel's, which makes that latency visible.
template <int NTE>
__device__ __forceinline__ void conv_epilogue_bulk(const ConvParams& p, f32x16 (&acc)[NTE], int b, int mt, int ncol0, int lane) {
    // flat mode (pointwise convs): the column axis runs over (batch item, t) and the descriptor spans every item
    const unsigned span = (unsigned)((p.flat ? (long long)p.y_bstride * (p.n_total / p.N) : p.y_bstride) * 4);
    const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * p.y_bstride, span);
    const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(p.res ? p.res + (long long)b * p.y_bstride : p.y, span);
    const bool has_res = p.res != nullptr;
    const bool accum = p.out_mode == OUT_ACCUM;
    int coff[NTE];   // element offset of the column (row part excluded), INT_MIN when the column does not exist
#pragma unroll
    for (int jn = 0; jn < NTE; ++jn) {
        const int n = ncol0 + jn * 32;
        if (p.convt) {
            coff[jn] = n < p.N ? n * p.u - p.pad_t : INT_MIN;   // polyphase: column q lands at t = q*u - padding + phase
        } else if (p.flat) {
            const int bb = n / p.N;
            coff[jn] = n < p.n_total ? bb * (int)p.y_bstride + (n - bb * p.N) : INT_MIN;
        } else {
            coff[jn] = n < p.N ? n : INT_MIN;
        }
    }
    auto offset = [&](int r, int jn) -> unsigned {   // byte offset of accumulator register r of n-tile jn, or the OOB marker
        const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (p.convt) {   // row m = (c_out, phase)
            const int co = m / p.u, ph = m - co * p.u;
            const int t = coff[jn] + ph;
            return (m < p.M && coff[jn] != INT_MIN && t >= 0 && t < p.Tout) ? (unsigned)(co * p.Tout + t) * 4u : 0xFFFFFFFFu;
        }
        return (m < p.M && coff[jn] != INT_MIN) ? (unsigned)(m * p.N + coff[jn]) * 4u : 0xFFFFFFFFu;
    };
    float rv[NTE][16];
    if (has_res) {
#pragma unroll
        for (int jn = 0; jn < NTE; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[jn][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, offset(r, jn), 0, 0));
    }
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int mc = m < p.M ? m : 0;
        // layer scale (ConvNeXt gamma, applied to conv + bias before the residual) folds into scale and bias
        const float g = p.gamma ? p.gamma[mc] : 1.0f;
        bias[r] = p.bias[mc] * g;
        if (p.gamma) {
#pragma unroll
            for (int jn = 0; jn < NTE; ++jn) acc[jn][r] *= g;
        }
    }
#pragma unroll
    for (int jn = 0; jn < NTE; ++jn) {
        float val[16], yo[16];
        if (accum) {   // MRF accumulate (2 of 18 launches per stage): one extra round trip per n-tile
#pragma unroll
            for (int r = 0; r < 16; ++r) yo[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, offset(r, jn), 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            val[r] = fmaf(acc[jn][r], p.acc_scale, bias[r]);
            if (has_res) val[r] += rv[jn][r];
        }
        act_apply_all(val, p.post_act, p.slope);
        if (accum) {
#pragma unroll
            for (int r = 0; r < 16; ++r) val[r] = (yo[r] + val[r]) * p.out_scale;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[r]), yrs, offset(r, jn), 0, 0);
    }
}

constexpr int kChunk16 = 16;
constexpr int kF16WeightPrefetch = 2;   // weight prefetch distance in k-blocks (one block = 16 channels x 1 tap = 3*NT MFMAs)
// 16-channel sub-chunks staged per barrier: pointwise convs have one k-block per sub-chunk, so they stage four
constexpr int f16_subs_for(int ks) { return ks == 1 ? 4 : ((ks == 2 || ks == 4) ? 2 : 1); }

template <int KS, int DIL, int WM, int WN, int NT>
__global__ __launch_bounds__(256, 2) void conv_f16x3_kernel(const ConvParams p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(NT % 2 == 0, "B fragments are loaded two n-tiles at a time");
    constexpr int N_BLK = WN * NT * 32;
    constexpr int SPAN = (KS - 1) * DIL;
    constexpr int W = N_BLK + SPAN;
    constexpr int SUBS = f16_subs_for(KS);
    constexpr int ITEMS = SUBS * 2 * W;                // (sub-chunk, k-half, column) staging items of 8 channels each
    constexpr int NE = (ITEMS + 255) / 256;
    constexpr int PLANE = 2 * W;                       // 16-byte slots per plane of one sub-chunk
    constexpr int SUB_SLOTS = 2 * PLANE;               // (xh, xl) planes
    __shared__ h8 xs[2][SUBS * SUB_SLOTS];             // [buffer][sub-chunk][plane][k-half][column]
    static_assert(sizeof(h8) == 16, "h8 is one 16-byte LDS slot");
    static_assert(sizeof(xs) <= 65536, "static LDS limit");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int bid = blockIdx.x;
    const int n_tile = bid % p.n_tiles;
    bid /= p.n_tiles;
    const int m_blk = bid % p.m_blks;
    const int b = bid / p.m_blks;                      // 0 in flat mode
    const int n0 = n_tile * N_BLK;
    const float* __restrict__ xb = p.x + (long long)b * p.x_bstride;
    const bool flat = KS == 1 && p.flat;

    f32x16 acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    // ---- staging plan: item e = tid + i*256 -> (sub-chunk, k-half h, column col); eight channel rows of the chunk ----
    // st_off = byte offset of (row 16*sub + 8h, t) relative to the chunk's first row, or a marker >= 0xC0000000 when the
    // column does not exist: adding the row offsets (< 2^30) cannot wrap it, and the raw buffer load returns 0 beyond the
    // descriptor's span (which also zero-fills the channels past C_in of the last chunk).
    unsigned st_off[NE];
    const int tbase = n0 - p.pad_l;
    const unsigned row_b = (unsigned)p.Tin * 4u;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        int e = tid + i * 256;
        const bool in_tile = e < ITEMS;
        e = in_tile ? e : ITEMS - 1;
        const int sh = e / W;                          // sub * 2 + h
        const int col = e - sh * W;
        int t = tbase + col;
        int boff = 0;
        bool ok = in_tile;
        if (flat) {
            ok = ok && t < p.n_total;
            const int bb = t / p.N;
            t -= bb * p.N;
            boff = bb * (int)p.x_bstride;
        }
        ok = ok && t >= 0 && t < p.Tin;
        st_off[i] = ok ? (unsigned)(boff + 8 * sh * p.Tin + t) * 4u : 0xC0000000u;
    }
    float stage[NE][8];
    __amdgpu_buffer_rsrc_t xrs;
    const long long x_items = flat ? (long long)(p.n_total / p.N) : 1;   // batch items spanned by the descriptor
    auto chunk_rsrc = [&](int c) {
        const int cbase = c * kChunk16 * SUBS;
        // flat mode: the span covers every batch item (the host guarantees C_in is a multiple of the chunk, so no row of
        // another item can be mistaken for a padded channel); otherwise it ends with this item's last channel
        const long long elems = flat ? x_items * p.x_bstride - (long long)cbase * p.Tin : (long long)(p.Cin - cbase) * p.Tin;
        xrs = uniform_rsrc(xb + (long long)cbase * p.Tin, (unsigned)(elems * 4));
    };
    auto load_item = [&](int i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            stage[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, st_off[i] + (unsigned)r * row_b, 0, 0));
    };
    const bool silu = p.pre_act == FV_ACT_SILU;   // the host only dispatches this kernel with pre_act in {NONE, SILU}
    // activate, split, pack.  Kept element-wise on purpose: a two-at-a-time formulation (v_cvt_pk_f16_f32 on float pairs,
    // halves re-read with v_cvt_f32_f16_sdwa) compiled by hipcc 7.2 produced timing-dependent wrong splits on gfx950 —
    // a VALU-write -> SDWA-read hazard one wait state short — and was no faster.
    auto store_chunk = [&](h8* dst) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256;
            h8 hi, lo;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float v = stage[i][r];
                if (silu) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                const _Float16 vh = (_Float16)v;
                hi[r] = vh;
                lo[r] = (_Float16)((v - (float)vh) * 2048.0f);
            }
            if (e < ITEMS) {
                // item e = (sub * 2 + h) * W + col  ->  slot sub * SUB_SLOTS + plane * PLANE + h * W + col
                const int sub = e / PLANE;
                const int slot = e + sub * PLANE;
                dst[slot] = hi;
                dst[slot + PLANE] = lo;
            }
        }
    };

    // ---- weights: k-block g = chunk16 * KS + tap of m-tile mt starts at byte ((mt * nch16 * KS) + g) * 2048 ----
    const int mt0 = m_blk * WM + wm;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wph, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    const int wbase = __builtin_amdgcn_readfirstlane(mt0 * p.nch16 * KS * 2048);
    auto load_a = [&](h8 (&dst)[2], int goff_b) {   // goff_b = g * 2048, wave-uniform
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // voffset carries the constant plane offset so that it folds into the instruction's immediate field: one
            // SALU add per k-block instead of three
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff + q * 1024, wbase + goff_b, 0);
            dst[q] = __builtin_bit_cast(h8, v);
        }
    };
    // ---- activation fragments: two n-tiles x (xh, xl) per group ----
    // request order = reverse of the order of first use (xh of n-tile 0 is consumed first): LDS returns in order, so the wait
    // before the first MFMA of a group covers the whole group and the other three waits disappear
    const int b_lane = (lane >> 5) * W + wn * (NT * 32) + (lane & 31);
    auto load_bgrp = [&](h8 (&dst)[2][2], const h8* xsb, int kb, int grp) {   // kb = k-block inside the chunk
        const int sub = kb / KS, j = kb - sub * KS;
#pragma unroll
        for (int q = 1; q >= 0; --q)
#pragma unroll
            for (int u = 1; u >= 0; --u)
                dst[u][q] = xsb[sub * SUB_SLOTS + q * PLANE + b_lane + (grp * 2 + u) * 32 + j * DIL];
    };

    constexpr int DA = kF16WeightPrefetch;
    constexpr int KB = SUBS * KS;         // k-blocks per LDS chunk
    constexpr int NG = NT / 2;
    constexpr int G = KB * NG;            // fragment groups per chunk
    constexpr int RA = DA + 1;            // weight-fragment ring: k-block kb of a chunk lives in slot kb % RA
    h8 aq[RA][2];   // (wh, wl); the third operand wh * 2^-11 is derived in registers right before its MFMAs
    h8 bq[2][2][2];
    const int nch = (p.nch16_real + SUBS - 1) / SUBS;   // LDS chunks; the packed planes are padded to whole chunks
    chunk_rsrc(0);
#pragma unroll
    for (int i = 0; i < NE; ++i) load_item(i);
#pragma unroll
    for (int d = 0; d < DA; ++d) load_a(aq[d], d * 2048);
    for (int c = 0; c < nch; ++c) {
        h8* xsb = xs[c & 1];
        store_chunk(xsb);
        __syncthreads();
        const bool more = c + 1 < nch;
        if (more) {
            chunk_rsrc(c + 1);
#pragma unroll
            for (int i = 0; i < NE; ++i) load_item(i);
        }
        const int gchunk_b = __builtin_amdgcn_readfirstlane((c * KB + DA) * 2048);
        load_bgrp(bq[0], xsb, 0, 0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            load_a(aq[(kb + DA) % RA], gchunk_b + kb * 2048);
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
                const int sidx = kb * NG + grp;            // compile-time after unrolling
                const int cur = sidx & 1;
                if (sidx + 1 < G) load_bgrp(bq[cur ^ 1], xsb, (sidx + 1) / NG, (sidx + 1) % NG);
                __builtin_amdgcn_sched_barrier(0);
                // product-major order: consecutive MFMAs never share an accumulator
                const h8 a_sc = aq[kb % RA][0] * (_Float16)(1.0f / 2048.0f);   // exact (power of two): 4 v_pk_mul_f16
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int jn = grp * 2 + u;
                        acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q == 2 ? a_sc : aq[kb % RA][q], bq[cur][u][q == 2 ? 1 : 0], acc[0][jn], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the DA fragments in flight for the next chunk sit in slots (KB + d) % RA: move them to slots d (once per chunk;
        // a per-tap rotation cost 2 v_mov per MFMA)
        if (KB % RA != 0) {
            h8 t[DA][2];
#pragma unroll
            for (int d = 0; d < DA; ++d)
#pragma unroll
                for (int q = 0; q < 2; ++q) t[d][q] = aq[(KB + d) % RA][q];
#pragma unroll
            for (int d = 0; d < DA; ++d)
#pragma unroll
                for (int q = 0; q < 2; ++q) aq[d][q] = t[d][q];
        }
    }

    // epilogue in groups of four n-tiles (two for the 128-accumulator tile: bounds the live registers)
    constexpr int EG = (NT >= 8 || NT < 4) ? 2 : 4;
#pragma unroll
    for (int hf = 0; hf < NT / EG; ++hf) {
        f32x16(&sub)[EG] = reinterpret_cast<f32x16(&)[EG]>(acc[0][hf * EG]);
        conv_epilogue_bulk<EG>(p, sub, b, mt0, n0 + wn * (NT * 32) + hf * EG * 32 + (lane & 31), lane);
    }
}

template <int KS, int DIL>
inline bool launch_f16x3_cfg(const ConvParams& p, int cfg, int batch, hipStream_t s) {
    const int grid = batch * p.m_blks * p.n_tiles;
    switch (cfg) {
        case SPLIT_128x128: hipLaunchKernelGGL((conv_f16x3_kernel<KS, DIL, 4, 1, 4>), dim3(grid), dim3(256), 0, s, p); return true;
        case SPLIT_64x256:
            if constexpr (KS == 3 || KS >= 5) {   // (a 256-column window of several sub-chunks would not fit the static LDS limit)
                hipLaunchKernelGGL((conv_f16x3_kernel<KS, DIL, 2, 2, 4>), dim3(grid), dim3(256), 0, s, p);
                return true;
            }
            return false;
        case SPLIT_32x256:   // a single m-tile of rows (C_out = 32): the four waves split the columns, two n-tiles each
            if constexpr (KS == 3 || KS >= 5) {
                hipLaunchKernelGGL((conv_f16x3_kernel<KS, DIL, 1, 4, 2>), dim3(grid), dim3(256), 0, s, p);
                return true;
            }
            return false;
        default: return false;
    }
}

}  // namespace fv
