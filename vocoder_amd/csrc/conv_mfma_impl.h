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
//     k-steps of a tap with one coalesced global_load_dwordx4 straight from L2 (weights are L2-resident; no LDS),
//     software-prefetched one tap ahead.
//   K ordering inside a chunk: tap-major, then channel pair p; MFMA k-half h (lane>>5) = channel 2p+h.
//   Accumulators: MT x NT tiles of 32x32 fp32 (16 VGPRs each) per wave.
//
// Memory-op discipline (measured, profiles/README.md): every global load is UNCONDITIONAL on a clamped address and
// the bounds test is applied to the loaded value — a `cond ? load : 0` makes hipcc branch around each load and wait
// vmcnt(0) per element, which serialised the staging loads and the epilogue's residual loads.
#pragma once

#include "fv_internal.h"

namespace fv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// GELU = v * Phi(v) with Phi(-|v|) = erfc(|v| / sqrt 2) / 2 ~ poly(t) * exp(-v^2 / 2) / 2, t = 1 / (1 + p |v| / sqrt 2)
// (Abramowitz & Stegun 7.1.26, |eps_erf| <= 1.5e-7): 16 VALU instructions, two of them transcendental, no branches — about a
// third of the correctly-rounded erff.  Measured in fp32 against the exact function over [-12, 12]: |err| <= 4.3e-7 (torch's
// own fp32 nn.GELU: 1.2e-6); tests/test_gpu_conv.py pins it.
// ONE GELU for every kernel (round 5): a pointwise layer that moves between gemm_pw and the k = 1 conv kernel keeps its bits.
__device__ __forceinline__ float gelu_fast(float v) {
    const float z = fabsf(v) * 0.70710678f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(v * v * -0.72134752f);   // exp(-v^2 / 2)
    const float h = 0.5f * poly * e;
    return v * (v >= 0.f ? 1.0f - h : h);
}

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case FV_ACT_SILU: return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        case FV_ACT_LEAKY_RELU: return v >= 0.f ? v : v * slope;
        case FV_ACT_GELU: return gelu_fast(v);
        case FV_ACT_TANH: return tanhf(v);
        case FV_ACT_LOG_CLAMP: return logf(fmaxf(v, 1e-5f));
        default: return v;
    }
}

// Buffer descriptor over `bytes` bytes at a wave-uniform address (readfirstlane makes the uniformity provable, otherwise
// hipcc wraps every buffer op in a waterfall loop).  Raw buffer loads return 0 and raw buffer stores are dropped for byte
// offsets >= bytes: that hardware bounds check replaces clamp / select / branch VALU code around flat accesses.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int N>
__device__ __forceinline__ void act_apply_all(float (&v)[N], int act, float slope) {
    switch (act) {   // one wave-uniform branch for the whole register tile, not one per value
        case FV_ACT_NONE: break;
        case FV_ACT_SILU:
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = v[i] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[i]));
            break;
        default:
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = act_apply(v[i], act, slope);
    }
}

// Fused epilogue shared by the conv kernels: bias, layer-scale, residual, post-activation, MRF accumulate, polyphase
// scatter.  acc follows the 32x32 MFMA C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
// All global traffic goes through raw buffer ops on byte offsets relative to the output tensor (masked elements get offset
// 0xFFFFFFFF: loads return 0, stores are dropped) — no 64-bit address arithmetic, no per-element branches.
// (conv_epilogue_cols: the caller supplies each n-tile's column offset and validity — conv_wino_impl.h's columns are output pairs)
#ifndef FV_X_EPI_ROWS
#define FV_X_EPI_ROWS 8
#endif
template <int MT, int NT, int RG = FV_X_EPI_ROWS / 4>
__device__ __forceinline__ void conv_epilogue_cols(const ConvParams& p, f32x16 (&acc)[MT][NT], int b, int mt0, const int (&coff)[NT],
                                                   const bool (&cok)[NT], int lane) {
    // in flat mode the tensor spans all batch items (b == 0), otherwise one item
    const unsigned span = (unsigned)((p.flat ? (long long)p.y_bstride * (p.n_total / p.N) : p.y_bstride) * 4);
    const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * p.y_bstride, span);
    const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(p.res ? p.res + (long long)b * p.y_bstride : p.y, span);
    const bool has_res = p.res != nullptr;
    const bool accum = p.out_mode == OUT_ACCUM;
    // RG row groups of 4 accumulator registers are processed together: their residual / accumulate operands are requested up
    // front, so an m-tile costs 16 / (4 RG) round trips to L2 / HBM instead of four.  Two groups (8 rows) fit the register
    // budget of three workgroups per CU; four (the whole m-tile) spill and measured +1.2 % on the headline step.  The gain of
    // two is small (-0.2 %): tools/probe_conv_timeline.py shows a 128 x 128 workgroup ~20 us in this function (12 us without a
    // residual), but that is bandwidth, not latency — every CU's workgroups start together, so the whole chip reads its
    // residual tiles and stores its outputs (2 x 49 MB at B = 32) in the same few microseconds, twice per launch.
    // (the Winograd kernels — conv_wino_impl.h, conv_wino4_impl.h, conv_wino44_impl.h — come here with the default RG = 2 for their general epilogue; their common case
    //  is a leaner path of their own that requests the whole m-tile's operands at once: their accumulator planes are dead by then)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int rq0 = 0; rq0 < 4; rq0 += RG) {
            unsigned off[4 * RG * NT];
            float val[4 * RG * NT], rv[4 * RG * NT], yo[4 * RG * NT];
#pragma unroll
            for (int rr4 = 0; rr4 < 4 * RG; ++rr4) {
                const int rq = rq0 + rr4 / 4, rr = rr4 % 4;
                const int m = (mt0 + i) * 32 + rr + 8 * rq + 4 * (lane >> 5);
                const bool mok = m < p.M;
                const int mc = mok ? m : 0;
                const float bias = p.bias[mc];
                const float gm = p.gamma ? p.gamma[mc] : 1.0f;
                int row_off, ph = 0;   // element offsets fit 32 bits (host guarantees the tensor is < 4 GiB)
                if (p.convt) {
                    const int co = mc / p.u;
                    ph = mc - co * p.u;
                    row_off = co * p.Tout;
                } else {
                    row_off = mc * p.N;
                }
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) {
                    const int t = coff[jn] + ph;
                    bool ok = mok && cok[jn];
                    if (p.convt) ok = ok && t >= 0 && t < p.Tout;
                    off[rr4 * NT + jn] = ok ? (unsigned)(row_off + t) * 4u : 0xFFFFFFFFu;
                    val[rr4 * NT + jn] = fmaf(acc[i][jn][rq * 4 + rr], p.acc_scale, bias) * gm;   // acc_scale == 1: exact
                }
            }
            if (has_res) {
#pragma unroll
                for (int q = 0; q < 4 * RG * NT; ++q) rv[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, off[q], 0, 0));
            }
            if (accum) {
#pragma unroll
                for (int q = 0; q < 4 * RG * NT; ++q) yo[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, off[q], 0, 0));
            }
            if (has_res) {
#pragma unroll
                for (int q = 0; q < 4 * RG * NT; ++q) val[q] += rv[q];
            }
            act_apply_all(val, p.post_act, p.slope);
            if (accum) {
#pragma unroll
                for (int q = 0; q < 4 * RG * NT; ++q) val[q] = (yo[q] + val[q]) * p.out_scale;
            }
#pragma unroll
            for (int q = 0; q < 4 * RG * NT; ++q) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[q]), yrs, off[q], 0, 0);
        }
    }
}

template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[MT][NT], int b, int mt0, int ncol0, int lane) {
    if (p.vec_store) {
        // Polyphase transposed conv with stride 8 (the first upsamplers, hifigan.py:175-187), bias only: GEMM row m = (c_out, phase m & 7),
        // and the accumulator layout gives a lane the four rows 8 rq + 4 (lane >> 5) + {0..3} of GEMM column n — four CONSECUTIVE output
        // samples t = 8 n - pad + 4 (lane >> 5) + {0..3} of one channel: one 16-byte store instead of four scattered dwords, and the two
        // half-waves of an instruction fill whole 32-byte sectors (round 4: these launches have K = 512 ... 1024 only, their epilogue was
        // a third of a workgroup's life).  The host sets vec_store only when every such quad is 16-byte aligned and entirely inside
        // or outside [0, Tout) (Tout % 4 == 0, pad % 4 == 0, aligned tensors) and nothing but the bias is applied.
        const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y + (long long)b * p.y_bstride, (unsigned)(p.y_bstride * 4));
        const int khalf = lane >> 5;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int m0 = (mt0 + i) * 32 + 8 * rq + 4 * khalf;
                const f32x4 bv = *(const f32x4*)(p.bias + m0);   // (bias is padded to whole 128-row blocks)
                const int row_off = (m0 >> 3) * p.Tout - p.pad_t + 4 * khalf;
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) {
                    const int n = ncol0 + jn * 32;
                    const int t = n * 8 - p.pad_t + 4 * khalf;
                    const bool ok = m0 < p.M && n < p.N && t >= 0 && t < p.Tout;
                    u32x4 v;
                    v.x = __float_as_uint(fmaf(acc[i][jn][4 * rq + 0], p.acc_scale, bv.x));
                    v.y = __float_as_uint(fmaf(acc[i][jn][4 * rq + 1], p.acc_scale, bv.y));
                    v.z = __float_as_uint(fmaf(acc[i][jn][4 * rq + 2], p.acc_scale, bv.z));
                    v.w = __float_as_uint(fmaf(acc[i][jn][4 * rq + 3], p.acc_scale, bv.w));
                    __builtin_amdgcn_raw_buffer_store_b128(v, yrs, ok ? (unsigned)(row_off + n * 8) * 4u : 0xFFFFFFFFu, 0, 0);
                }
            }
        return;
    }
    // column part of the element offset (independent of the row) and its validity
    int coff[NT];
    bool cok[NT];
#pragma unroll
    for (int jn = 0; jn < NT; ++jn) {
        const int n = ncol0 + jn * 32;
        if (p.convt) {
            coff[jn] = n * p.u - p.pad_t;   // + phase added per row
            cok[jn] = n < p.N;
        } else if (p.flat) {
            const int bb = n / p.N;         // column = (batch item, t)
            coff[jn] = bb * (int)p.y_bstride + (n - bb * p.N);
            cok[jn] = n < p.n_total;
        } else {
            coff[jn] = n;
            cok[jn] = n < p.N;
        }
    }
    conv_epilogue_cols<MT, NT>(p, acc, b, mt0, coff, cok, lane);
}

#ifndef FV_X_INTERLEAVE
#define FV_X_INTERLEAVE 1
#endif
#ifndef FV_X_DA
#define FV_X_DA 3
#endif
constexpr int kWeightPrefetch = FV_X_DA;   // weight-fragment prefetch distance in k-steps (taps)

// 8-channel sub-chunks staged per barrier (LDS budget 2 * 8*SUBS * W floats, <= 36 KiB)
#ifndef FV_X_SUBS_LONG
#define FV_X_SUBS_LONG 2
#endif
#ifndef FV_X_SUBS_MID
#define FV_X_SUBS_MID 2
#endif
constexpr int subs_for(int ks, int w) {
    // measured: k = 3 +4 % with 16-channel chunks, +2 % more with 32.  k >= 7: 16-channel chunks were -3 % in round 1 (before the
    // operand pipeline and the interleaved memory operations) and are +0.8 % on the B = 32 step now (14.89 -> 14.77 ms, three
    // interleaved rounds; 32 channels: 14.83); half the barriers and staging phases per tile
    int s = ks <= 3 ? 4 : (ks <= 4 ? FV_X_SUBS_MID : FV_X_SUBS_LONG);
    while (s > 1 && s * kChunk * w > 4608) s /= 2;
    return s;
}

// min 3 waves per SIMD for the 64-accumulator tiles, 4 for the smaller ones: caps VGPR+AGPR so that several workgroups
// stay resident per CU (their MFMA phases cover each other's staging / barrier phases)
// SUM3: the input is ((x + x2) + x3) / 3 — the stack-mean of the three ResBlock branches of the previous stage, formed here on
// the way into LDS instead of by a pass of its own (mean_of_three_kernel: 4 tensors of HBM traffic, 65 us per wide stage at B = 32,
// ~10 us of launch + dependency per stage for a single clip).  Two more staging register sets: two workgroups per CU.
template <int KS, int DIL, int WM, int WN, int MT, int NT, bool SUM3 = false>
#ifndef FV_X_SUM3_OCC
#define FV_X_SUM3_OCC 0
#endif
// (FV_X_SUM3_OCC=1, experiment: asks for four / three workgroups per CU in the narrow SUM3 tiles — the HBM-bound last upsamplers are bound by bytes in flight.
//  Measured nothing on top of dropping the flat-mode row plan from these instances, which already took the 64 x 128 tile from two to three: LOG R6.12)
__global__ __launch_bounds__(256, (SUM3 ? (FV_X_SUM3_OCC && MT * NT == 1 ? 4 : (FV_X_SUM3_OCC && MT * NT == 2 ? 3 : 2)) : (NT >= 4 ? 2 : (MT * NT >= 4 ? 3 : 4)))) void conv_mfma_kernel(const ConvParams p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int N_BLK = WN * NT * 32;
    constexpr int SPAN = (KS - 1) * DIL;
    constexpr int W = N_BLK + SPAN;                    // staged columns per channel row
    // channels per LDS chunk = 8 * SUBS: short kernels (pointwise, polyphase convT, k=3) stage more channels per
    // barrier so that the MFMA run between two barriers stays long
    constexpr int SUBS = subs_for(KS, W);
    constexpr int CH = kChunk * SUBS;
    // (a wave-private window variant without workgroup barriers measured 17 % slower: extra halo traffic)
    constexpr int WL = W;                              // window width staged by the workgroup
    constexpr int TOT = CH * WL;
    constexpr int NTHR = 256;                          // threads staging the window
    constexpr int NE = (TOT + NTHR - 1) / NTHR;        // staged elements per thread
    __shared__ float xs[2][TOT];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int bid = blockIdx.x;
    const int n_tile = bid % p.n_tiles;
    bid /= p.n_tiles;
    const int m_blk = bid % p.m_blks;
    const int b = bid / p.m_blks;      // 0 in flat mode (the grid then has no batch factor)
    const int n0 = n_tile * N_BLK;
    const float* __restrict__ xb = p.x + (long long)b * p.x_bstride;
    const bool flat = KS == 1 && !SUM3 && p.flat;   // (SUM3: never flat — the transposed convs that use it have a halo; the row plan of flat mode is dead code there)

#ifdef FV_X_CONV_TS
#define FV_CV_STAMP(i) do { if (p.dbg_ts && threadIdx.x == 0) p.dbg_ts[(long long)blockIdx.x * 16 + (i)] = (long long)wall_clock64(); } while (0)
    if (p.dbg_ts && threadIdx.x == 0) p.dbg_ts[(long long)blockIdx.x * 16 + 15] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) | ((long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32);   // HW_ID, XCC_ID
#else
#define FV_CV_STAMP(i) do { } while (0)
#endif
    FV_CV_STAMP(0);
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- staging plan: thread handles elements e = sid + i*NTHR of the [CH][WL] chunk window (same for every chunk) ----
    // st_voff = byte offset of the element relative to the chunk's first channel row, or 0xFFFFFFFF outside [0, Tin):
    // the loads are raw buffer loads on a per-chunk descriptor, so out-of-range offsets — that marker, and rows of
    // zero-padded channels past C_in — come back as 0 from the hardware bounds check (no clamp / select / 64-bit VALU).
    unsigned st_voff[NE];
    int st_row[KS == 1 ? NE : 1];   // flat mode only: the descriptor spans every batch item, rows are checked explicitly
    const int sid = tid;
    const int tbase = n0 - p.pad_l;
    // Pointwise convs over flattened columns with an even T (p.flat == 2): columns (2q, 2q + 1) are adjacent in memory,
    // 8-byte aligned and never straddle two batch items, so the window is staged as pairs — half the vector-memory and
    // LDS-write instructions (staging is 15 - 25 % of a k = 1 launch: one chunk feeds only 4 k-steps, not 4 * KS).
    // Pair q of the chunk window is elements (2q, 2q + 1) of the same linear [CH][WL] layout; the plan of the NE2 pairs
    // lives in the first half of the same arrays.
    constexpr int NE2 = KS == 1 ? (TOT / 2 + NTHR - 1) / NTHR : 1;
    static_assert(KS != 1 || (TOT % (2 * NTHR) == 0 && NE == 2 * NE2), "pair staging covers the window exactly");
    const bool pair2 = KS == 1 && p.flat == 2;
    if (KS == 1 && pair2) {
#pragma unroll
        for (int i = 0; i < NE2; ++i) {
            const int q = sid + i * NTHR;
            const int r = q / (WL / 2);
            int t = tbase + 2 * (q - r * (WL / 2));
            const bool ok = t < p.n_total;
            const int bb = t / p.N;
            t -= bb * p.N;
            st_voff[i] = ok ? (unsigned)(bb * (int)p.x_bstride + r * p.Tin + t) * 4u : 0xFFFFFFF8u;
            st_row[KS == 1 ? i : 0] = r;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            int e = sid + i * NTHR;
            const bool in_tile = e < TOT;
            e = in_tile ? e : TOT - 1;
            const int r = e / WL;
            const int col = e - r * WL;
            int t = tbase + col;
            int boff = 0;
            bool ok = in_tile;
            if (flat) {   // column = (batch item, t)
                ok = ok && t < p.n_total;
                const int bb = t / p.N;
                t -= bb * p.N;
                boff = bb * (int)p.x_bstride;
            }
            ok = ok && t >= 0 && t < p.Tin;
            st_voff[i] = ok ? (unsigned)(boff + r * p.Tin + t) * 4u : 0xFFFFFFFFu;
            if (KS == 1) st_row[i < NE ? i : 0] = r;
        }
    }
    // Staging is split in two halves one chunk apart: load_chunk only ISSUES the loads (no dependent ALU, so no wait),
    // store_chunk — one chunk of MFMAs later — applies the activation and writes LDS (act(0) == 0 keeps the padding).
    float stage[NE];
    float stage2[SUM3 ? NE : 1], stage3[SUM3 ? NE : 1];
    const long long x_items = flat ? (long long)(p.n_total / p.N) : 1;   // batch items spanned by the descriptor
    auto load_chunk = [&](int c) {
        const int cbase = c * CH;
        const long long span = x_items * p.x_bstride - (long long)cbase * p.Tin;   // elements from this chunk's first row
        const long long rows = (long long)(p.Cin - cbase) * p.Tin;
        const __amdgpu_buffer_rsrc_t xrs =
            uniform_rsrc(xb + (long long)cbase * p.Tin, (unsigned)((flat ? span : (rows < span ? rows : span)) * 4));
        if (KS == 1 && pair2) {
#pragma unroll
            for (int i = 0; i < NE2; ++i) {
                const unsigned off = st_row[KS == 1 ? i : 0] < p.Cin - cbase ? st_voff[i] : 0xFFFFFFF8u;
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, off, 0, 0);
                stage[(2 * i) % NE] = __uint_as_float(v.x);
                stage[(2 * i + 1) % NE] = __uint_as_float(v.y);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            unsigned off = st_voff[i];
            if (KS == 1 && flat) off = st_row[KS == 1 ? i : 0] < p.Cin - cbase ? off : 0xFFFFFFFFu;
            stage[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, off, 0, 0));
        }
        if constexpr (SUM3) {   // (never flat: the transposed convs that use it have a halo)
            const unsigned bytes = (unsigned)((rows < span ? rows : span) * 4);
            const long long boff = (long long)b * p.x_bstride + (long long)cbase * p.Tin;
            const __amdgpu_buffer_rsrc_t xrs2 = uniform_rsrc(p.x2 + boff, bytes), xrs3 = uniform_rsrc(p.x3 + boff, bytes);
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                stage2[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs2, st_voff[i], 0, 0));
                stage3[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs3, st_voff[i], 0, 0));
            }
        }
    };
    auto act_in = [&](float v) {
        if (p.pre_act == FV_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        if (p.pre_act != FV_ACT_NONE) return act_apply(v, p.pre_act, p.slope);
        return v;
    };
    auto store_chunk = [&](float* dst, int c) {
        (void)c;
        if (KS == 1 && pair2) {
#pragma unroll
            for (int i = 0; i < NE2; ++i)
                reinterpret_cast<float2*>(dst)[sid + i * NTHR] =
                    make_float2(act_in(stage[(2 * i) % NE]), act_in(stage[(2 * i + 1) % NE]));
            return;
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = sid + i * NTHR;
            float v = stage[i];
            if constexpr (SUM3) v = ((v + stage2[i]) + stage3[i]) * (1.0f / 3.0f);   // the additions and the scale of the accumulate chain
            v = act_in(v);
            if (e < TOT) dst[e] = v;
        }
    };

    const int mt0 = (m_blk * WM + wm) * MT;
    // Weights of m-tile mt are one contiguous stream over the global k-step index g = sub_chunk * KS + tap.  They are
    // fetched with raw buffer loads: descriptor + wave-uniform byte offset in SGPRs, the per-lane part (lane * 16 B) in
    // one constant VGPR — no 64-bit VALU address arithmetic in the MFMA loop (it cost ~10 % of the kernel as flat loads).
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    const int wstride_b = p.nchunk * KS * 1024;   // bytes per m-tile (64 lanes x 16 B per k-step)
    // byte offset of k-step g of this wave's m-tile i = wbase[i] + g * 1024, with wbase[i] held in an SGPR
    int wbase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) wbase[i] = __builtin_amdgcn_readfirstlane((mt0 + i) * wstride_b);
    auto load_a = [&](float4 (&dst)[MT], int goff_b) {   // goff_b = g * 1024, wave-uniform
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wbase[i] + goff_b, 0);
            dst[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    };
    const int b_lane = (lane >> 5) * WL + wn * (NT * 32) + (lane & 31);

    auto load_b = [&](float (&dst)[4][NT], const float* xsb, int sub, int j) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) dst[pp][jn] = xsb[b_lane + (sub * kChunk + 2 * pp) * WL + jn * 32 + j * DIL];
    };

    // Software pipeline.  Activation fragments (LDS) run one k-step ahead.  Weight fragments (global, L2) run DA k-steps
    // ahead: vmcnt retires loads in issue order, so the first weight wait after the next chunk's staging loads were issued
    // also waits for those (HBM latency); keeping DA weight loads older than the staging loads gives them DA taps
    // (DA * 4*MT*NT MFMAs) to land.  The sched_barriers pin the issue order; without them hipcc sinks every load next to
    // its first use.  Weight prefetch may run past the last k-step: the packed buffer carries DA+1 steps of padding.
    constexpr int STEPS = SUBS * KS;
    constexpr int DA = kWeightPrefetch;
    float4 aq[DA + 1][MT];
    float b_cur[4][NT], b_nxt[4][NT];
    const int nch = (p.nchunk_real + SUBS - 1) / SUBS;   // LDS chunks; packed weights are padded to whole chunks
    load_chunk(0);
#pragma unroll
    for (int d = 0; d < DA; ++d) load_a(aq[d], d * 1024);
    for (int c = 0; c < nch; ++c) {
        float* xsb = xs[c & 1];
        store_chunk(xsb, c);
        __syncthreads();
        if (c < 12) FV_CV_STAMP(1 + c);
        if (c + 1 < nch) load_chunk(c + 1);
        // readfirstlane: c is wave-uniform, this makes the weight offsets provably so (else hipcc wraps every buffer
        // load in a waterfall loop)
        const int gchunk_b = __builtin_amdgcn_readfirstlane((c * STEPS + DA) * 1024);
        load_b(b_cur, xsb, 0, 0);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
#if FV_X_INTERLEAVE
            // the step's memory operations (MT weight loads for k-step st + DA, 4 NT LDS fragment reads for k-step st + 1)
            // spread between its MFMAs instead of issued in a burst before them: an in-order wave hides a memory
            // instruction's issue time only under an MFMA that is already executing (gemm_pw.hip, round 2)
            {
                constexpr int NM = 4 * MT * NT, NLDX = MT + 4 * NT;
                const int sub_n = (st + 1) / KS, j_n = (st + 1) % KS;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int pp = m / (MT * NT), i = (m / NT) % MT, jn = m % NT;
                    const float av = pp == 0 ? aq[0][i].x : pp == 1 ? aq[0][i].y : pp == 2 ? aq[0][i].z : aq[0][i].w;
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_cur[pp][jn], acc[i][jn], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < NLDX; ++k) {
                        if (k * NM / NLDX == m) {
                            if (k < MT) {
                                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wbase[k] + gchunk_b + st * 1024, 0);
                                aq[DA][k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                            } else if (st + 1 < STEPS) {
                                const int pp2 = (k - MT) / NT, jn2 = (k - MT) % NT;
                                b_nxt[pp2][jn2] = xsb[b_lane + (sub_n * kChunk + 2 * pp2) * WL + jn2 * 32 + j_n * DIL];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
#else
            load_a(aq[DA], gchunk_b + st * 1024);
            if (st + 1 < STEPS) load_b(b_nxt, xsb, (st + 1) / KS, (st + 1) % KS);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float av = pp == 0 ? aq[0][i].x : pp == 1 ? aq[0][i].y : pp == 2 ? aq[0][i].z : aq[0][i].w;
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_cur[pp][jn], acc[i][jn], 0, 0, 0);
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < DA; ++d)
#pragma unroll
                for (int i = 0; i < MT; ++i) aq[d][i] = aq[d + 1][i];
            if (st + 1 < STEPS) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn) b_cur[pp][jn] = b_nxt[pp][jn];
            }
        }
    }

    FV_CV_STAMP(13);
    conv_epilogue<MT, NT>(p, acc, b, mt0, n0 + wn * (NT * 32) + (lane & 31), lane);
#ifdef FV_X_CONV_TS
    __builtin_amdgcn_s_waitcnt(0);
#endif
    FV_CV_STAMP(14);
}

// Latency variant for launches that cannot fill the chip with regular tiles (small batch x short T): the workgroup
// owns a 32 x 64 (or 32 x 32) output tile and its four waves split K between them — wave w takes the w-th 8-channel
// sub-chunk of every 32-channel LDS chunk, so its weights for one tap are the whole float4 of the packed layout: one
// coalesced 16-byte load per lane and tap feeds four MFMAs.  (The first version gave wave w dword w of every float4 —
// four times the load instructions, each at a 16-byte lane stride; on a single-clip forward those loads cost as much as the
// MFMA chain itself.  An eight-wave version spilled and measured slower.)  The staged window and the barriers are shared,
// partial accumulators are reduced through LDS and wave 0 runs the epilogue.  16x more workgroups than the 128 x 128 tile
// for the same problem.
#ifdef FV_X_SPLITK_TS
#define FV_SK_STAMP(i) do { if (p.dbg_ts && threadIdx.x == 0) p.dbg_ts[(long long)blockIdx.x * 8 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define FV_SK_STAMP(i) do { } while (0)
#endif
template <int KS, int DIL, int NT, int NW, bool SUM3 = false>
__global__ __launch_bounds__(NW * 64, 2) void conv_mfma_splitk_kernel(const ConvParams p) {
    FV_SK_STAMP(0);
    static_assert(NW == 4, "four waves split K: one 8-channel sub-chunk of the 32-channel LDS chunk each");
    constexpr int THREADS = NW * 64;
    constexpr int CHW = 8 * NW;              // channels per LDS chunk (= per barrier): rows 8w .. 8w + 7 belong to wave w
    constexpr int N_BLK = NT * 32;
    constexpr int SPAN = (KS - 1) * DIL;
    constexpr int W = N_BLK + SPAN;
    constexpr int TOT = CHW * W;
    constexpr int NE = (TOT + THREADS - 1) / THREADS;
    __shared__ float xs[2][TOT];
    __shared__ float red[NW - 1][NT * 16][64];   // [wave-1][acc register][lane]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

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

    // (every index into the per-thread staging arrays below is a compile-time constant — static_for, not unrolled loops — and
    //  the helper lambdas are always_inline: with 11+ elements per thread hipcc otherwise outlines store_chunk, which leaves the
    //  arrays in scratch memory and makes every window load synchronous: a 4x slower kernel)
    unsigned st_voff[NE];   // byte offset inside the chunk's rows, 0xFFFFFFFF outside [0, Tin) (see conv_mfma_kernel)
    const int tbase = n0 - p.pad_l;
    static_for<NE>([&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        int e = tid + i * THREADS;
        const bool in_tile = e < TOT;
        e = in_tile ? e : TOT - 1;
        const int r = e / W;
        const int col = e - r * W;
        const int t = tbase + col;
        const bool ok = in_tile && t >= 0 && t < p.Tin;
        st_voff[i] = ok ? (unsigned)(r * p.Tin + t) * 4u : 0xFFFFFFFFu;
    });
    // Prefetch distance in LDS chunks for both operands: a chunk holds 4 * KS * NT MFMAs per wave (0.33 us at KS = 3, NT = 1;
    // 1.2 us at KS = 11) against a round trip to L2 / HBM of 1 ... 2 us, and a single-clip launch has no other wave to
    // cover it.  The operand rings have PF + 1 slots addressed by compile-time indices (the chunk loop is unrolled by the
    // ring size): shifting the rings instead — registers with loads still in flight — made every chunk end wait for the
    // loads it had just issued, which pinned the real distance at one chunk whatever PF said.
#ifndef FV_X_SPLITK_PF_LONG
#define FV_X_SPLITK_PF_LONG 1
#endif
#ifndef FV_X_SPLITK_PF_MID
#define FV_X_SPLITK_PF_MID 2
#endif
#ifndef FV_X_SPLITK_PF_SHORT
#define FV_X_SPLITK_PF_SHORT 3
#endif
    constexpr int PF = KS >= 8 ? FV_X_SPLITK_PF_LONG : (KS >= 5 ? FV_X_SPLITK_PF_MID : FV_X_SPLITK_PF_SHORT);
    constexpr int RING = PF + 1;
    float stage[RING][NE];
    float stage2[SUM3 ? RING : 1][SUM3 ? NE : 1], stage3[SUM3 ? RING : 1][SUM3 ? NE : 1];   // SUM3: see conv_mfma_kernel
    const float* __restrict__ xb2 = SUM3 ? p.x2 + (long long)b * p.x_bstride : nullptr;
    const float* __restrict__ xb3 = SUM3 ? p.x3 + (long long)b * p.x_bstride : nullptr;
    auto load_from = [&](const float* __restrict__ base, float (&dst)[SUM3 ? NE : 1], int c) __attribute__((always_inline)) {   // SUM3 operands
        const int cbase = c * CHW;
        const int rows = p.Cin - cbase > 0 ? p.Cin - cbase : 0;
        const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(base + (long long)(rows ? cbase : 0) * p.Tin, (unsigned)(rows * p.Tin) * 4u);
        static_for<(SUM3 ? NE : 1)>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            dst[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, st_voff[i], 0, 0));
        });
    };
    auto load_chunk = [&](auto slot_c, int c) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_c)::value;
        const int cbase = c * CHW;
        // chunks past the last one (prefetch overrun) get an empty descriptor: every load returns 0
        const int rows = p.Cin - cbase > 0 ? p.Cin - cbase : 0;
        const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(xb + (long long)(rows ? cbase : 0) * p.Tin, (unsigned)(rows * p.Tin) * 4u);
        static_for<NE>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            stage[SL][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, st_voff[i], 0, 0));
        });
        if constexpr (SUM3) {
            load_from(xb2, stage2[SL], c);
            load_from(xb3, stage3[SL], c);
        }
    };
    auto store_chunk = [&](float* dst, auto slot_c) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_c)::value;
        static_for<NE>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            const int e = tid + i * THREADS;
            float v = stage[SL][i];
            if constexpr (SUM3) v = ((v + stage2[SL][i]) + stage3[SL][i]) * (1.0f / 3.0f);
            if (e < TOT) dst[e] = act_apply(v, p.pre_act, p.slope);
        });
    };

    // this wave's float4s of the packed weights: 8-channel sub-chunk 4c + wave, one float4 (four k-steps) per lane and tap,
    // by raw buffer loads (SGPR base + constant VGPR part).  The packed buffer is zero-padded to whole groups of four
    // sub-chunks plus eight k-steps, and the descriptor is unbounded, so nothing a prefetch touches is out of bounds.
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    const int wtile_b = __builtin_amdgcn_readfirstlane(m_blk * p.nchunk * KS * 1024);
    auto load_a = [&](int c, int j) __attribute__((always_inline)) {
        const int soff = __builtin_amdgcn_readfirstlane(wtile_b + ((c * NW + wave) * KS + j) * 1024);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, soff, 0));
    };
    // k-step q of the wave's sub-chunk: lanes 0..31 supply channel 8w + 2q, lanes 32..63 channel 8w + 2q + 1
    const int b_lane = (8 * wave + (lane >> 5)) * W + (lane & 31);
    auto load_b = [&](float (&dst)[NT], const float* xsb, int sj) __attribute__((always_inline)) {   // sj = tap * 4 + q
        const int j = sj >> 2, q = sj & 3;
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) dst[jn] = xsb[b_lane + 2 * q * W + jn * 32 + j * DIL];
    };

    constexpr int KSS = 4 * KS;   // k-steps per wave and LDS chunk
    // B fragments are requested DB k-steps ahead: a single-clip launch has one wave per SIMD, so nothing else covers the
    // LDS round trip (~120 cycles against 64 per MFMA: one step ahead left every MFMA waiting for its operand — the K loop
    // ran at 115 cycles per MFMA)
#ifndef FV_X_SPLITK_DB
#define FV_X_SPLITK_DB 3
#endif
    constexpr int DB = FV_X_SPLITK_DB < KSS ? FV_X_SPLITK_DB : KSS - 1;
    f32x4 aq[RING][KS];
    float bq[DB + 1][NT];
    const int nchunks = (p.Cin + CHW - 1) / CHW;
    const int last = nchunks - 1;
    static_for<PF>([&](auto d_c) {
        constexpr int d = decltype(d_c)::value;
        load_chunk(d_c, d);
#pragma unroll
        for (int j = 0; j < KS; ++j) aq[d][j] = load_a(d <= last ? d : last, j);
    });
    for (int c0 = 0; c0 < nchunks; c0 += RING) {
        static_for<RING>([&](auto slot_c) __attribute__((always_inline)) {
            constexpr int S = decltype(slot_c)::value;        // ring slot of chunk c
            constexpr int SN = (S + PF) % RING;               // slot of chunk c + PF (the one chunk c - 1 used)
            const int c = c0 + S;
            if (c < nchunks) {
                float* xsb = xs[c & 1];
                store_chunk(xsb, slot_c);
                __syncthreads();
                if (c == 0) FV_SK_STAMP(1);
                // The NL loads of chunk c + PF are issued one at a time BETWEEN the MFMAs, window elements first: a wave issues
                // in order, and a burst of 17 loads ahead of the first MFMA (x 4 waves into one CU's address path) held
                // every chunk's MFMA chain back by ~0.8 us
                const int cx = c + PF;
                const int cbase = cx * CHW;
                const int rows = p.Cin - cbase > 0 ? p.Cin - cbase : 0;   // past the last chunk: empty descriptor, loads return 0
                const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(xb + (long long)(rows ? cbase : 0) * p.Tin, (unsigned)(rows * p.Tin) * 4u);
                const __amdgpu_buffer_rsrc_t xrs2 = uniform_rsrc((SUM3 ? xb2 : xb) + (long long)(rows ? cbase : 0) * p.Tin, (unsigned)(rows * p.Tin) * 4u);
                const __amdgpu_buffer_rsrc_t xrs3 = uniform_rsrc((SUM3 ? xb3 : xb) + (long long)(rows ? cbase : 0) * p.Tin, (unsigned)(rows * p.Tin) * 4u);
                const int cn = cx <= last ? cx : last;
                constexpr int NL = NE + KS;
                constexpr int LSTEP = KSS / NL > 0 ? KSS / NL : 1;
                auto issue_load = [&](auto idx_c) __attribute__((always_inline)) {   // compile-time index: the rings must stay in registers
                    constexpr int idx = decltype(idx_c)::value;
                    if constexpr (idx < NE) {
                        stage[SN][idx] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, st_voff[idx], 0, 0));
                        if constexpr (SUM3) {
                            stage2[SN][idx] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs2, st_voff[idx], 0, 0));
                            stage3[SN][idx] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs3, st_voff[idx], 0, 0));
                        }
                    } else if constexpr (idx < NL) {
                        aq[SN][idx - NE] = load_a(cn, idx - NE);
                    }
                };
#pragma unroll
                for (int d = 0; d < DB; ++d) load_b(bq[d], xsb, d);
                __builtin_amdgcn_sched_barrier(0);
                static_for<KSS>([&](auto sj_c) __attribute__((always_inline)) {
                    constexpr int sj = decltype(sj_c)::value;
                    if constexpr (sj + DB < KSS) load_b(bq[(sj + DB) % (DB + 1)], xsb, sj + DB);
                    if constexpr (sj % LSTEP == 0) issue_load(std::integral_constant<int, sj / LSTEP>{});
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn)
                        acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[S][sj >> 2][sj & 3], bq[sj % (DB + 1)][jn], acc[0][jn], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
                constexpr int ISSUED = (KSS + LSTEP - 1) / LSTEP;
                static_for<(NL > ISSUED ? NL - ISSUED : 0)>([&](auto k_c) { issue_load(std::integral_constant<int, ISSUED + decltype(k_c)::value>{}); });
            }
        });
    }

    FV_SK_STAMP(2);
    // reduce the partial tiles
    if (wave > 0) {
#pragma unroll
        for (int jn = 0; jn < NT; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][jn * 16 + r][lane] = acc[0][jn][r];
    }
    __syncthreads();
    FV_SK_STAMP(3);
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < NW - 1; ++w)
#pragma unroll
            for (int jn = 0; jn < NT; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][jn][r] += red[w][jn * 16 + r][lane];
        // (requesting the residual / accumulate / bias operands of the whole tile at once, or before the K loop, instead of
        //  conv_epilogue's four load -> compute -> store groups changed nothing, or cost a wave of occupancy)
        conv_epilogue<1, NT>(p, acc, b, m_blk, n0 + (lane & 31), lane);
#ifdef FV_X_SPLITK_TS
        __builtin_amdgcn_s_waitcnt(0);   // stores issued and loads returned
#endif
        FV_SK_STAMP(4);
    }
}

// The same tile with the B operands loaded STRAIGHT FROM GLOBAL MEMORY (round 5): for the few-tap convs of a single-clip forward — the polyphase
// upsamplers (two taps per phase), conv_pre — a 32-channel chunk is 4 KS MFMAs per wave, and staging its window through LDS costs the kernel one
// workgroup barrier and one LDS round trip per 4 KS matrix instructions: the K loop of the stage-0 upsampler (16 chunks, 8 MFMAs each) ran at twice its
// matrix time.  Here a lane requests the element its B operand needs — channel 8 (4 c + w) + 2 q + (lane >> 5), column n0 + (lane & 31) + 32 jn + j D —
// one dword per MFMA, PF chunks ahead in a register ring (range-checked by the chunk's buffer descriptor: rows past C_in and columns outside [0, T_in)
// return 0 = the conv's zero padding after the activation); no LDS, no barrier until the reduction of the four waves' K shares.  SUM3: the three branch
// tensors are requested side by side and averaged on arrival.  The pre-activation runs per operand (a window element is the operand of up to KS taps).
// (which launches take it: splitk_direct_pf() / splitk_direct_shape() in fv_internal.h — conv_layer.hip names the launch by the same rule)
template <int KS, int DIL, int NT, bool SUM3>
constexpr bool splitk_direct_ok = splitk_direct_shape(KS, DIL, NT, SUM3);

template <int KS, int DIL, int NT, bool SUM3 = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_splitk_direct_kernel(const ConvParams p) {
    constexpr int NW = 4;
    constexpr int N_BLK = NT * 32;
    constexpr int KSS = 4 * KS;              // k-steps (MFMAs per n-tile) per wave and 32-channel chunk
    constexpr int NB = KSS * NT;             // B operands per wave and chunk
    __shared__ float red[NW - 1][NT * 16][64];   // [wave-1][acc register][lane]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int n_tile = bid % p.n_tiles;
    bid /= p.n_tiles;
    const int m_blk = bid % p.m_blks;
    const int b = bid / p.m_blks;
    const int n0 = n_tile * N_BLK;
    const float* __restrict__ xb = p.x + (long long)b * p.x_bstride;
    const float* __restrict__ xb2 = SUM3 ? p.x2 + (long long)b * p.x_bstride : nullptr;
    const float* __restrict__ xb3 = SUM3 ? p.x3 + (long long)b * p.x_bstride : nullptr;

    f32x16 acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    // byte offset of operand (tap j, k-step q, n-tile jn) inside the wave's 8-row sub-chunk; 0xFFFFFFFF outside [0, Tin)
    unsigned bvo[NB];
    {
        const int khalf = lane >> 5, col = lane & 31;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sj = i / NT, jn = i % NT, j = sj >> 2, q = sj & 3;
            const int t = n0 - p.pad_l + jn * 32 + col + j * DIL;
            bvo[i] = (t >= 0 && t < p.Tin) ? (unsigned)((2 * q + khalf) * p.Tin + t) * 4u : 0xFFFFFFFFu;
        }
    }
    // prefetch distance in chunks (both operands), by the registers a ring slot takes: NB (x 3: SUM3) operands + KS weight float4s
    constexpr int PF = splitk_direct_pf(KS, NT, SUM3);
    static_assert(PF > 0, "see splitk_direct_ok");
    constexpr int RING = PF + 1;
    float bq[RING][NB];
    float bq2[SUM3 ? RING : 1][SUM3 ? NB : 1], bq3[SUM3 ? RING : 1][SUM3 ? NB : 1];
    f32x4 aq[RING][KS];
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 0x7fffffff, 0x00020000);
    const int wvoff = lane * 16;
    const int wtile_b = __builtin_amdgcn_readfirstlane(m_blk * p.nchunk * KS * 1024);
    const int nchunks = (p.Cin + 8 * NW - 1) / (8 * NW);
    const int last = nchunks - 1;
    auto load_a = [&](int c, int j) __attribute__((always_inline)) {
        const int soff = __builtin_amdgcn_readfirstlane(wtile_b + ((c * NW + wave) * KS + j) * 1024);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, soff, 0));
    };
    // descriptors of the wave's sub-chunk of chunk c: rows 8 (4 c + w) ... of the item, as many as C_in leaves (none past the last chunk)
    auto rsrc_of = [&](const float* __restrict__ base, int c) __attribute__((always_inline)) {
        const int r0 = 8 * (c * NW + wave);
        const int rows = p.Cin - r0 > 0 ? p.Cin - r0 : 0;
        return uniform_rsrc(base + (long long)(rows ? r0 : 0) * p.Tin, (unsigned)(rows * p.Tin) * 4u);
    };
    // operand idx of chunk c -> slot SL: B operands first (x, then x2, x3), the KS weight float4s last
    constexpr int NL = (SUM3 ? 3 : 1) * NB + KS;
    auto issue_load = [&](auto sl_c, auto idx_c, const __amdgpu_buffer_rsrc_t& r1, const __amdgpu_buffer_rsrc_t& r2, const __amdgpu_buffer_rsrc_t& r3,
                          int cn) __attribute__((always_inline)) {
        constexpr int SL = decltype(sl_c)::value, idx = decltype(idx_c)::value;
        if constexpr (idx < NB) {
            bq[SL][idx] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, bvo[idx], 0, 0));
        } else if constexpr (SUM3 && idx < 2 * NB) {
            bq2[SL][idx - NB] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r2, bvo[idx - NB], 0, 0));
        } else if constexpr (SUM3 && idx < 3 * NB) {
            bq3[SL][idx - 2 * NB] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r3, bvo[idx - 2 * NB], 0, 0));
        } else if constexpr (idx < NL) {
            aq[SL][idx - (NL - KS)] = load_a(cn, idx - (NL - KS));
        }
    };
    static_for<PF>([&](auto d_c) {
        constexpr int d = decltype(d_c)::value;
        const __amdgpu_buffer_rsrc_t r1 = rsrc_of(xb, d), r2 = rsrc_of(SUM3 ? xb2 : xb, d), r3 = rsrc_of(SUM3 ? xb3 : xb, d);
        static_for<NL>([&](auto i_c) { issue_load(d_c, i_c, r1, r2, r3, d <= last ? d : last); });
    });
    for (int c0 = 0; c0 < nchunks; c0 += RING) {
        static_for<RING>([&](auto slot_c) __attribute__((always_inline)) {
            constexpr int S = decltype(slot_c)::value;        // ring slot of chunk c
            constexpr int SN = (S + PF) % RING;               // slot of chunk c + PF (the one chunk c - 1 used)
            const int c = c0 + S;
            if (c < nchunks) {
                float bv[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    float v = bq[S][i];
                    if constexpr (SUM3) v = ((v + bq2[S][i]) + bq3[S][i]) * (1.0f / 3.0f);
                    bv[i] = v;
                }
                act_apply_all(bv, p.pre_act, p.slope);
                const int cx = c + PF;
                const __amdgpu_buffer_rsrc_t r1 = rsrc_of(xb, cx), r2 = rsrc_of(SUM3 ? xb2 : xb, cx), r3 = rsrc_of(SUM3 ? xb3 : xb, cx);
                const int cn = cx <= last ? cx : last;
                constexpr int LPS = (NL + KSS - 1) / KSS;     // loads of chunk c + PF requested per k-step, between the MFMAs
                __builtin_amdgcn_sched_barrier(0);
                static_for<KSS>([&](auto sj_c) __attribute__((always_inline)) {
                    constexpr int sj = decltype(sj_c)::value;
                    static_for<LPS>([&](auto l_c) {
                        constexpr int idx = sj * LPS + decltype(l_c)::value;
                        if constexpr (idx < NL) issue_load(std::integral_constant<int, SN>{}, std::integral_constant<int, idx>{}, r1, r2, r3, cn);
                    });
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn)
                        acc[0][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[S][sj >> 2][sj & 3], bv[sj * NT + jn], acc[0][jn], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        });
    }

    // reduce the partial tiles
    if (wave > 0) {
#pragma unroll
        for (int jn = 0; jn < NT; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][jn * 16 + r][lane] = acc[0][jn][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < NW - 1; ++w)
#pragma unroll
            for (int jn = 0; jn < NT; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][jn][r] += red[w][jn * 16 + r][lane];
        conv_epilogue<1, NT>(p, acc, b, m_blk, n0 + (lane & 31), lane);
    }
}

template <int KS, int DIL, int WM, int WN, int MT, int NT>
inline void launch_one(const ConvParams& p, int batch, hipStream_t s) {
    const int grid = batch * p.m_blks * p.n_tiles;
    if constexpr (KS == 1 || KS == 2 || KS == 4) {   // tap counts of the polyphase transposed convs (the upsamplers): SUM3 variants
        if (p.x2) {
            hipLaunchKernelGGL((conv_mfma_kernel<KS, DIL, WM, WN, MT, NT, true>), dim3(grid), dim3(256), 0, s, p);
            return;
        }
    }
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
        case TILE_256x32:
            if constexpr (KS == 1) { launch_one<KS, DIL, 4, 1, 2, 1>(p, batch, s); return true; }
            return false;
        case TILE_128x96:
            if constexpr (KS == 2) { launch_one<KS, DIL, 4, 1, 1, 3>(p, batch, s); return true; }
            return false;
        case TILE_256x64:
            if constexpr (KS == 1 || KS == 2) { launch_one<KS, DIL, 4, 1, 2, 2>(p, batch, s); return true; }   // (KS == 2: the stride-8 upsamplers, 1024 / 2048 GEMM rows)
            return false;
        case TILE_SPLITK_32x64:
            if constexpr (splitk_direct_ok<KS, DIL, 2, true> && KS != 7) {
                if (knobs().splitk_direct && p.x2) {
                    hipLaunchKernelGGL((conv_mfma_splitk_direct_kernel<KS, DIL, 2, true>), dim3(batch * p.m_blks * p.n_tiles), dim3(256), 0, s, p);
                    return true;
                }
            }
            if constexpr (splitk_direct_ok<KS, DIL, 2, false>) {
                if (knobs().splitk_direct && !p.x2) {
                    hipLaunchKernelGGL((conv_mfma_splitk_direct_kernel<KS, DIL, 2>), dim3(batch * p.m_blks * p.n_tiles), dim3(256), 0, s, p);
                    return true;
                }
            }
            if constexpr (KS == 1 || KS == 2 || KS == 4) {
                if (p.x2) {
                    hipLaunchKernelGGL((conv_mfma_splitk_kernel<KS, DIL, 2, 4, true>), dim3(batch * p.m_blks * p.n_tiles), dim3(256), 0, s, p);
                    return true;
                }
            }
            hipLaunchKernelGGL((conv_mfma_splitk_kernel<KS, DIL, 2, 4>), dim3(batch * p.m_blks * p.n_tiles), dim3(256), 0, s, p);
            return true;
        case TILE_SPLITK_32x32:
            if constexpr (splitk_direct_ok<KS, DIL, 1, true> && KS != 7) {
                if (knobs().splitk_direct && p.x2) {
                    hipLaunchKernelGGL((conv_mfma_splitk_direct_kernel<KS, DIL, 1, true>), dim3(batch * p.m_blks * p.n_tiles), dim3(256), 0, s, p);
                    return true;
                }
            }
            if constexpr (splitk_direct_ok<KS, DIL, 1, false>) {
                if (knobs().splitk_direct && !p.x2) {
                    hipLaunchKernelGGL((conv_mfma_splitk_direct_kernel<KS, DIL, 1>), dim3(batch * p.m_blks * p.n_tiles), dim3(256), 0, s, p);
                    return true;
                }
            }
            if constexpr (KS == 1 || KS == 2 || KS == 4) {
                if (p.x2) {
                    hipLaunchKernelGGL((conv_mfma_splitk_kernel<KS, DIL, 1, 4, true>), dim3(batch * p.m_blks * p.n_tiles), dim3(256), 0, s, p);
                    return true;
                }
            }
            hipLaunchKernelGGL((conv_mfma_splitk_kernel<KS, DIL, 1, 4>), dim3(batch * p.m_blks * p.n_tiles), dim3(256), 0, s, p);
            return true;
        default: return false;
    }
}

}  // namespace fv
